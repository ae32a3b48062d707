"""Alias of collision_amd.misc so that `import collision.misc` keeps working (drop-in import path)."""
from collision_amd.misc import *  # noqa: F401,F403
from collision_amd import misc as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

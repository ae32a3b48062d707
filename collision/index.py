"""Alias of collision_amd.index so that `import collision.index` keeps working (drop-in import path)."""
from collision_amd.index import *  # noqa: F401,F403
from collision_amd import index as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

"""Alias of collision_amd.bounds so that `import collision.bounds` keeps working (drop-in import path)."""
from collision_amd.bounds import *  # noqa: F401,F403
from collision_amd import bounds as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

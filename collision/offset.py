"""Alias of collision_amd.offset so that `import collision.offset` keeps working (drop-in import path)."""
from collision_amd.offset import *  # noqa: F401,F403
from collision_amd import offset as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

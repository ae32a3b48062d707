"""Alias of collision_amd.collision so that `import collision.collision` keeps working (drop-in import path)."""
from collision_amd.collision import *  # noqa: F401,F403
from collision_amd import collision as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

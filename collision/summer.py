"""Alias of collision_amd.summer so that `import collision.summer` keeps working (drop-in import path)."""
from collision_amd.summer import *  # noqa: F401,F403
from collision_amd import summer as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

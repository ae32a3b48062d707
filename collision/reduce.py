"""Alias of collision_amd.reduce so that `import collision.reduce` keeps working (drop-in import path)."""
from collision_amd.reduce import *  # noqa: F401,F403
from collision_amd import reduce as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

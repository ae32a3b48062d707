"""Alias of collision_amd.scan so that `import collision.scan` keeps working (drop-in import path)."""
from collision_amd.scan import *  # noqa: F401,F403
from collision_amd import scan as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

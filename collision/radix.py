"""Alias of collision_amd.radix so that `import collision.radix` keeps working (drop-in import path)."""
from collision_amd.radix import *  # noqa: F401,F403
from collision_amd import radix as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

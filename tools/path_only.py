"""Runs only the single-GPU path a few times, for rocprofv3 passes.
    python tools/path_only.py [steps] [n_spheres] [plan: auto|lsd|msd] [scene: uniform|config3] [leaf-block k] [traverse variant]
(leaf-block k: col_debug_leaf_blocks, 0 = no marks; traverse variant: col_debug_traverse, 128 = the walk without block code;
COLLISION_PATH_DTYPE=float64 in the environment: float64 coordinates and radii; COLLISION_LBVH_MODE: col_debug_lbvh)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from collision_amd import hip
from collision_amd.collision import Collider

ctx = hip.Context()
cq = hip.CommandQueue(ctx)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.N_SPHERES
plan = sys.argv[3] if len(sys.argv) > 3 else "auto"
scene = sys.argv[4] if len(sys.argv) > 4 else "uniform"
if scene == "config3":
    coords, radii = bench.clustered_scene(n, 0.0152)
    cap = 1 << 25
else:
    coords, radii = bench.uniform_scene(n)
    radii[:] = bench.RADIUS * (1e6 / n) ** (1.0 / 3.0)
    cap = bench.PAIR_CAPACITY * 8
dtype = os.environ.get("COLLISION_PATH_DTYPE", "float32")
cb, rb = hip.Buffer(ctx, hostbuf=coords.astype(dtype)), hip.Buffer(ctx, hostbuf=radii.astype(dtype))
nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
if len(sys.argv) > 5:
    import ctypes
    from collision_amd._lib import cdll
    lib = cdll()
    lib.col_debug_leaf_blocks.argtypes = [ctypes.c_float]
    lib.col_debug_leaf_blocks(ctypes.c_float(float(sys.argv[5])))
    if len(sys.argv) > 6:
        lib.col_debug_traverse(int(sys.argv[6]))
if os.environ.get("COLLISION_LBVH_MODE"):      # col_debug_lbvh (8192: Karras' searches instead of the climb)
    from collision_amd._lib import cdll as _cdll
    _cdll().col_debug_lbvh(int(os.environ["COLLISION_LBVH_MODE"]))
col = Collider(ctx, n, bench.NGROUPS, bench.GROUP_SIZE, coord_dtype=dtype)
col.sort_plan = plan
for _ in range(steps):
    col.get_collisions(cq, cb, rb, nb, pb, cap)
cq.finish()
print("pairs", int(hip.read_buffer(cq, nb, "uint32", 1)[0]))

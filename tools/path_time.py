"""Whole-path and col_lbvh time at the given sizes (uniform scene, contacts per sphere of config 2) and config 3 -- one
line per size; run it under COLLISION_AMD_LIB=<another build> to compare two builds on one box (tools/ab_builds.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
tag = os.environ.get("COLLISION_AMD_LIB", "tree")[-28:]
DT = np.dtype(os.environ.get("COLLISION_PATH_DTYPE", "float32"))       # float64: the same scenes in double precision
for n in [int(a) for a in sys.argv[1:]] or [1000000, 2000000, 16000000]:
    coords, radii = bench.uniform_scene(n)
    radii[:] = bench.RADIUS * (1e6 / n) ** (1.0 / 3.0)
    coords, radii = coords.astype(DT), radii.astype(DT)
    cap = max(1 << 17, n // 8)
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
    col = Collider(ctx, n, 64, 256, coord_dtype=DT)
    scratch = hip.Buffer(ctx, call.col_lbvh_scratch_bytes(n, DT.itemsize))
    def lbvh():
        call.col_lbvh(cq.stream, col._codes_bufs[1].ptr, col._ids_bufs[1].ptr, cb.ptr, rb.ptr, col._nodes_buf.ptr,
                      col._bounds_buf.ptr, scratch.ptr, n, DT.itemsize)
    def path():
        col.get_collisions(cq, cb, rb, nb, pb, cap)
    reps = 20 if n <= 4000000 else 6
    out = []
    for f in (path, lbvh):
        ts = []
        for _ in range(3):
            for _ in range(3):
                f()
            cq.finish()
            ts.append(bench.time_events(hip, cq, f, reps))
        out.append(min(ts))
    pairs = int(hip.read_buffer(cq, nb, np.uint32, 1)[0])
    print("%-28s n = %9d: path %.4f ms  lbvh %.4f ms  pairs %d" % (tag, n, out[0], out[1], pairs), flush=True)
    del col, cb, rb, nb, pb, scratch
if DT != np.float32 or os.environ.get("COLLISION_PATH_NO_C3"):
    sys.exit(0)
c3 = bench.config3_leg(hip, ctx, cq)
print("%-28s config 3: %.4f ms  pairs %d" % (tag, c3["ms_per_step"], c3["pairs"]), flush=True)

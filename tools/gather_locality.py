"""How much of k_chunk is the row gather?  The same scene twice: spheres in arrival (random) order, and the same spheres
stored in Morton order (the collider's own sorted ids applied on the host), where ids[p] ~ p and the gather of leaf rows is
coalesced.  Same tree, same boxes, same pairs (ids renamed); col_lbvh and whole path, uniform scene."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
for n in [int(a) for a in sys.argv[1:]] or [1000000, 2000000, 16000000]:
    coords, radii = bench.uniform_scene(n)
    radii[:] = bench.RADIUS * (1e6 / n) ** (1.0 / 3.0)
    cap = max(1 << 17, n // 8)
    order = None
    for what in ("arrival order", "Morton order "):
        if order is not None:
            coords, radii = np.ascontiguousarray(coords[order]), np.ascontiguousarray(radii[order])
        cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
        nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
        col = Collider(ctx, n, 64, 256)
        scratch = hip.Buffer(ctx, call.col_lbvh_scratch_bytes(n, 4))
        def lbvh():
            call.col_lbvh(cq.stream, col._codes_bufs[1].ptr, col._ids_bufs[1].ptr, cb.ptr, rb.ptr, col._nodes_buf.ptr,
                          col._bounds_buf.ptr, scratch.ptr, n, 4)
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
        print("n = %9d  %s: path %.4f ms  col_lbvh %.4f ms  pairs %d" % (n, what, out[0], out[1], pairs), flush=True)
        if order is None:
            order = hip.read_buffer(cq, col._ids_bufs[1], np.uint32, n).astype(np.int64)
        del col, cb, rb, nb, pb, scratch

"""COLLISION_LBVH_AB=<mode> (default 1024) names the col_debug_lbvh mode compared with production (8192: Karras' searches
for every node instead of the climb).  Originally: k_chunk's prefix / suffix unions on the DPP network (production, f32) against the ds_bpermute shuffles
(col_debug_lbvh(1024)): col_lbvh alone and the whole path, uniform scene, interleaved rounds in one process;
the node boxes of the two variants are compared byte for byte."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
ALT = int(os.environ.get('COLLISION_LBVH_AB', '1024'))
for n in [int(a) for a in sys.argv[1:]] or [1000000, 2000000, 16000000]:
    coords, radii = bench.uniform_scene(n)
    radii[:] = bench.RADIUS * (1e6 / n) ** (1.0 / 3.0)
    cap = max(1 << 17, n // 8)
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
    col = Collider(ctx, n, 64, 256)
    col.get_collisions(cq, cb, rb, nb, pb, cap); cq.finish()
    scratch = hip.Buffer(ctx, call.col_lbvh_scratch_bytes(n, 4))
    def lbvh():
        call.col_lbvh(cq.stream, col._codes_bufs[1].ptr, col._ids_bufs[1].ptr, cb.ptr, rb.ptr, col._nodes_buf.ptr,
                      col._bounds_buf.ptr, scratch.ptr, n, 4)
    def path():
        col.get_collisions(cq, cb, rb, nb, pb, cap)
    res, boxes = {}, {}
    reps = 20 if n <= 4000000 else 6
    for rnd in range(3):
        for mode in (ALT, 0):
            cdll().col_debug_lbvh(mode)
            for f, key in ((lbvh, "lbvh"), (path, "path")):
                for _ in range(3):
                    f()
                cq.finish()
                res.setdefault((mode, key), []).append(bench.time_events(hip, cq, f, reps))
            if rnd == 0:
                lbvh(); cq.finish()
                boxes[mode] = hip.read_buffer(cq, col._bounds_buf, np.uint32, (2 * n - 1) * 8).copy()
    cdll().col_debug_lbvh(0)
    same = bool((boxes[0] == boxes[ALT]).all())
    for key in ("lbvh", "path"):
        print(("n = %9d %s: mode " + str(ALT) + " %s | production %s ms   same boxes: %s") % (
            n, key, " ".join("%.4f" % v for v in res[(ALT, key)]), " ".join("%.4f" % v for v in res[(0, key)]), same), flush=True)

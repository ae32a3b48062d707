import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
coords, radii = bench.uniform_scene(n)
radii[:] = bench.RADIUS * (1e6 / n) ** (1.0 / 3.0)
cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, (1 << 20) * 8)
col = Collider(ctx, n, 64, 256)
col.get_collisions(cq, cb, rb, nb, pb, 1 << 20); cq.finish()
scratch = hip.Buffer(ctx, call.col_lbvh_scratch_bytes(n, 4))
def run():
    call.col_lbvh(cq.stream, col._codes_bufs[1].ptr, col._ids_bufs[1].ptr, cb.ptr, rb.ptr, col._nodes_buf.ptr,
                  col._bounds_buf.ptr, scratch.ptr, n, 4)
for mode, what in ((0, "full"), (1, "no gather (coords[p])"), (2, "no parent stores"), (4, "leaves only"),
                   (8, "no box wait / record store"), (3, "no gather, no parent"), (32, "chunk = blockIdx (plain order)"), (0, "full"),
                   (32, "chunk = blockIdx (plain order)")):
    cdll().col_debug_lbvh(mode)
    run(); cq.finish()
    print("mode %d %-28s %.4f ms" % (mode, what, bench.time_events(hip, cq, run, 20)))
cdll().col_debug_lbvh(0)

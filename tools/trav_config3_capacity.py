import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from collision_amd import hip
from collision_amd._lib import call, cdll
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1000000
coords, radii = bench.clustered_scene(n, 0.0152)
cap = 1 << 25
pb = hip.Buffer(ctx, cap * 8); nb = hip.Buffer(ctx, 4)
cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
col = Collider(ctx, n, 64, 256)
col.get_collisions(cq, cb, rb, nb, pb, cap); cq.finish()
z = np.zeros(1, np.uint32)
for c in (cap, 0, 1 << 20, cap, 0):
    def run():
        call.col_fill(cq.stream, nb.ptr, z.ctypes.data, 4, 1)
        call.col_traverse(cq.stream, pb.ptr if c else None, nb.ptr, c, None, col._bounds_buf.ptr, n, 4)
    run(); cq.finish()
    ms = bench.time_events(hip, cq, run, 10)
    print("config3 traverse capacity %d: %.4f ms pairs %d" % (c, ms, hip.read_buffer(cq, nb, np.uint32, 1)[0]))

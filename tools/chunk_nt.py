"""k_chunk's row gather as a non-temporal load (col_debug_lbvh(16)) against the same diagnostics instance without it (32: a no-op bit)
and the production instance (0): whole path at 1 M / 2 M / 16 M uniform spheres."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import cdll
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
for n in (1000000, 2000000, 16000000):
    coords, radii = bench.uniform_scene(n)
    radii[:] = bench.RADIUS * (1e6 / n) ** (1.0 / 3.0)
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, (1 << 20) * 8)
    col = Collider(ctx, n, 64, 256)
    def step():
        col.get_collisions(cq, cb, rb, nb, pb, 1 << 20)
    for rnd in range(2):
        for mode in (0, 32, 16):
            cdll().col_debug_lbvh(mode)
            for _ in range(3): step()
            cq.finish()
            ms = bench.time_events(hip, cq, step, 10 if n <= 2000000 else 4)
            print("n %8d round %d mode %2d: %.4f ms  pairs %d" % (n, rnd, mode, ms, int(hip.read_buffer(cq, nb, np.uint32, 1)[0])))
    cdll().col_debug_lbvh(0)
    del col, cb, rb, nb, pb

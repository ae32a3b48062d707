import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1000000

def clustered(n, sigma, r, seed=4):
    rng = np.random.RandomState(seed)
    centres = rng.uniform(0.2, 0.8, size=(8, 3))
    pts = np.concatenate([rng.normal(c, sigma, size=(n // 8, 3)) for c in centres])
    c = np.zeros((n, 4), np.float32); c[:, :3] = pts
    return c, np.full(n, r, np.float32)

scenes = [("uniform r=0.001",) + bench.uniform_scene(n)]
for sigma in (0.05, 0.02, 0.0152):
    scenes.append(("clustered sigma=%g r=0.001" % sigma,) + clustered(n, sigma, 0.001))
cap = 1 << 26
pb = hip.Buffer(ctx, cap * 8)
nb = hip.Buffer(ctx, 4)
z = np.zeros(1, np.uint32)
for name, coords, radii in scenes:
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    col = Collider(ctx, n, 64, 256)
    col.get_collisions(cq, cb, rb, nb, pb, cap); cq.finish()
    for label, variant in (("asm walk (production)", 0), ("generic loop", 64), ("asm walk (production)", 0), ("generic loop", 64)):
        cdll().col_debug_traverse(variant)
        def run():
            call.col_fill(cq.stream, nb.ptr, z.ctypes.data, 4, 1)
            call.col_traverse(cq.stream, pb.ptr, nb.ptr, cap, None, col._bounds_buf.ptr, n, 4)
        run(); cq.finish()
        ms = bench.time_events(hip, cq, run, 10)
        print("%-30s %-22s: %.4f ms, pairs %d" % (name, label, ms, hip.read_buffer(cq, nb, np.uint32, 1)[0]))
cdll().col_debug_traverse(0)

"""Whole-path time against the number of spheres (uniform scene, contacts/sphere held constant), f32 and f64."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
for dtype in ("float32", "float64"):
    for n in (10000, 100000, 500000, 1000000, 2000000, 4000000, 16000000):
        if dtype == "float64" and n > 4000000:
            continue
        rng = np.random.RandomState(4)
        coords = np.zeros((n, 4), dtype)
        coords[:, :3] = rng.random_sample((n, 3))
        r = 0.001 * (1e6 / n) ** (1.0 / 3.0)
        radii = np.full(n, r, dtype)
        cap = max(1 << 17, n // 8)
        cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
        nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
        col = Collider(ctx, n, 64, 256, dtype)
        def step():
            col.get_collisions(cq, cb, rb, nb, pb, cap)
        for _ in range(5):
            step()
        cq.finish()
        ms = bench.time_events(hip, cq, step, 20 if n <= 4000000 else 5)
        pairs = int(hip.read_buffer(cq, nb, np.uint32, 1)[0])
        print("%s n = %9d: %.4f ms  %8.1f M spheres/s  pairs %d" % (dtype, n, ms, n / ms / 1e3, pairs))
        del col, cb, rb, nb, pb

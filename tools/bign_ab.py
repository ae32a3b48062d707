"""A/B: the 4096-pair against the 8192-pair sort tile (and its fused 8192-code Morton kernel) inside the whole path at
8.5 .. 20 M spheres: col_debug_radix_tile(16 << 20) = the round-2 threshold (4096 up to 16 Mi), (8 << 20) = round 3.
Uniform scene, contacts per sphere of config 2; interleaved rounds in one process, a new Collider per setting."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
sizes = [int(a) for a in sys.argv[1:]] or [8500000, 12000000, 16000000, 20000000]
for n in sizes:
    rng = np.random.RandomState(4)
    coords = np.zeros((n, 4), np.float32)
    coords[:, :3] = rng.random_sample((n, 3))
    radii = np.full(n, 0.001 * (1e6 / n) ** (1.0 / 3.0), np.float32)
    cap = n // 8
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
    res, pairs = {}, {}
    for rnd in range(3):
        for big in (16 << 20, 8 << 20):
            call.col_debug_radix_tile(big)
            col = Collider(ctx, n, 64, 256, "float32")
            def step():
                col.get_collisions(cq, cb, rb, nb, pb, cap)
            for _ in range(3):
                step()
            cq.finish()
            res.setdefault(big, []).append(bench.time_events(hip, cq, step, 8))
            pairs[big] = int(hip.read_buffer(cq, nb, np.uint32, 1)[0])
            del col
    print("n = %9d  (tile %d -> %d): " % (n, call.col_radix_tile(n, 4, 4), 8192) +
          "   ".join("BIG_N %2d Mi: %s ms, pairs %d" % (b >> 20, " ".join("%.4f" % v for v in t), pairs[b]) for b, t in res.items()), flush=True)
call.col_debug_radix_tile(8 << 20)

import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
def scenes():
    yield "uniform 1M", bench.uniform_scene(1000000)
    yield "config3 1M", bench.clustered_scene(1000000, 0.0152)
    for m in (2000000, 4000000):
        c, r = bench.uniform_scene(m)
        r[:] = bench.RADIUS * (1e6 / m) ** (1.0 / 3.0)
        yield "uniform %dM" % (m // 1000000), (c, r)
for name, (coords, radii) in scenes():
    n = len(coords)
    cap = 1 << 25
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
    for plan in ("lsd", "msd", "auto"):
        col = Collider(ctx, n, 64, 256)
        col.sort_plan = plan
        def step():
            col.get_collisions(cq, cb, rb, nb, pb, cap)
        for _ in range(3):
            step()
        cq.finish()
        ms = bench.time_events(hip, cq, step, 10)
        print("%s plan %s: %.4f ms, pairs %d" % (name, plan, ms, hip.read_buffer(cq, nb, np.uint32, 1)[0]))

"""The reference's count-only benchmark shape (tests/test_collide.py: 307 200 spheres in [-1, 1]^3, radii U(0.006, 0.06),
n_collisions = 0) and config 3 in count-only mode: one line per scene; COLLISION_AMD_LIB=<another build> for A/Bs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
tag = os.environ.get("COLLISION_AMD_LIB", "tree")[-28:]
rng = np.random.RandomState(7)
scenes = []
size = 307200
coords = np.zeros((size, 4), np.float32); coords[:, :3] = rng.uniform(-1, 1, size=(size, 3))
scenes.append(("reference shape 307200", coords, rng.uniform(0.006, 0.06, size=size).astype(np.float32), 128))
c3, r3 = bench.clustered_scene(1000000, 0.0152)
scenes.append(("config 3, count only", c3, r3, 256))
c2, r2 = bench.uniform_scene(1000000)
scenes.append(("config 2, count only", c2, r2, 256))
for name, coords, radii, gs in scenes:
    cb, rb, nb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii), hip.Buffer(ctx, 4)
    col = Collider(ctx, len(coords), 8, gs)
    col.traverse_plan = os.environ.get("COLLISION_TRAVERSE_PLAN", "auto")      # exact | chunked | auto
    f = lambda: col.get_collisions(cq, cb, rb, nb, None, 0)
    ts = []
    for _ in range(3):
        for _ in range(4):
            f()
        cq.finish()
        ts.append(bench.time_events(hip, cq, f, 10))
    print("%-28s %-24s %.4f ms  pairs %d" % (tag, name, min(ts), int(hip.read_buffer(cq, nb, np.uint32, 1)[0])), flush=True)

"""Whole path with the traversal's exact / chunked pair allocation: BASELINE config 3 (25.4 M pairs), a medium scene and config 2."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1000000
scenes = [("config3 (25.4 M pairs)", bench.clustered_scene(n, 0.0152), 1 << 25),
          ("clustered sigma 0.03 (~3 M pairs)", bench.clustered_scene(n, 0.03), 1 << 24),
          ("config2 uniform", bench.uniform_scene(n), 1 << 23)]
for name, (coords, radii), cap in scenes:
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
    col = Collider(ctx, n, 64, 256)
    if "config2" not in name:
        col.sort_plan = "lsd"
    for rnd in range(2):
        for plan in ("exact", "chunked", "auto"):
            col.traverse_plan = plan
            def step():
                col.get_collisions(cq, cb, rb, nb, pb, cap)
            for _ in range(4): step()
            cq.finish()
            ms = bench.time_events(hip, cq, step, 10)
            print("%-34s %-8s %.4f ms  pairs %d" % (name, plan, ms, int(hip.read_buffer(cq, nb, np.uint32, 1)[0])))

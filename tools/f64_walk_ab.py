"""Whole-path A/B of the float64 asm walk (col_debug_traverse(64) forces the generic loop) and of the float64 DPP scans of
k_chunk (col_debug_lbvh(2048) keeps the shuffle scans): python tools/f64_walk_ab.py [n ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import cdll
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
for n in [int(a) for a in sys.argv[1:]] or [1000000]:
    for name, (coords, radii), cap in (("uniform", bench.uniform_scene(n), 1 << 20), ("config3", bench.clustered_scene(n, 0.0152), 1 << 25)):
        radii = radii * (1e6 / n) ** (1.0 / 3.0) if name == "uniform" else radii
        for dt in ("float64", "float32"):
            cb, rb = hip.Buffer(ctx, hostbuf=coords.astype(dt)), hip.Buffer(ctx, hostbuf=radii.astype(dt))
            nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
            col = Collider(ctx, n, 64, 256, coord_dtype=dt)
            ref = None
            for variant in ((64, 0, 2048, 0, 2048, 0) if dt == "float64" else (0,)):
                cdll().col_debug_traverse(64 if variant == 64 else 0)
                cdll().col_debug_lbvh(2048 if variant == 2048 else 0)
                def step():
                    col.get_collisions(cq, cb, rb, nb, pb, cap)
                for _ in range(4): step()
                cq.finish()
                cnt = int(hip.read_buffer(cq, nb, np.uint32, 1)[0])
                pairs = np.sort(hip.read_buffer(cq, pb, np.uint64, min(cnt, cap)))
                if ref is None: ref = pairs
                same = pairs.shape == ref.shape and bool((pairs == ref).all())
                ms = bench.time_events(hip, cq, step, 20)
                print("n %9d %-8s %s %s: %.4f ms, pairs %d, same set: %s" % (n, name, dt, {64: "generic walk ", 2048: "shuffle scans", 0: "as shipped   "}[variant], ms, cnt, same), flush=True)
            cdll().col_debug_traverse(0)
            cdll().col_debug_lbvh(0)
            del cb, rb, nb, pb, col

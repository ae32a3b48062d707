"""Whole-path A/B of col_debug_traverse variants, interleaved on one box: python tools/order_ab.py [n ...] [vVARIANT ...]
(default: bit 21 = 2097152, no walk order, against production; bit 22 = 4194304: the first batch drawn from the counter too)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import cdll
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
VARIANTS = [int(a[1:]) for a in sys.argv[1:] if a.startswith('v')] or [2097152, 0]
NAMES = {2097152: 'natural order', 0: 'production', 4194304: 'first batch drawn', 6291456: 'natural, first batch drawn', 8388608: 'long walks at ordinary priority'}
for n in [int(a) for a in sys.argv[1:] if not a.startswith('v')] or [1000000, 2000000]:
    scenes = [("uniform",) + tuple(bench.uniform_scene(n)) + (1 << 20,)]
    scenes[0][2][:] = bench.RADIUS * (1e6 / n) ** (1.0 / 3.0)
    if n <= 2000000:
        scenes.append(("config3",) + tuple(bench.clustered_scene(n, 0.0152)) + (1 << 27 if n > 1000000 else 1 << 25,))
    for name, coords, radii, cap in scenes:
        cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
        nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
        col = Collider(ctx, n, 64, 256)
        ref = None
        for variant in VARIANTS * 3:
            cdll().col_debug_traverse(variant)
            def step():
                col.get_collisions(cq, cb, rb, nb, pb, cap)
            for _ in range(6): step()
            cq.finish()
            cnt = int(hip.read_buffer(cq, nb, np.uint32, 1)[0])
            pairs = np.sort(hip.read_buffer(cq, pb, np.uint64, min(cnt, cap)))
            if ref is None: ref = pairs
            same = pairs.shape == ref.shape and bool((pairs == ref).all())
            ms = bench.time_events(hip, cq, step, 30)
            print("n %9d %-8s %s: %.4f ms, pairs %d, same set: %s" % (n, name, NAMES.get(variant, str(variant)), ms, cnt, same), flush=True)
        cdll().col_debug_traverse(0)
        del cb, rb, nb, pb, col

"""A/B of col_traverse under col_debug_traverse variants: same pair set as variant 0?  time?  With bit 12 (4096, the
profiling instance) also the time every packet spends in its phases and what other schedules of the same packets would take.
    python tools/walk_ab.py [n ...] [vVARIANT ...]        e.g.  1000000 v0 v8192 v4096 v12288
variants: 0 production (static packet order), 8192 dynamic packet order, 256 / 512 skip phase 1 / 2, 4 half the waves"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
from collision_amd.collision import Collider
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
sizes = [int(a) for a in sys.argv[1:] if not a.startswith('v')] or [1000000]
VARIANTS = [int(a[1:]) for a in sys.argv[1:] if a.startswith('v')] or [0, 8192, 0, 8192]
z = np.zeros(1, np.uint32)
for n in sizes:
    scenes = [("uniform",) + tuple(bench.uniform_scene(n)) + (1 << 20,)]
    scenes[0][2][:] = bench.RADIUS * (1e6 / n) ** (1.0 / 3.0)
    if n <= 2000000:
        scenes.append(("config3",) + tuple(bench.clustered_scene(n, 0.0152)) + (1 << 25,))
    for name, coords, radii, cap in scenes:
        cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
        nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
        col = Collider(ctx, n, 64, 256)
        col.traverse_plan = "exact"
        col.get_collisions(cq, cb, rb, nb, pb, cap); cq.finish()
        ref = None
        for variant in VARIANTS:
            cdll().col_debug_traverse(variant)
            def run():
                call.col_fill(cq.stream, nb.ptr, z.ctypes.data, 4, 1)
                call.col_traverse(cq.stream, pb.ptr, nb.ptr, cap, None, col._bounds_buf.ptr, n, 4)
            run(); cq.finish()
            cnt = int(hip.read_buffer(cq, nb, np.uint32, 1)[0])
            pairs = np.sort(hip.read_buffer(cq, pb, np.uint64, min(cnt, cap)))
            if ref is None: ref = pairs
            same = pairs.shape == ref.shape and bool((pairs == ref).all())
            ms = bench.time_events(hip, cq, run, 10)
            print("n %9d %-8s variant %4d: %.4f ms, pairs %d, same set as variant 0: %s" % (n, name, variant, ms, cnt, same), flush=True)
            if variant & 4096:
                np_ = (n + 63) // 64
                out = np.zeros((np_, 4), np.uint32)
                run(); cq.finish()
                cdll().col_debug_walk_profile(out.ctypes.data, np_)
                t = out.astype(np.float64) * 0.01
                start = (out[:, 3] - out[:, 3].min()).astype(np.float64) * 0.01
                end = start + t[:, 0] + t[:, 1] + t[:, 2]
                print("    per packet (us): own records %.2f, phase 1 %.2f, phase 2 %.2f (p10 %.2f p50 %.2f p90 %.2f max %.2f); "
                      "packet starts: p50 %.1f max %.1f, last end %.1f" % (t[:, 2].mean(), t[:, 0].mean(), t[:, 1].mean(),
                      *np.percentile(t[:, 1], [10, 50, 90, 100]), np.percentile(start, 50), start.max(), end.max()), flush=True)
                np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "walk_prof_%s_%d_v%d.npy" % (name, n, variant)), out)
                # what would other schedules of the same packets take?  (list scheduling of the measured durations)
                import heapq
                d = t[:, 0] + t[:, 1] + t[:, 2]
                def dynamic(order_by_xcd, slots):
                    worst = 0.0
                    for pk in order_by_xcd:
                        h = [0.0] * slots
                        for x in pk:
                            heapq.heapreplace(h, h[0] + d[x])
                        worst = max(worst, max(h))
                    return worst
                ngroups = (np_ + 15) // 16
                xcd_groups = [range(ngroups * x // 8, ngroups * (x + 1) // 8) for x in range(8)]
                per_xcd = [[g * 16 + w for g in gr for w in range(16) if g * 16 + w < np_] for gr in xcd_groups]
                print("    sum of packet times / 8192 slots %.1f us; dynamic per XCD (1024 slots each, packets in order) %.1f us; "
                      "the same, longest first %.1f us" % (d.sum() / 8192, dynamic(per_xcd, 1024),
                      dynamic([sorted(pk, key=lambda x: -d[x]) for pk in per_xcd], 1024)), flush=True)
        cdll().col_debug_traverse(0)
        del cb, rb, nb, pb, col

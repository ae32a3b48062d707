"""A/B of the scatter pass at 64 Mi pairs between tile classes (col_debug_radix_tile), interleaved rounds in one process:
8192 pairs (512 threads, two workgroups per CU: production) against the forced classes given on the command line
(6144 = 384 threads x 16, three workgroups per CU).  Each under modes 0 (production), 2 (coalesced output) and 16384
((key, value) pairs interleaved in one output array).  Also verifies that a whole sort with the forced class is correct.

    python tools/radix_tri_ab.py [rounds] [tile,tile,...]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1 << 26
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tiles = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8192, 6144]
rng = np.random.RandomState(4)
keys = rng.randint(0, 2 ** 30, size=n).astype(np.uint32)
kin, vin = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
kout, vout = hip.Buffer(ctx, n * 8), hip.Buffer(ctx, n * 4)      # (kout: room for mode 16384's interleaved pairs)
hists = {}
for t in tiles:
    call.col_debug_radix_tile(t)
    nb = -(-n // t)
    h = hip.Buffer(ctx, 256 * nb * 4)
    ss = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nb))
    call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, 0, h.ptr)
    call.col_scan_u32(cq.stream, h.ptr, 256 * nb, ss.ptr)
    cq.finish()
    hists[t] = h
modes = ((0, "production"), (2, "coalesced out"), (16384, "pairs out"))
times = {(t, m): [] for t in tiles for m, _ in modes}
for rnd in range(rounds):
    for t in tiles:
        call.col_debug_radix_tile(t)
        for m, _ in modes:
            cdll().col_debug_radix(m)
            def run():
                call.col_radix_scatter(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, 0, hists[t].ptr)
            for _ in range(3):
                run()
            cq.finish()
            times[(t, m)].append(bench.time_events(hip, cq, run, 10))
cdll().col_debug_radix(0)
for t in tiles:
    for m, what in modes:
        v = sorted(times[(t, m)])
        med = v[len(v) // 2]
        print("tile %5d mode %5d %-14s min %.4f median %.4f ms = %.3f of 8 TB/s   in order: %s" % (
            t, m, what, v[0], med, n * 16 / med / 1e6 / 8000, " ".join("%.3f" % x for x in times[(t, m)])))
# whole sorts: correctness with each forced class + time
want = None
for t in tiles:
    call.col_debug_radix_tile(t)
    scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
    def whole():
        call.col_radix_sort(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, scratch.ptr, 0)
    for _ in range(3):
        whole()
    cq.finish()
    ms = bench.time_events(hip, cq, whole, 10)
    got_k = hip.read_buffer(cq, kout, np.uint32, n)
    got_v = hip.read_buffer(cq, vout, np.uint32, n)
    if want is None:
        order = np.argsort(keys, kind="stable")
        want = (keys[order], order.astype(np.uint32))
    ok = bool((got_k == want[0]).all() and (got_v == want[1]).all())
    print("tile %5d whole sort %.4f ms = %.2f Gkeys/s   correct: %s" % (t, ms, n / ms / 1e6, ok))
    del scratch
call.col_debug_radix_tile(0)

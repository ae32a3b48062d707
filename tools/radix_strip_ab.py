"""A/B of the scatter pass at 64 Mi pairs: XCD-contiguous tile ranges (production) against strips of G consecutive tiles
going round the XCDs (col_debug_radix 1 << 22 | log2(G) << 24), separate arrays out and (key, value) pairs out; interleaved
rounds in one process.    python tools/radix_strip_ab.py [rounds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1 << 26
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.RandomState(4)
keys = rng.randint(0, 2 ** 30, size=n).astype(np.uint32)
kin, vin = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
kout, vout = hip.Buffer(ctx, n * 8), hip.Buffer(ctx, n * 4)
kref, vref = hip.Buffer(ctx, n * 4), hip.Buffer(ctx, n * 4)
tile = call.col_radix_tile(n, 4, 4); nb = -(-n // tile)
hist = hip.Buffer(ctx, 256 * nb * 4)
ss = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nb))
call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, 0, hist.ptr)
call.col_scan_u32(cq.stream, hist.ptr, 256 * nb, ss.ptr)
call.col_radix_scatter(cq.stream, kin.ptr, kref.ptr, vin.ptr, vref.ptr, n, 4, 4, 0, hist.ptr)
cq.finish()
want_k, want_v = hip.read_buffer(cq, kref, np.uint32, n), hip.read_buffer(cq, vref, np.uint32, n)
S = 1 << 22
modes = [(0, "production"), (1 << 30, "diag instance, xcd ranges")]
for lg in (1, 2, 3, 4, 5, 6, 7):
    modes.append((S | (lg << 24), "strips of %d" % (1 << lg)))
modes.append((16384, "pairs out, xcd ranges"))
for lg in (2, 4, 6):
    modes.append((16384 | S | (lg << 24), "pairs out, strips of %d" % (1 << lg)))
def run():
    call.col_radix_scatter(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, 0, hist.ptr)
# correctness of the strip order (separate arrays)
for mode, what in modes:
    if mode & 16384:
        continue
    cdll().col_debug_radix(mode)
    call.col_fill(cq.stream, kout.ptr, None, 0, n * 4) if False else None
    run(); cq.finish()
    ok = bool((hip.read_buffer(cq, kout, np.uint32, n) == want_k).all() and (hip.read_buffer(cq, vout, np.uint32, n) == want_v).all())
    if not ok:
        print("!! mode %d (%s): output differs" % (mode, what))
times = {m: [] for m, _ in modes}
for rnd in range(rounds):
    for mode, _ in modes:
        cdll().col_debug_radix(mode)
        for _ in range(3):
            run()
        cq.finish()
        times[mode].append(bench.time_events(hip, cq, run, 10))
cdll().col_debug_radix(0)
for mode, what in modes:
    t = sorted(times[mode])
    print("%-28s min %.4f median %.4f ms = %.3f of 8 TB/s   in order: %s" % (
        what, t[0], t[len(t) // 2], n * 16 / t[len(t) // 2] / 1e6 / 8000, " ".join("%.3f" % v for v in times[mode])))

import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1 << 26
rng = np.random.RandomState(4)
keys = rng.randint(0, 2 ** 30, size=n).astype(np.uint32)
kin, vin = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
kout, vout = hip.Buffer(ctx, n * 4), hip.Buffer(ctx, n * 4)
tile = call.col_radix_tile(n, 4, 4); nb = -(-n // tile)
hist = hip.Buffer(ctx, 256 * nb * 4)
ss = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nb))
call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, 0, hist.ptr)
call.col_scan_u32(cq.stream, hist.ptr, 256 * nb, ss.ptr)
def copy():
    call.col_memcpy_d2d(cq.stream, kout.ptr, kin.ptr, n * 4); call.col_memcpy_d2d(cq.stream, vout.ptr, vin.ptr, n * 4)
copy(); cq.finish()
ms = bench.time_events(hip, cq, copy, 5)
print("hipMemcpy d2d keys+vals: %.4f ms  %.0f GB/s" % (ms, n * 16 / ms / 1e6))
modes = ((0, "full"), (2, "rank + coalesced write"), (4, "blockIdx tile order (no XCD remap)"), (64, "non-temporal loads"),
         (512, "+4.5 KB LDS per block"), (1024, "+18 KB LDS per block (1 block/CU at the 8192 tile)"))
times = {m: [] for m, _ in modes}
def run():
    call.col_radix_scatter(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, 0, hist.ptr)
for rnd in range(12):                      # interleaved rounds in one process (rule 24)
    for mode, _ in modes:
        cdll().col_debug_radix(mode)
        run(); cq.finish()
        times[mode].append(bench.time_events(hip, cq, run, 3))
for mode, what in modes:
    t = sorted(times[mode])
    print("mode %d %-26s min %.4f  median %.4f ms  -> %.0f GB/s (median)   in order: %s" % (mode, what, t[0], t[len(t) // 2], n * 16 / t[len(t) // 2] / 1e6, " ".join("%.3f" % v for v in times[mode])))
cdll().col_debug_radix(32)
call.col_debug_radix_stamps(None, 1)
run(); cq.finish()
st = np.zeros(8, np.uint64)
call.col_debug_radix_stamps(st.ctypes.data, 1)
tot = float(st[:5].sum())
print("k_scatter phase shares (thread 0 of every block, s_memtime):",
      {k: round(float(v) / tot, 3) for k, v in zip(("load+transpose", "rank", "digit scan", "LDS scatter", "readback+stores"), st[:5])},
      "cycles/block", round(tot / nb))
cdll().col_debug_radix(0)
def h():
    call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, 0, hist.ptr)
h(); cq.finish()
ms = bench.time_events(hip, cq, h, 5); print("hist %.4f ms %.0f GB/s" % (ms, n * 4 / ms / 1e6))

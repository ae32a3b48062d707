"""Interleaved A/B timing of the 64 Mi-pair scatter pass (config 5 geometry) under the col_debug_radix modes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1 << 26
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.RandomState(4)
keys = rng.randint(0, 2 ** 30, size=n).astype(np.uint32)
kin, vin = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
kout, vout = hip.Buffer(ctx, n * 8), hip.Buffer(ctx, n * 4)      # (kout: room for mode 16384's interleaved pairs)
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
modes = ((0, "production"), (1 << 30, "production (diag instance)"), (2, "coalesced write"), (4, "blockIdx tile order"),
         (64, "non-temporal loads"), (32768, "4 MiB output window (stores hit L2)"),
         (8192, "every run an aligned 128-byte cell of its own"), (16384, "(key, value) pairs interleaved in one output"), (1024, "one resident block per CU (+18 KB LDS)"),
         (1 << 23, "store phase at raised wave priority"), (1 << 27, "load phase at raised wave priority"),
         (1 << 28, "ranking at raised wave priority"), (1 << 29, "s_sleep between the store steps"))
if len(sys.argv) > 2:
    want = set(int(x) for x in sys.argv[2].split(","))
    modes = tuple(m for m in modes if m[0] in want)
times = {m: [] for m, _ in modes}
def run():
    call.col_radix_scatter(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, 0, hist.ptr)
for rnd in range(rounds):                      # interleaved rounds in one process (rule 24)
    for mode, _ in modes:
        cdll().col_debug_radix(mode)
        for _ in range(3):
            run()
        cq.finish()
        times[mode].append(bench.time_events(hip, cq, run, 10))
for mode, what in modes:
    t = sorted(times[mode])
    print("mode %4d %-36s min %.4f  median %.4f ms  -> %.0f GB/s = %.3f of 8 TB/s (median)   in order: %s" % (
        mode, what, t[0], t[len(t) // 2], n * 16 / t[len(t) // 2] / 1e6, n * 16 / t[len(t) // 2] / 1e6 / 8000,
        " ".join("%.3f" % v for v in times[mode])))
cdll().col_debug_radix(0)
# keys only (its own histogram: from 32 Mi keys a key-only sort uses the 16384-key tile)
nb0 = -(-n // call.col_radix_tile(n, 4, 0))
hist0 = hip.Buffer(ctx, 256 * nb0 * 4)
call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 0, 0, hist0.ptr)
call.col_scan_u32(cq.stream, hist0.ptr, 256 * nb0, ss.ptr)
def run0():
    call.col_radix_scatter(cq.stream, kin.ptr, kout.ptr, None, None, n, 4, 0, 0, hist0.ptr)
for mode in (0,):
    cdll().col_debug_radix(mode)
    run0(); cq.finish()
    ms = bench.time_events(hip, cq, run0, 10)
    print("keys only, mode %d: %.4f ms -> %.0f GB/s" % (mode, ms, n * 8 / ms / 1e6))
cdll().col_debug_radix(0)
def h():
    call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, 0, hist.ptr)
h(); cq.finish()
ms = bench.time_events(hip, cq, h, 5); print("hist %.4f ms %.0f GB/s" % (ms, n * 4 / ms / 1e6))
def whole():
    call.col_radix_sort(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, scratch.ptr, 0)
scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
for mode in (0,):
    cdll().col_debug_radix(mode)
    for _ in range(5): whole()
    cq.finish()
    ms = bench.time_events(hip, cq, whole, 10)
    print("whole sort, mode %d: %.4f ms -> %.2f Gkeys/s" % (mode, ms, n / ms / 1e6))
cdll().col_debug_radix(0)

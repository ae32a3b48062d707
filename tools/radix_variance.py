"""Scatter-pass time over consecutive launches (diagnostic): does it drift, and with what?"""
import os, sys, time
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
def run():
    call.col_radix_scatter(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, 0, hist.ptr)
def series(label, mode, rounds, reps=3, pause=0.0):
    cdll().col_debug_radix(mode)
    out = []
    for _ in range(rounds):
        out.append(bench.time_events(hip, cq, run, reps))
        if pause: time.sleep(pause)
    print("%-34s %s" % (label, " ".join("%.3f" % v for v in out)))
run(); cq.finish()
series("mode 0 x30 back to back", 0, 30)
series("mode 0 x10, 0.2 s pauses", 0, 10, pause=0.2)
series("mode 0 x15 back to back", 0, 15)
series("mode 2 x10 (coalesced out)", 2, 10)
series("mode 0 x10 after mode 2", 0, 10)
cdll().col_debug_radix(0)

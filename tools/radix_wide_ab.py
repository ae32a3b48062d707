"""A/B: the 8192-pair scatter tile as 512 threads x 16 items (production) against 1024 threads x 8 items
(col_debug_radix_tile(8193)): 64 Mi (u32 key, u32 value) pairs, pass 0 and pass 3, interleaved series in one process."""
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
outs = {v: (hip.Buffer(ctx, n * 4), hip.Buffer(ctx, n * 4)) for v in (8194, 8193)}
tile = call.col_radix_tile(n, 4, 4); nb = -(-n // tile)
hist = hip.Buffer(ctx, 256 * nb * 4); ss = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nb))
for rpass in (0, 3):
    call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, rpass, hist.ptr)
    call.col_scan_u32(cq.stream, hist.ptr, 256 * nb, ss.ptr)
    def run(v):
        def f():
            call.col_radix_scatter(cq.stream, kin.ptr, outs[v][0].ptr, vin.ptr, outs[v][1].ptr, n, 4, 4, rpass, hist.ptr)
        return f
    for rnd in range(3):
        for v in (8194, 8193):
            cdll().col_debug_radix_tile(v)
            f = run(v)
            for _ in range(30): f()
            cq.finish()
            each = bench.time_events_each(hip, cq, f, 100)
            print("pass %d round %d  %s: median %.4f ms  p10 %.4f p90 %.4f  -> %.3f of 8 TB/s" %
                  (rpass, rnd, "512x16 " if v == 8194 else "1024x8 ", each[50], each[10], each[90], n * 16 / each[50] / 1e6 / 8000))
    a = [hip.read_buffer(cq, b, np.uint32, n) for b in outs[8194]]
    b = [hip.read_buffer(cq, b, np.uint32, n) for b in outs[8193]]
    print("pass %d outputs equal: %s" % (rpass, bool((a[0] == b[0]).all() and (a[1] == b[1]).all())))
cdll().col_debug_radix_tile(8194)

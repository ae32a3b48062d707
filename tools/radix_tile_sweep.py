"""Whole-sort time of (u32 key, u32 id) pairs against n for each tile class (col_debug_radix_tile),
interleaved rounds in one process.  Decides SMALL_N / BIG_N in csrc/radix.hip."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
rng = np.random.RandomState(4)
nmax = 1 << 26
keys = rng.randint(0, 2 ** 30, size=nmax).astype(np.uint32)
kin, vin = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, hostbuf=np.arange(nmax, dtype=np.uint32))
kout, vout = hip.Buffer(ctx, nmax * 4), hip.Buffer(ctx, nmax * 4)
call.col_debug_radix_tile(1024)
scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(nmax, 4, 4))      # sized for the smallest tile
ref = None
sizes = [int(a) for a in sys.argv[1:]] or [1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26]
for n in sizes:
    res = {}
    for rnd in range(6 if n < (1 << 22) else 4):
        for tile in (1024, 4096, 8192):
            call.col_debug_radix_tile(tile)
            def whole():
                call.col_radix_sort(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, scratch.ptr, 0)
            for _ in range(min(50, max(2, (1 << 24) // n))):
                whole()
            cq.finish()
            res.setdefault(tile, []).append(bench.time_events(hip, cq, whole, min(100, max(3, (1 << 25) // n))))
            if rnd == 0:
                out = hip.read_buffer(cq, kout, np.uint32, min(n, 1 << 20))
                assert (np.diff(out.astype(np.int64)) >= 0).all()
    print("n = %9d: " % n + "   ".join("tile %d: %.4f ms (%.1f Gkeys/s)" % (t, min(v), n / min(v) / 1e6) for t, v in res.items()))
call.col_debug_radix_tile(0)

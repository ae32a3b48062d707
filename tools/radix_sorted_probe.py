"""Per-pass scatter / histogram time on already-sorted keys (arange) for the 4096- and 8192-pair tiles."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1 << 26
kinds = {"arange": np.arange(n, dtype=np.uint32),
         "random30": np.random.RandomState(4).randint(0, 2 ** 30, size=n).astype(np.uint32)}
vin = hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
kout, vout = hip.Buffer(ctx, n * 4), hip.Buffer(ctx, n * 4)
for name, keys in kinds.items():
    kin = hip.Buffer(ctx, hostbuf=keys)
    for tile in (4096, 8192):
        call.col_debug_radix_tile(tile)
        nb = -(-n // tile)
        hist = hip.Buffer(ctx, 256 * nb * 4)
        ss = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nb))
        row = []
        for rpass in range(4):
            def h():
                call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, rpass, hist.ptr)
            def sc():
                call.col_radix_scatter(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, rpass, hist.ptr)
            for _ in range(30):
                h()
            th = bench.time_events(hip, cq, h, 10)
            h(); call.col_scan_u32(cq.stream, hist.ptr, 256 * nb, ss.ptr)
            for _ in range(60):
                sc()
            ts = bench.time_events(hip, cq, sc, 20)
            row.append("p%d hist %.3f scat %.3f" % (rpass, th, ts))
        print("%-9s tile %d: %s" % (name, tile, " | ".join(row)))
call.col_debug_radix_tile(0)

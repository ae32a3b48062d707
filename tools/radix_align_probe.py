"""What does the alignment of the digit runs cost the scatter pass?  Pass 0 of the 64 Mi-pair sort on keys whose per-tile digit
counts are (a) random (the production case), (b) exactly 32 each: every run is one aligned 128-byte line, (c) multiples of 16
(16 / 48): runs aligned to 64 bytes, (d) multiples of 8 (24 / 40): aligned to 32 bytes, (e) 31 / 33: runs of the right length at
odd offsets.  Same kernel, same bytes; only where the runs start differs.
    python tools/radix_align_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1 << 26
tile = call.col_radix_tile(n, 4, 4)
nblocks = n // tile
rng = np.random.RandomState(3)


def keys_with_counts(pattern):
    """every tile holds digit d pattern[d % len(pattern)] times (sum over 256 digits = tile), shuffled inside the tile"""
    counts = np.array([pattern[d % len(pattern)] for d in range(256)])
    assert counts.sum() == tile, counts.sum()
    one = np.repeat(np.arange(256, dtype=np.uint32), counts)
    out = np.empty(n, np.uint32)
    for t in range(0, nblocks, 256):                      # a few hundred distinct shuffles are plenty
        blk = rng.permutation(one)
        out[t * tile:(t + 256) * tile] = np.tile(blk, min(256, nblocks - t))
    return out | (rng.randint(0, 1 << 22, size=n).astype(np.uint32) << 8)


cases = [("random", rng.randint(0, 2 ** 30, size=n).astype(np.uint32)),
         ("32 each (128 B aligned)", keys_with_counts([32])),
         ("16 / 48 (64 B aligned)", keys_with_counts([16, 48])),
         ("24 / 40 (32 B aligned)", keys_with_counts([24, 40])),
         ("28 / 36 (16 B aligned)", keys_with_counts([28, 36])),
         ("30 / 34 (8 B aligned)", keys_with_counts([30, 34])),
         ("31 / 33 (odd offsets)", keys_with_counts([31, 33])),
         ("8 / 56 (32 B aligned, uneven)", keys_with_counts([8, 56])),
         ("12 / 52 (16 B aligned, uneven)", keys_with_counts([12, 52])),
         ("13 / 51 (odd, uneven)", keys_with_counts([13, 51]))]
vin = hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
kout, vout = hip.Buffer(ctx, n * 4), hip.Buffer(ctx, n * 4)
hist = hip.Buffer(ctx, 256 * nblocks * 4)
scan_scratch = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nblocks))
bufs = [(name, hip.Buffer(ctx, hostbuf=k)) for name, k in cases]
for rnd in range(2):
    for name, kin in bufs:
        call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, 0, hist.ptr)
        call.col_scan_u32(cq.stream, hist.ptr, 256 * nblocks, scan_scratch.ptr)

        def scatter():
            call.col_radix_scatter(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, 0, hist.ptr)
        for _ in range(30):
            scatter()
        cq.finish()
        each = bench.time_events_each(hip, cq, scatter, 100)
        ms = each[len(each) // 2]
        print("%-28s %.4f ms  = %.3f of 8 TB/s" % (name, ms, n * 16 / ms / 1e6 / 8000), flush=True)

"""Experiment: the scatter pass's ranking from one returning LDS atomic per item (col_debug_radix 1 << 21) against the
production match-any ranking.  (1) interleaved timing at 64 Mi (u32 key, u32 value) pairs, pass 0 and pass 3;
(2) is the result the same, i.e. does the LDS resolve same-address lanes in lane order (stability)?  Whole sorts of
several key distributions and sizes under the mode are compared with the production sort, ids included."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
PROD, NOOP, ATOM = 0, 1 << 20, 1 << 21
names = {PROD: "production      ", NOOP: "diag inst, no-op", ATOM: "LDS-atomic rank "}


def sort_pairs(keys, mode):
    n = len(keys)
    cdll().col_debug_radix(mode)
    kin, vin = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
    kout, vout = hip.Buffer(ctx, n * 4), hip.Buffer(ctx, n * 4)
    scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
    call.col_radix_sort(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, scratch.ptr, 0)
    cq.finish()
    cdll().col_debug_radix(0)
    return hip.read_buffer(cq, kout, np.uint32, n), hip.read_buffer(cq, vout, np.uint32, n)


rng = np.random.RandomState(7)
ok_all = True
cases = []
for n in (1000, 70001, 1 << 20, 5000003, (16 << 20) + 12345):
    cases += [("random32 n=%d" % n, rng.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)),
              ("4 digits n=%d" % n, (rng.randint(0, 4, size=n).astype(np.uint32) * np.uint32(0x01010101))),
              ("all equal n=%d" % n, np.full(n, 0xDEADBEEF, np.uint32)),
              ("sorted n=%d" % n, np.sort(rng.randint(0, 2 ** 30, size=n).astype(np.uint32))),
              ("runs of 64 n=%d" % n, np.repeat(rng.randint(0, 2 ** 32, size=n // 64 + 1, dtype=np.uint64).astype(np.uint32), 64)[:n])]
for what, keys in cases:
    a = sort_pairs(keys, PROD)
    b = sort_pairs(keys, ATOM)
    same = bool((a[0] == b[0]).all() and (a[1] == b[1]).all())
    ok_all &= same
    print("whole sort, %-28s atomic rank == production: %s" % (what, same), flush=True)
print("STABLE (every case identical): %s" % ok_all, flush=True)

n = 1 << 26
rng = np.random.RandomState(4)
keys = rng.randint(0, 2 ** 30, size=n).astype(np.uint32)
kin, vin = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
outs = {m: (hip.Buffer(ctx, n * 4), hip.Buffer(ctx, n * 4)) for m in (PROD, NOOP, ATOM)}
tile = call.col_radix_tile(n, 4, 4); nb = -(-n // tile)
hist = hip.Buffer(ctx, 256 * nb * 4); ss = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nb))
for rpass in (0, 3):
    call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, rpass, hist.ptr)
    call.col_scan_u32(cq.stream, hist.ptr, 256 * nb, ss.ptr)

    def run(m):
        def f():
            call.col_radix_scatter(cq.stream, kin.ptr, outs[m][0].ptr, vin.ptr, outs[m][1].ptr, n, 4, 4, rpass, hist.ptr)
        return f
    for rnd in range(3):
        for m in (PROD, NOOP, ATOM):
            cdll().col_debug_radix(m)
            f = run(m)
            for _ in range(30):
                f()
            cq.finish()
            each = bench.time_events_each(hip, cq, f, 100)
            print("pass %d round %d  %s: median %.4f ms  p10 %.4f p90 %.4f  -> %.3f of 8 TB/s" %
                  (rpass, rnd, names[m], each[50], each[10], each[90], n * 16 / each[50] / 1e6 / 8000), flush=True)
    cdll().col_debug_radix(0)
    a = [hip.read_buffer(cq, b, np.uint32, n) for b in outs[PROD]]
    b = [hip.read_buffer(cq, b, np.uint32, n) for b in outs[ATOM]]
    print("pass %d outputs equal: %s" % (rpass, bool((a[0] == b[0]).all() and (a[1] == b[1]).all())), flush=True)

"""k_hist and the whole sort at 64 Mi (30-bit keys, u32 values): one line; run it under COLLISION_AMD_LIB=<another build>
to compare two builds on one box (alternating processes, like tools/ab_builds.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
tag = os.environ.get("COLLISION_AMD_LIB", "tree")[-28:]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
rng = np.random.RandomState(5)
keys = rng.randint(0, 1 << 30, n, dtype=np.int64).astype(np.uint32)
kin, kout = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, n * 4)
vin, vout = hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32)), hip.Buffer(ctx, n * 4)
scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
hist = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
def histo():
    call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, 0, hist.ptr)
def sort():
    call.col_radix_sort(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, scratch.ptr, 0)
def keysort():
    call.col_radix_sort(cq.stream, kin.ptr, kout.ptr, None, None, n, 4, 0, scratch.ptr, 0)
out = []
for f, reps in ((histo, 100), (sort, 10), (keysort, 10)):
    ts = []
    for _ in range(3):
        for _ in range(5):
            f()
        cq.finish()
        ts.append(bench.time_events(hip, cq, f, reps))
    out.append(min(ts))
got = hip.read_buffer(cq, kout, np.uint32, n)
ok = bool((np.diff(got.astype(np.int64)) >= 0).all())
print("%-28s n = %d: k_hist %.4f ms (%.2f TB/s)  pair sort %.4f ms (%.1f Gkeys/s)  key sort %.4f ms  sorted %s"
      % (tag, n, out[0], n * 4 / out[0] / 1e9, out[1], n / out[1] / 1e6, out[2], ok), flush=True)

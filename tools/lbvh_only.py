"""col_lbvh alone on sorted random 30-bit codes (no traversal: safe for experimental builds whose trees are wrong).
    python tools/lbvh_only.py [n ...]      COLLISION_AMD_LIB selects the build, COLLISION_LBVH_MODE the col_debug_lbvh mode"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
import bench
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
if os.environ.get("COLLISION_LBVH_MODE"):
    cdll().col_debug_lbvh(int(os.environ["COLLISION_LBVH_MODE"]))
for n in [int(a) for a in sys.argv[1:]] or [1000000, 16000000]:
    rng = np.random.default_rng(3)
    codes = np.sort(rng.integers(0, 1 << 30, n, dtype=np.uint32))
    ids = rng.permutation(n).astype(np.uint32)
    coords = rng.random((n, 4), dtype=np.float32); radii = np.full(n, 0.002, np.float32)
    bufs = [hip.Buffer(ctx, hostbuf=a) for a in (codes, ids, coords, radii)]
    nodes, bounds = hip.Buffer(ctx, (2 * n - 1) * 16), hip.Buffer(ctx, (2 * n - 1) * 32 + 64)
    scratch = hip.Buffer(ctx, call.col_lbvh_scratch_bytes(n, 4))
    def lbvh():
        call.col_lbvh(cq.stream, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, nodes.ptr, bounds.ptr, scratch.ptr, n, 4)
    for _ in range(3): lbvh()
    cq.finish()
    t = [bench.time_events(hip, cq, lbvh, 20 if n <= 4000000 else 6) for _ in range(3)]
    print("n %9d col_lbvh %s ms  (%s)" % (n, " ".join("%.4f" % v for v in t), os.environ.get("COLLISION_AMD_LIB", "in-tree build")), flush=True)
    del bufs, nodes, bounds, scratch

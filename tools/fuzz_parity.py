"""Randomised parity soak on the GPU box: random scenes (size, distribution, radii, dtype, group size)
through the whole path against the CPU oracle, every array bit for bit; random radix sorts (size, key
and value widths, key distributions) against NumPy's stable argsort.  Test infrastructure: uses the
oracle as the checker, like tests/.

    python tools/fuzz_parity.py [seconds] [seed] [logfile] [first]     (first: skip the iterations before it)
COLLISION_FUZZ_TRAVERSE=<col_debug_traverse variant>, e.g. 32768: the dynamic packet order at every size.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as oracle_mod                                   # noqa: E402
from collision_amd import hip                                 # noqa: E402
from collision_amd._lib import call                           # noqa: E402
from tests.test_pipeline_parity import check_against_oracle   # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
log = open(sys.argv[3], "a") if len(sys.argv) > 3 and sys.argv[3] != "-" else sys.stdout
first = int(sys.argv[4]) if len(sys.argv) > 4 else 1


def say(msg):
    print(msg, file=log, flush=True)


rng = np.random.RandomState(seed)
if os.environ.get("COLLISION_FUZZ_TRAVERSE"):
    from collision_amd._lib import cdll
    cdll().col_debug_traverse(int(os.environ["COLLISION_FUZZ_TRAVERSE"]))
    say("col_debug_traverse(%s)" % os.environ["COLLISION_FUZZ_TRAVERSE"])
oracle_mod.build()
ctx = hip.Context()
cq = hip.CommandQueue(ctx)
env = (ctx, cq)
t_end = time.time() + budget
scenes = sorts = 0
last = time.time()
while time.time() < t_end:
    # ---- a scene
    u = rng.rand()
    # (10 % of the scenes between 1 M and 4.3 M spheres: the 4096-pair tile, both bucket finishes of the MSD plan)
    n = (int(2 + rng.randint(0, 10) ** 5 * 2 + rng.randint(0, 3000)) if u < 0.65 else int(rng.randint(2, 300000)) if u < 0.9
         else int(rng.randint(1000000, 4300000)))
    dtype = "float32" if rng.rand() < 0.7 else "float64"
    kind = rng.randint(0, 5)
    if kind == 0:
        pts = rng.random_sample((n, 3))
    elif kind == 1:
        k = rng.randint(1, 9)
        centres = rng.uniform(0.2, 0.8, size=(k, 3))
        pts = centres[rng.randint(0, k, size=n)] + rng.normal(0, 10.0 ** rng.uniform(-3, -1), size=(n, 3))
    elif kind == 2:                                            # many exact duplicates
        base = rng.random_sample((max(1, n // rng.randint(2, 50)), 3))
        pts = base[rng.randint(0, len(base), size=n)]
    elif kind == 3:                                            # a flat or a linear scene
        pts = rng.random_sample((n, 3))
        pts[:, rng.randint(0, 3)] = 0.5
        if rng.rand() < 0.5:
            pts[:, rng.randint(0, 3)] = 0.25
    else:                                                      # large offsets, negative coordinates
        pts = rng.random_sample((n, 3)) * 10.0 ** rng.uniform(-2, 3) - 10.0 ** rng.uniform(-2, 3)
    coords = pts.astype(dtype)
    scale = (coords.max() - coords.min() + 1e-9) * max(n, 2) ** (-1.0 / 3.0)
    if kind in (1, 2, 3) and n > 20000:                        # dense scenes: keep the pair count checkable
        scale *= 0.05 if n < 1000000 else 0.01
    rk = rng.randint(0, 3)
    radii = (np.full(n, scale * rng.uniform(0.05, 0.6)) if rk == 0 else
             rng.uniform(0.0, scale * 0.8, size=n) if rk == 1 else
             scale * 10.0 ** rng.uniform(-3, -0.1, size=n)).astype(dtype)
    gs = int(2 ** rng.randint(3, 9))
    scenes += 1
    if scenes >= first:
      t0 = time.time()
      try:
        _, _, count, _ = check_against_oracle(oracle_mod, env, coords, radii, group_size=gs, ngroups=int(rng.randint(1, 65)),
                                              sort_plan=("lsd", "msd")[scenes % 2])
      except AssertionError:
        from collision_amd.collision import Collider
        from tests.util import run_collider
        col = Collider(ctx, n, 8, gs, coords.dtype)
        ref = oracle_mod.collide(oracle_mod.pad4(coords), radii, padded=col.padded_size, capacity=1 << 24)
        cnt, got = run_collider(ctx, cq, col, coords, radii, ref["count"])
        np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_fail_seed%d_scene%d.npz" % (seed, scenes)), coords=coords,
                 radii=radii, gs=gs)
        a = set(map(tuple, np.asarray(got, dtype=np.int64).tolist()))
        b = set(map(tuple, np.asarray(ref["pairs"], dtype=np.int64).tolist()))
        pos = np.empty(n, np.int64)
        pos[ref["ids"][:n].astype(np.int64)] = np.arange(n)
        say("gpu count %d (%d listed, %d distinct), oracle count %d; missing %d, extra %d" %
            (cnt, len(got), len(a), ref["count"], len(b - a), len(a - b)))
        for name, d in (("missing", sorted(b - a)[:12]), ("extra", sorted(a - b)[:12])):
            for (x, y) in d:
                say("  %s (%d, %d): sorted positions %d, %d  packets %d, %d" % (name, x, y, pos[x], pos[y], pos[x] // 64, pos[y] // 64))
        say("scene %d: n=%d kind=%d %s gs=%d FAILED (inputs saved)" % (scenes, n, kind, dtype, gs))
        raise
      say("scene %d: n=%d kind=%d %s gs=%d pairs=%d ok (%.1f s)" % (scenes, n, kind, dtype, gs, count, time.time() - t0))
    else:
      rng.randint(1, 65)
    # ---- a sort
    n = int(rng.randint(1, 3000000)) if rng.rand() < 0.5 else int(rng.randint(1, 20000))
    kb = 4 if rng.rand() < 0.7 else 8
    vb = int(rng.choice([0, 4, 4, 8, 16, 32, 1, 2, 64, 128]))
    kd = rng.randint(0, 4)
    keys = (rng.randint(0, 2 ** 32, size=n, dtype=np.uint64) if kd == 0 else
            rng.randint(0, 2 ** rng.randint(1, 31), size=n, dtype=np.uint64) if kd == 1 else
            np.sort(rng.randint(0, 2 ** 32, size=n, dtype=np.uint64)) if kd == 2 else
            np.full(n, rng.randint(0, 2 ** 32), dtype=np.uint64))
    if kb == 8:
        keys = (keys << np.uint64(rng.randint(0, 33))) ^ rng.randint(0, 2 ** 32, size=n, dtype=np.uint64)
    keys = keys.astype(np.uint32 if kb == 4 else np.uint64)
    vals = rng.randint(0, 255, size=(n, max(vb, 1)), dtype=np.uint8)
    if scenes < first:
        sorts += 1
        continue
    kbuf, vbuf = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, hostbuf=vals)
    ko, vo = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, vals.nbytes)
    scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, kb, vb))
    call.col_radix_sort(cq.stream, kbuf.ptr, ko.ptr, vbuf.ptr if vb else None, vo.ptr if vb else None, n, kb, vb,
                        scratch.ptr, 0)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(hip.read_buffer(cq, ko, keys.dtype, n), keys[order])
    if vb:
        np.testing.assert_array_equal(hip.read_buffer(cq, vo, np.uint8, (n, vb)), vals[order])
    sorts += 1
    say("sort %d: n=%d key %d B value %d B dist %d ok" % (sorts, n, kb, vb, kd))
say("fuzz ok: %d scenes, %d sorts, seed %d" % (scenes, sorts, seed))

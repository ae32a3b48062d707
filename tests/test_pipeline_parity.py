"""Bit-exact parity of every intermediate of the path against the CPU oracle, through the C ABI.

sorted Morton keys / ids, Node records, node AABBs and the emitted pair set (with orientation)
must equal the oracle's on the same seeded inputs (BASELINE.json north_star).  Also the
size-independent properties at full size: sortedness, permutation, tree invariants, count vs
count-only, pairs verified box-by-box.
"""
import numpy as np
import pytest

from collision_amd.collision import NO_NODE, Collider
from tests.util import assert_same_pair_set, collider_state, pad4, pair_set, run_collider

pytestmark = pytest.mark.gpu


def uniform_scene(n, r, dtype, seed=4):
    rng = np.random.RandomState(seed)
    coords = rng.random_sample((n, 3)).astype(dtype)
    return coords, np.full(n, r, dtype=dtype)


def clustered_scene(n, sigma, r, dtype, seed=4):
    rng = np.random.RandomState(seed)
    centres = rng.uniform(0.2, 0.8, size=(8, 3))
    pts = np.concatenate([rng.normal(c, sigma, size=(n // 8, 3)) for c in centres])
    pts = np.concatenate([pts, rng.normal(centres[0], sigma, size=(n - len(pts), 3))])
    return pts.astype(dtype), np.full(n, r, dtype=dtype)


def check_against_oracle(oracle, hip_env, coords, radii, group_size=64, ngroups=8, capacity=None, sort_plan=None,
                         traverse_plan=None, extra_capacity=0):
    ctx, cq = hip_env
    dt = coords.dtype
    n = len(coords)
    collider = Collider(ctx, n, ngroups, group_size, dt)
    if sort_plan is not None:
        collider.sort_plan = sort_plan
    if traverse_plan is not None:
        collider.traverse_plan = traverse_plan
    first_cap = capacity if capacity is not None else max(64 * n, min(n * (n - 1) // 2, 1 << 22))
    ref = oracle.collide(oracle.pad4(coords), radii, padded=collider.padded_size, capacity=first_cap)
    if capacity is None and ref["count"] > first_cap:          # dense scene: the oracle's list was cut short
        ref = oracle.collide(oracle.pad4(coords), radii, padded=collider.padded_size, capacity=ref["count"])
    cap = ref["count"] + extra_capacity if capacity is None else capacity
    count, pairs = run_collider(ctx, cq, collider, coords, radii, cap)
    st = collider_state(cq, collider)
    np.testing.assert_array_equal(st["codes"], ref["codes"])
    np.testing.assert_array_equal(st["ids"], ref["ids"])
    leaf = n - 1
    np.testing.assert_array_equal(st["nodes"]["right_edge"], ref["nodes"]["right_edge"])
    np.testing.assert_array_equal(st["nodes"]["parent"][1:], ref["nodes"]["parent"][1:])   # root parent unwritten
    np.testing.assert_array_equal(st["nodes"]["data"][:leaf], ref["nodes"]["data"][:leaf])
    np.testing.assert_array_equal(st["nodes"]["data"][leaf:, 0], ref["nodes"]["data"][leaf:, 0])
    np.testing.assert_array_equal(st["bounds"][:, :, :3], ref["bounds"][:, :, :3])
    assert count == ref["count"]
    if capacity is None or capacity >= count:
        assert_same_pair_set(pairs, ref["pairs"])               # same orientation, any order, each once
    return collider, st, count, pairs


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n,r,gs", [(2, 0.6, 8), (3, 0.3, 8), (7, 0.3, 8), (64, 0.1, 32), (1000, 0.03, 64),
                                    (4099, 0.01, 128), (100000, 0.003, 256)])
def test_uniform_scene_matches_oracle(oracle, hip_env, dtype, n, r, gs):
    coords, radii = uniform_scene(n, r, dtype)
    check_against_oracle(oracle, hip_env, coords, radii, group_size=gs)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_box_edges_at_signed_zero(oracle, hip_env, dtype):
    """Box edges at -0.0 and +0.0.  The refit's unions are v_min / v_max instructions (csrc/lbvh.hip), which return -0 for
    min(-0, +0) and +0 for max whatever the order, where the reference's `y < x ? y : x` (collision.cl:157) keeps whichever
    came first: the two can differ in the SIGN of a zero, never in value.  So: node boxes equal BY VALUE (assert_array_equal
    treats -0 == +0), every other array bit for bit, and the same pair set -- touching boxes (an edge of one at +0, of the
    other at -0) do not collide, boxes that straddle zero do."""
    rng = np.random.RandomState(4)
    n = 4096
    coords = (rng.random_sample((n, 3)) * 2 - 1).astype(dtype) * np.asarray(0.05, dtype)
    radii = np.full(n, 0.004, dtype=dtype)
    # zero-radius spheres at -0.0 and +0.0 (boxes [-0, +0] and [+0, +0]), spheres whose box ends exactly at zero from
    # either side (c = +-r), and spheres that straddle zero
    coords[0] = (-0.0, -0.0, -0.0); radii[0] = 0.0
    coords[1] = (0.0, 0.0, 0.0); radii[1] = 0.0
    for k, sign in ((2, 1.0), (3, -1.0), (4, 1.0), (5, -1.0)):
        radii[k] = 0.002 if k < 4 else 0.003
        coords[k] = sign * radii[k]                       # lo (or hi) = c - r = exactly +0.0 (or c + r = -r + r = +0.0)
    coords[6] = (0.001, -0.001, 0.0005); radii[6] = 0.002
    coords[7] = (-0.0005, 0.0, -0.0); radii[7] = 0.001
    lo, hi = coords[:8] - radii[:8, None], coords[:8] + radii[:8, None]
    assert (np.signbit(lo) & (lo == 0)).any() and (~np.signbit(hi) & (hi == 0)).any()      # both zeros do occur as edges
    _, st, count, _ = check_against_oracle(oracle, hip_env, coords, radii, group_size=64)
    assert count > 0


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_clustered_scene_matches_oracle(oracle, hip_env, dtype):
    coords, radii = clustered_scene(20000, 0.01, 0.002, dtype)
    _, _, count, _ = check_against_oracle(oracle, hip_env, coords, radii, group_size=256)
    assert count > 20000          # a dense scene: many contacts per sphere


def test_duplicate_centres_and_equal_codes(oracle, hip_env):
    # equal Morton codes fall back to the index tie-break (collision.cl:74-76); stability of the
    # sort decides which id lands where (tests/test_radix_py.py:201)
    rng = np.random.RandomState(4)
    base = rng.random_sample((50, 3)).astype("float32")
    coords = np.repeat(base, 40, axis=0)
    rng.shuffle(coords)
    radii = np.full(len(coords), 1e-4, dtype="float32")
    check_against_oracle(oracle, hip_env, coords, radii, group_size=64)


def test_all_spheres_identical(oracle, hip_env):
    # max == min on every axis: 0/0 -> NaN -> code 0 for everybody
    coords = np.full((300, 3), 0.25, dtype="float32")
    radii = np.full(300, 0.1, dtype="float32")
    _, st, count, _ = check_against_oracle(oracle, hip_env, coords, radii, group_size=32)
    assert count == 300 * 299 // 2 and (st["codes"][:300] == 0).all()


def test_degenerate_axis(oracle, hip_env):
    coords, radii = uniform_scene(2000, 0.02, "float32")
    coords[:, 2] = 0.5
    check_against_oracle(oracle, hip_env, coords, radii)


def test_touching_boxes_do_not_collide(oracle, hip_env):
    # strict inequalities (collision.cl:164-166): faces that touch exactly are not an overlap
    coords = np.array([[0, 0, 0], [2, 0, 0], [4, 0, 0], [4, 1.5, 0]], dtype="float32")
    radii = np.ones(4, dtype="float32")
    _, _, count, pairs = check_against_oracle(oracle, hip_env, coords, radii, group_size=8)
    assert count == 1 and pair_set(np.sort(pairs, axis=1)) == {(2, 3)}


def test_varied_radii_negative_coords(oracle, hip_env):
    rng = np.random.RandomState(4)
    coords = rng.uniform(-1, 1, size=(30000, 3)).astype("float32")     # tests/benchmarks/test_collide.py:24-30
    radii = rng.uniform(0.006, 0.06, size=30000).astype("float32")
    check_against_oracle(oracle, hip_env, coords, radii, group_size=128)


def test_capacity_overflow_counts_everything(oracle, hip_env):
    coords, radii = uniform_scene(5000, 0.02, "float32")
    _, _, count, pairs = check_against_oracle(oracle, hip_env, coords, radii, capacity=100)
    assert count > 100 and len(pairs) == 100


def test_config2_full_size_properties(hip_env):
    """BASELINE config 2 (1M uniform, r=0.001) at full size, checked through properties the
    domain offers (the oracle run at this size lives in bench.py's cpu_baseline leg)."""
    ctx, cq = hip_env
    n = 1000000
    coords, radii = uniform_scene(n, 0.001, "float32")
    collider = Collider(ctx, n, 64, 256, "float32")
    count, pairs = run_collider(ctx, cq, collider, coords, radii, 1 << 17)
    st = collider_state(cq, collider)
    codes, ids, nodes, bounds = st["codes"], st["ids"], st["nodes"], st["bounds"]
    # sortedness + permutation + padding (collision.py:137-146)
    assert (np.diff(codes.astype(np.int64)) >= 0).all()
    assert (codes[n:] == 0xFFFFFFFF).all() and (ids[n:] == np.arange(n, collider.padded_size)).all()
    assert (np.sort(ids[:n]) == np.arange(n)).all()
    eq = codes[1:n] == codes[:n - 1]
    assert (ids[1:n][eq] > ids[:n - 1][eq]).all()                     # stability
    # codes are the Morton codes of the spheres they travel with
    rng_ = np.stack([coords.min(axis=0), coords.max(axis=0)])
    q = np.clip(((coords - rng_[0]) / (rng_[1] - rng_[0])) * np.float32(1023), 0, 1023).astype(np.uint32)

    def expand(v):
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        return (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    expect = (expand(q[:, 0]) << 2) + (expand(q[:, 1]) << 1) + expand(q[:, 2])
    np.testing.assert_array_equal(codes[:n], expect[ids[:n]])
    # tree invariants: every internal node is the parent of exactly its two children
    leaf = n - 1
    kids = nodes["data"][:leaf]
    assert (nodes["parent"][kids[:, 0]] == np.arange(leaf)).all()
    assert (nodes["parent"][kids[:, 1]] == np.arange(leaf)).all()
    assert len(np.unique(kids)) == 2 * leaf and nodes["right_edge"][0] == n - 1
    # a parent's box is exactly the union of its children's; the root's is the scene's
    np.testing.assert_array_equal(bounds[:leaf, 0, :3], np.minimum(bounds[kids[:, 0], 0, :3], bounds[kids[:, 1], 0, :3]))
    np.testing.assert_array_equal(bounds[:leaf, 1, :3], np.maximum(bounds[kids[:, 0], 1, :3], bounds[kids[:, 1], 1, :3]))
    np.testing.assert_array_equal(bounds[0, 0, :3], (coords - radii[:, None]).min(axis=0))
    # every reported pair really overlaps, is unique, and is oriented by sorted position
    assert count == len(pairs) and len(pair_set(pairs)) == count
    lo, hi = coords - radii[:, None], coords + radii[:, None]
    a, b = pairs[:, 0], pairs[:, 1]
    assert ((hi[a] > lo[b]) & (lo[a] < hi[b])).all()
    pos = np.empty(n, np.int64)
    pos[ids[:n]] = np.arange(n)
    assert (pos[a] < pos[b]).all()
    # count-only mode agrees
    count2, _ = run_collider(ctx, cq, collider, coords, radii, 0)
    assert count2 == count
    # expected number of AABB contacts for uniform points: n^2/2 * (4r)^3 within a few percent
    assert abs(count - n * n / 2 * (4 * 0.001) ** 3) < 0.1 * count


def test_config2_full_size_matches_oracle(oracle, hip_env):
    """BASELINE config 2 at full size (1 M uniform spheres, r = 0.001, gs = 256), every intermediate
    bit for bit against the CPU oracle (the C restatement needs < 1 s for this scene)."""
    coords, radii = uniform_scene(1000000, 0.001, "float32")
    _, _, count, _ = check_against_oracle(oracle, hip_env, coords, radii, group_size=256, ngroups=64)
    assert 30000 < count < 34000


@pytest.mark.parametrize("n", [307200, 307201])
def test_reference_benchmark_shape_count_only(oracle, hip_env, n):
    """tests/benchmarks/test_collide.py:24-54: coords U(-1,1)^3, radii U(0.006,0.06), ngroups 8,
    group_size 128, count-only mode (collisions_buf None, capacity 0)."""
    ctx, cq = hip_env
    rng = np.random.RandomState(4)
    coords = rng.uniform(-1, 1, size=(n, 3)).astype("float32")
    radii = rng.uniform(0.006, 0.06, size=n).astype("float32")
    collider = Collider(ctx, n, 8, 128, "float32")
    count, _ = run_collider(ctx, cq, collider, coords, radii, 0)
    ref = oracle.collide(oracle.pad4(coords), radii, padded=collider.padded_size, capacity=0, want=False)
    assert count == ref["count"] and count > n


def test_config3_clustered_matches_oracle(oracle, hip_env):
    """BASELINE config 3 shape at 200 k spheres: dense clusters, ~20 contacts per sphere."""
    coords, radii = clustered_scene(200000, 0.012, 0.001, "float32")
    _, _, count, _ = check_against_oracle(oracle, hip_env, coords, radii, group_size=256, ngroups=64)
    assert count > 1000000


def test_config3_full_size_matches_oracle(oracle, hip_env):
    """BASELINE config 3 exactly as bench.py's config3 leg builds it: 1 M spheres in 8 Gaussian clusters,
    sigma = 0.0152, r = 0.001, pair capacity 2^25 (about 25.4 M pairs, 51 contacts per sphere, ~50 k
    flushes of the pair staging): every array and the whole pair set against the oracle."""
    import bench
    coords4, radii = bench.clustered_scene(bench.N_SPHERES, 0.0152)
    _, _, count, pairs = check_against_oracle(oracle, hip_env, coords4[:, :3].copy(), radii, group_size=bench.GROUP_SIZE,
                                              ngroups=bench.NGROUPS, capacity=1 << 25)
    assert 24000000 < count < 27000000 and len(pairs) == count


def test_single_sphere_and_tiny_scenes(oracle, hip_env):
    ctx, cq = hip_env
    for n in (1, 2):
        coords = np.full((n, 3), 0.5, dtype="float32")
        radii = np.full(n, 0.1, dtype="float32")
        collider = Collider(ctx, n, 1, 8, "float32")
        count, pairs = run_collider(ctx, cq, collider, coords, radii, 4)
        assert count == n * (n - 1) // 2
        if n == 2:
            assert pair_set(pairs) == {(0, 1)}


def test_twenty_million_spheres_match_oracle(oracle, hip_env):
    """20 M uniform spheres on one GPU (more than 65536 chunks: three levels of group tables in the
    fused LBVH; the big radix tile; multi-level scans), every array bit for bit against the oracle."""
    n = 20000000
    rng = np.random.RandomState(7)
    coords = rng.random_sample((n, 3)).astype("float32")
    radii = np.full(n, 0.0004, dtype="float32")
    _, _, count, _ = check_against_oracle(oracle, hip_env, coords, radii, group_size=256, ngroups=64, capacity=1 << 23)
    assert abs(count - n * n / 2 * (4 * 0.0004) ** 3) < 0.05 * count          # ~819 k pairs


def test_three_million_spheres_match_oracle(oracle, hip_env):
    """3 M spheres: the 4096-pair radix tile (fused front end: the Morton kernel folds the bounds partials and counts the
    first digit), 11 719 LBVH chunks = 46 groups whose table is built while the top level is
    scanned linearly; every array bit for bit against the oracle."""
    n = 3000000
    rng = np.random.RandomState(11)
    coords = rng.random_sample((n, 3)).astype("float32")
    radii = np.full(n, 0.0007, dtype="float32")
    _, _, count, _ = check_against_oracle(oracle, hip_env, coords, radii, group_size=256, ngroups=64, capacity=1 << 21)
    assert abs(count - n * n / 2 * (4 * 0.0007) ** 3) < 0.05 * count          # ~99 k pairs


@pytest.mark.parametrize("plan", ["lsd", "msd"])
@pytest.mark.parametrize("kind,n", [("uniform", 1000), ("uniform", 200000), ("uniform", 1000000), ("clustered", 150000),
                                    ("identical", 20000), ("uniform", 2000000), ("uniform", 4000000), ("clustered", 1500000)])
def test_both_sort_plans_give_the_oracle_bits(oracle, hip_env, plan, kind, n):
    """Up to 4 M spheres col_collide_plan has two sorts (four LSD passes, or one MSD pass + a bucket
    finish in LDS: 1024-pair tile and 8192-pair buckets below 1 Mi, 4096-pair tile above, 16384-pair
    buckets above 1.9 M); clustered and identical centres put more codes into one bucket than fit, which
    the MSD plan must still sort exactly (its slow path)."""
    if kind == "uniform":
        coords, radii = uniform_scene(n, 0.1 * n ** (-1.0 / 3.0), "float32")
    elif kind == "clustered":
        coords, radii = clustered_scene(n, 0.004, 0.0004, "float32")
    else:
        coords = np.full((n, 3), 0.25, dtype="float32")
        coords[: n // 2] = 0.75
        radii = np.zeros(n, dtype="float32")                  # empty boxes never overlap each other ...
        radii[::1000] = 0.3                                   # ... a few big ones find everybody (400 k pairs)
    check_against_oracle(oracle, hip_env, coords, radii, group_size=128, ngroups=16, sort_plan=plan)


def test_auto_plan_falls_back_after_an_oversize_bucket(oracle, hip_env):
    """The Collider starts on the MSD plan; a clustered scene overflows a bucket, the kernel says so in the
    pinned word, and the following calls use the LSD plan (and stay exact throughout)."""
    ctx, cq = hip_env
    n = 120000
    coords, radii = clustered_scene(n, 0.003, 0.0003, "float32")
    collider = Collider(ctx, n, 8, 64, np.dtype("float32"))
    ref = oracle.collide(oracle.pad4(coords), radii, padded=collider.padded_size, capacity=1 << 24)
    for _ in range(4):
        count, pairs = run_collider(ctx, cq, collider, coords, radii, ref["count"])
        assert count == ref["count"] and pair_set(pairs) == pair_set(ref["pairs"])
        st = collider_state(cq, collider)
        np.testing.assert_array_equal(st["codes"], ref["codes"])
        np.testing.assert_array_equal(st["ids"], ref["ids"])
    assert collider._lsd_calls_left > 0                        # it has switched


def test_back_to_back_calls_on_a_clustered_scene(oracle, hip_env):
    """Calls enqueued WITHOUT waiting in between take the plan chosen when they were made: until the oversize
    word of an earlier call has been written that is the MSD plan through its slow path -- still exact -- and
    a call made after the word is visible (at the latest after a wait) switches to the LSD plan.  A pinned MSD
    plan reports the oversize bucket too."""
    from collision_amd import hip
    from tests.util import upload
    ctx, cq = hip_env
    n = 150000
    coords, radii = clustered_scene(n, 0.003, 0.0003, "float32")
    collider = Collider(ctx, n, 8, 64, np.dtype("float32"))
    ref = oracle.collide(oracle.pad4(coords), radii, padded=collider.padded_size, capacity=1 << 24)
    cb, rb = upload(ctx, pad4(coords)), upload(ctx, radii)
    outs = [(hip.Buffer(ctx, 4), hip.Buffer(ctx, ref["count"] * 8)) for _ in range(6)]
    for nb, pb in outs:                                        # six calls, no synchronisation
        collider.get_collisions(cq, cb, rb, nb, pb, ref["count"])
    cq.finish()
    for nb, pb in outs:
        assert int(hip.read_buffer(cq, nb, np.uint32, 1)[0]) == ref["count"]
        assert_same_pair_set(hip.read_buffer(cq, pb, np.uint32, (ref["count"], 2)), ref["pairs"])
    assert collider.oversize_bucket > 8192 or collider._lsd_calls_left > 0      # reported, or already acted upon
    collider.get_collisions(cq, cb, rb, outs[0][0], outs[0][1], ref["count"])     # sees the word: LSD from here on
    assert collider._lsd_calls_left > 0
    cq.finish()
    pinned = Collider(ctx, n, 8, 64, np.dtype("float32"))
    pinned.sort_plan = "msd"
    pinned.get_collisions(cq, cb, rb, outs[1][0], outs[1][1], ref["count"])
    cq.finish()
    assert pinned.oversize_bucket > 8192
    assert int(hip.read_buffer(cq, outs[1][0], np.uint32, 1)[0]) == ref["count"]


def test_a_clustered_scene_pays_for_one_msd_probe_only(oracle, hip_env):
    """Every LSD call reports the largest MSD bucket of its codes (read off the sorted codes); while that exceeds the
    bucket capacity the MSD plan is not tried again: 300 calls on a clustered scene take it exactly once.  When the
    scene stops being clustered the MSD plan comes back."""
    from collision_amd import hip
    from tests.util import upload
    ctx, cq = hip_env
    n = 150000
    coords, radii = clustered_scene(n, 0.003, 0.0003, "float32")
    collider = Collider(ctx, n, 8, 64, np.dtype("float32"))
    ref = oracle.collide(oracle.pad4(coords), radii, padded=collider.padded_size, capacity=1 << 24)
    cb, rb = upload(ctx, pad4(coords)), upload(ctx, radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, ref["count"] * 8)
    plans = []
    choose = collider._choose_sort_plan
    collider._choose_sort_plan = lambda: plans.append(choose()) or plans[-1]
    for _ in range(300):
        collider.get_collisions(cq, cb, rb, nb, pb, ref["count"])
        cq.finish()
    assert plans[0] == 1 and sum(plans) == 1, plans[:80]
    assert int(hip.read_buffer(cq, nb, np.uint32, 1)[0]) == ref["count"]
    assert_same_pair_set(hip.read_buffer(cq, pb, np.uint32, (ref["count"], 2)), ref["pairs"])
    # the same collider on a uniform scene: the next LSD call reports small groups, the MSD plan is taken again
    ucoords, uradii = uniform_scene(n, 0.004, "float32")
    uref = oracle.collide(oracle.pad4(ucoords), uradii, padded=collider.padded_size, capacity=1 << 24)
    ub, urb, upb = upload(ctx, pad4(ucoords)), upload(ctx, uradii), hip.Buffer(ctx, max(uref["count"], 1) * 8)
    del plans[:]
    for _ in range(collider.PLAN_RETRY_MAX // 16):
        collider.get_collisions(cq, ub, urb, nb, upb, uref["count"])
        cq.finish()
        if plans[-1] == 1:
            break
    assert plans[-1] == 1 and len(plans) <= 3, plans
    assert int(hip.read_buffer(cq, nb, np.uint32, 1)[0]) == uref["count"]


@pytest.mark.parametrize("k", [0.0, 1.5, 3.0, 1e30])
@pytest.mark.parametrize("scene", ["clustered", "identical", "uniform_dense", "tiny_capacity"])
def test_leaf_blocks_do_not_change_the_result(hip_env, oracle, k, scene):
    """Leaf blocks (col_common.h: small dense nodes whose leaves the packet walk tests at once instead of descending)
    are an optimisation of the walk only: with no marks (k = 0), the default criterion, a tight one and every small
    node marked (k huge) the whole path equals the oracle -- on heaps of overlapping spheres, on identical spheres
    (every pair collides: the staging area overflows inside blocks), and with a pair buffer that is too small."""
    from collision_amd._lib import cdll
    import ctypes
    lib = cdll()
    lib.col_debug_leaf_blocks.argtypes = [ctypes.c_float]
    lib.col_debug_leaf_blocks(ctypes.c_float(k))
    try:
        if scene == "clustered":
            coords, radii = clustered_scene(60000, 0.01, 0.002, "float32")
            check_against_oracle(oracle, hip_env, coords, radii)
        elif scene == "identical":
            coords = np.full((700, 3), 0.25, np.float32)
            check_against_oracle(oracle, hip_env, coords, np.full(700, 0.01, np.float32))
        elif scene == "uniform_dense":
            coords, radii = uniform_scene(30000, 0.03, "float32")
            check_against_oracle(oracle, hip_env, coords, radii, group_size=256)
        else:
            coords, radii = clustered_scene(20000, 0.01, 0.002, "float32")
            check_against_oracle(oracle, hip_env, coords, radii, capacity=1000)
    finally:
        lib.col_debug_leaf_blocks(ctypes.c_float(3.0))


@pytest.mark.parametrize("scene", ["clustered", "identical", "uniform_sparse", "two_blocks"])
@pytest.mark.parametrize("room", ["exact", "ample", "tiny", "count_only"])
def test_chunked_pair_allocation_gives_the_same_list(hip_env, oracle, scene, room):
    """The traversal's chunked pair allocation (dense scenes: workgroups take list space 8192 pairs at a time, a small
    kernel closes the holes) must give the same dense list as the exact one: with ample room (holes closed by
    k_pairs_compact), with a buffer of exactly the pair count (the chunks run past it: the exact walk runs again), with
    a buffer that is too small (min(count, capacity) valid distinct pairs, the counter counts everything) and in
    count-only mode."""
    if scene == "clustered":
        coords, radii = clustered_scene(200000, 0.01, 0.002, "float32")       # ~ 10^7 pairs, every block allocates
    elif scene == "identical":
        coords, radii = np.full((3000, 3), 0.25, np.float32), np.full(3000, 0.01, np.float32)      # 4.5 M pairs from 47 packets
    elif scene == "uniform_sparse":
        coords, radii = uniform_scene(100000, 0.002, "float32")                # a few hundred pairs: almost only holes
    else:
        coords, radii = uniform_scene(1500, 0.05, "float32")                    # two workgroups
    kw = dict(traverse_plan="chunked", group_size=256, ngroups=16)
    if room == "exact":
        check_against_oracle(oracle, hip_env, coords, radii, **kw)
    elif room == "ample":
        check_against_oracle(oracle, hip_env, coords, radii, extra_capacity=512 * 8192 + 1000, **kw)
    elif room == "tiny":
        _, _, count, pairs = check_against_oracle(oracle, hip_env, coords, radii, capacity=777, **kw)
        ref = oracle.collide(oracle.pad4(coords), radii, capacity=max(count, 1))
        got = pair_set(pairs)
        assert len(pairs) == min(777, count) == len(got) and got <= pair_set(ref["pairs"])
    else:
        from collision_amd import hip
        from tests.util import upload
        ctx, cq = hip_env
        collider = Collider(ctx, len(coords), 16, 256)
        collider.traverse_plan = "chunked"
        nb = hip.Buffer(ctx, 4)
        e = collider.get_collisions(cq, upload(ctx, pad4(coords)), upload(ctx, radii), nb, None, 0)
        count = int(hip.read_buffer(cq, nb, np.uint32, 1, wait_for=[e])[0])
        assert count == oracle.collide(oracle.pad4(coords), radii, capacity=0, want=False)["count"]


@pytest.mark.parametrize("room", ["exact", "ample", "tiny"])
def test_chunked_pair_allocation_float64(hip_env, oracle, room):
    """Round 4: float64 records have an asm walk of their own (one s_load_dwordx16 per 64-byte record, v_cmpx_*_f64) and
    with it the chunked pair allocation.  Same contract as for float32, on a dense clustered scene (every workgroup
    allocates, marked nodes are entered through their leaf chain) and on identical spheres (staging areas overflow)."""
    for coords, radii in (clustered_scene(60000, 0.01, 0.002, "float64"),
                          (np.full((2000, 3), 0.25, np.float64), np.full(2000, 0.01, np.float64))):
        kw = dict(traverse_plan="chunked", group_size=256, ngroups=16)
        if room == "exact":
            check_against_oracle(oracle, hip_env, coords, radii, **kw)
        elif room == "ample":
            check_against_oracle(oracle, hip_env, coords, radii, extra_capacity=512 * 8192 + 1000, **kw)
        else:
            _, _, count, pairs = check_against_oracle(oracle, hip_env, coords, radii, capacity=777, **kw)
            ref = oracle.collide(oracle.pad4(coords), radii, capacity=max(count, 1))
            got = pair_set(pairs)
            assert len(pairs) == min(777, count) == len(got) and got <= pair_set(ref["pairs"])


def test_dense_scenes_switch_to_chunked_allocation_on_their_own(hip_env, oracle):
    """"auto": a call publishes the pair count it is about to zero; from DENSE_PAIRS pairs (and with room in the list) the
    calls that follow allocate in chunks.  Exact lists throughout."""
    from collision_amd import hip
    from tests.util import upload
    ctx, cq = hip_env
    coords, radii = clustered_scene(200000, 0.01, 0.002, "float32")
    ref = oracle.collide(oracle.pad4(coords), radii, capacity=0, want=False)
    assert ref["count"] > 3000000
    ref = oracle.collide(oracle.pad4(coords), radii, capacity=ref["count"])
    cap = ref["count"] + 512 * 8192 + 4096
    collider = Collider(ctx, len(coords), 16, 256)
    collider.DENSE_PAIRS = 3000000
    cb, rb, nb, pb = upload(ctx, pad4(coords)), upload(ctx, radii), hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
    plans = []
    choose = collider._choose_plan
    collider._choose_plan = lambda capacity=0: plans.append(choose(capacity)) or plans[-1]
    for _ in range(4):
        collider.get_collisions(cq, cb, rb, nb, pb, cap)
        cq.finish()
        assert int(hip.read_buffer(cq, nb, np.uint32, 1)[0]) == ref["count"]
        assert_same_pair_set(hip.read_buffer(cq, pb, np.uint32, (ref["count"], 2)), ref["pairs"])
    assert [p & 2 for p in plans] == [0, 0, 2, 2], plans          # call k's count is published by call k + 1, seen by call k + 2


@pytest.mark.parametrize("order", [False, True])
@pytest.mark.parametrize("scene", ["uniform_f32", "uniform_f64", "clustered", "ragged_last_packet", "tiny_capacity",
                                   "back_to_back", "back_to_back_450k", "back_to_back_clustered"])
def test_dynamic_packet_order_gives_the_same_pairs(hip_env, oracle, scene, order):
    """From 1.5 M spheres on (and always with chunked allocation) the traversal's workgroups draw their packets from
    per-XCD counters instead of a fixed stride (csrc/bvh.hip: dynamic packet order; the counters are cleared by the tree
    build's last kernel).  col_debug_traverse bit 15 makes every size take that way: same arrays, same pair set, also
    when the same collider runs again (the counters must be cleared every time) and with a list that is too small.
    `order`: the walks are handed out by the cost the previous call's walks left in the collider's scratch (col_common.h: WALK
    ORDER; garbage on a first call, real times on the later calls of "back_to_back") -- any order gives the same pairs.
    Without `order` the back-to-back scenes take the sparse scenes' rule: the long walks of the previous call dealt over the
    first round's batches, at raised priority (k_cross; a few rounds of packets per XCD at 450 k spheres)."""
    from collision_amd._lib import cdll
    lib = cdll()
    lib.col_debug_traverse(32768)
    lib.col_debug_lbvh(4096 if order else 0)         # round 4: + the longest-first walk order, whatever the previous costs
    try:
        if scene == "uniform_f32":
            coords, radii = uniform_scene(300000, 0.004, "float32")
            check_against_oracle(oracle, hip_env, coords, radii, group_size=256)
        elif scene == "uniform_f64":
            coords, radii = uniform_scene(70001, 0.006, "float64")
            check_against_oracle(oracle, hip_env, coords, radii, group_size=128)
        elif scene == "clustered":
            coords, radii = clustered_scene(120000, 0.01, 0.002, "float32")
            check_against_oracle(oracle, hip_env, coords, radii, traverse_plan="exact")
        elif scene == "ragged_last_packet":
            coords, radii = uniform_scene(8192 + 129, 0.02, "float32")          # 131 packets: the 8 XCD ranges are 16 or 17 packets
            check_against_oracle(oracle, hip_env, coords, radii)
        elif scene == "tiny_capacity":
            coords, radii = clustered_scene(50000, 0.01, 0.002, "float32")
            _, _, count, pairs = check_against_oracle(oracle, hip_env, coords, radii, capacity=1000, traverse_plan="exact")
            ref = oracle.collide(oracle.pad4(coords), radii, capacity=count)
            assert len(pairs) == 1000 == len(pair_set(pairs)) and pair_set(pairs) <= pair_set(ref["pairs"])
        else:
            ctx, cq = hip_env
            if scene == "back_to_back_450k": coords, radii = uniform_scene(450000, 0.003, "float32")
            elif scene == "back_to_back_clustered": coords, radii = clustered_scene(60000, 0.01, 0.002, "float32")
            else: coords, radii = uniform_scene(200000, 0.004, "float32")
            cap = 1 << 23 if scene == "back_to_back_clustered" else 1 << 20
            ref = oracle.collide(oracle.pad4(coords), radii, capacity=cap)
            assert ref["count"] <= cap
            collider = Collider(ctx, len(coords), 8, 256)
            for _ in range(4 if scene != "back_to_back" else 3):
                count, pairs = run_collider(ctx, cq, collider, coords, radii, cap)
                assert count == ref["count"]
                assert_same_pair_set(pairs, ref["pairs"])
    finally:
        lib.col_debug_traverse(0)
        lib.col_debug_lbvh(0)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("scene,n,bound,plan", [("uniform", 50000, 65000, 1), ("uniform", 50000, 50000, 1), ("uniform", 700000, 912000, 1),
                                                 ("uniform", 700000, 912000, 0), ("clustered", 60000, 100000, 1), ("uniform", 300, 70000, 1),
                                                 ("uniform", 1, 5000, 1), ("uniform", 0, 5000, 1), ("dense", 20000, 26000, 3)])
def test_count_on_the_device_gives_the_same_arrays(hip_env, oracle, dtype, scene, n, bound, plan):
    """col_collide_plan_dev (round 4: the multi-GPU step does not wait for its owned count): the number of spheres lives in a
    device word, the host passes a BOUND (the capacity of the arrays) that sizes grids, scratch and the sort.  The first n
    sorted codes / ids, all 2 n - 1 nodes and boxes and the pair set must be those of the oracle on the n spheres -- with
    the MSD plan (the pads beyond n sit behind bucket 255's real codes and are copied through), the LSD plan, chunked pair
    allocation, a bound equal to n, and n = 0 / 1 (no tree)."""
    import ctypes as C
    from collision_amd import hip
    from collision_amd._lib import call
    from collision_amd.collision import Node
    from tests.util import download, upload
    ctx, cq = hip_env
    if scene == "clustered":
        coords, radii = clustered_scene(n, 0.01, 0.002, dtype)
    elif scene == "dense":
        coords, radii = clustered_scene(n, 0.004, 0.002, dtype)
    else:
        coords, radii = uniform_scene(max(n, 1), 0.5 * max(n, 1) ** (-1.0 / 3.0), dtype)
        coords, radii = coords[:n], radii[:n]
    gs, cb = 256, np.dtype(dtype).itemsize
    col = Collider(ctx, bound, 16, gs, dtype)
    col._allocate()
    rows = np.zeros((bound, 4), dtype)
    rows[:n, :3] = coords
    rows[n:] = 7.0                                   # rows beyond the count: never read as spheres
    rad = np.full(bound, 3.0, dtype)
    rad[:n] = radii
    rb, qb = upload(ctx, rows), upload(ctx, rad)
    cap = 1 << 22
    nb, pb, word = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8), upload(ctx, np.array([n, n], np.uint32))
    partials, parts = hip.Buffer(ctx, 256 * 8 * cb), C.c_uint32(0)
    call.col_minmax4_stage1_dev(cq.stream, rb.ptr, word.ptr, bound, cb, partials.ptr, C.byref(parts))
    call.col_collide_plan_dev(cq.stream, rb.ptr, qb.ptr, bound, col.padded_size, cb, col._codes_bufs[0].ptr, col._codes_bufs[1].ptr,
                              col._ids_bufs[0].ptr, col._ids_bufs[1].ptr, col._nodes_buf.ptr, col._bounds_buf.ptr, None,
                              col._alloc["scratch"].ptr, nb.ptr, pb.ptr, cap, plan, None, partials.ptr, parts.value, word.ptr)
    cq.finish()
    count = int(download(cq, nb, np.uint32, 1)[0])
    if n < 2:
        assert count == 0
        return
    ref = oracle.collide(oracle.pad4(coords), radii, padded=-(-n // (2 * gs)) * (2 * gs), capacity=cap)
    assert count == ref["count"]
    codes, ids = download(cq, col._codes_bufs[1], np.uint32, col.padded_size), download(cq, col._ids_bufs[1], np.uint32, col.padded_size)
    np.testing.assert_array_equal(codes[:n], ref["codes"][:n])
    np.testing.assert_array_equal(ids[:n], ref["ids"][:n])
    assert (codes[n:] == 0xFFFFFFFF).all() and (np.sort(ids[n:]) == np.arange(n, col.padded_size)).all()       # the pads, each once
    nodes = download(cq, col._nodes_buf, Node, 2 * n - 1)
    bounds = download(cq, col._bounds_buf, dtype, (2 * n - 1, 2, 4))
    np.testing.assert_array_equal(nodes["right_edge"], ref["nodes"]["right_edge"])
    np.testing.assert_array_equal(nodes["parent"][1:], ref["nodes"]["parent"][1:])
    np.testing.assert_array_equal(nodes["data"][:n - 1], ref["nodes"]["data"][:n - 1])
    np.testing.assert_array_equal(nodes["data"][n - 1:, 0], ref["nodes"]["data"][n - 1:, 0])
    np.testing.assert_array_equal(bounds[:, :, :3], ref["bounds"][:, :, :3])
    assert_same_pair_set(download(cq, pb, np.uint32, (count, 2)), ref["pairs"])

"""Kernel-level LBVH tests through the C ABI; mirrors tests/test_collision.py of the reference
(test_fill_internal :50-75, test_generate_bvh :78-128, test_generate_odd_bvh :131-179,
test_compute_bounds :182-248, test_codes :251-299, test_traverse :302-422,
test_problem_codes :425-480)."""
import numpy as np
import pytest

from collision_amd import hip
from collision_amd._lib import call
from collision_amd.collision import NO_NODE, Node
from tests.util import download, pad4, pair_set, upload

pytestmark = pytest.mark.gpu
DTYPES = ["float32", "float64"]


def _build(ctx, cq, codes, ids, bounds_buf=None, coord_bytes=4):
    n = len(codes)
    nodes_buf = upload(ctx, np.full(2 * n - 1, NO_NODE, np.uint32).repeat(4))
    codes_buf, ids_buf = upload(ctx, codes), upload(ctx, ids)      # keep alive until the read-back
    call.col_bvh_build(cq.stream, codes_buf.ptr, ids_buf.ptr, nodes_buf.ptr,
                       None if bounds_buf is None else bounds_buf.ptr, n, coord_bytes)
    return nodes_buf, download(cq, nodes_buf, Node, 2 * n - 1)


def test_fill_internal(hip_env):
    ctx, cq = hip_env
    n = 8
    ids = np.random.RandomState(4).permutation(n).astype(np.uint32)
    _, nodes = _build(ctx, cq, np.arange(n, dtype=np.uint32), ids)
    np.testing.assert_equal(nodes["data"][n - 1:, 0], ids)
    np.testing.assert_equal(nodes["right_edge"][n - 1:], np.arange(n))


@pytest.mark.parametrize("name", ["bvh_fig3_8", "bvh_fig3_7"])
def test_generate_bvh(hip_env, vectors, name):
    ctx, cq = hip_env
    v = vectors[name]
    codes = np.array(v["codes"], dtype=np.uint32)
    n = len(codes)
    _, nodes = _build(ctx, cq, codes, np.arange(n, dtype=np.uint32))
    expected = np.array([(p, r, d) for p, r, d in v["internal"]], dtype=Node)
    np.testing.assert_equal(nodes[1:n - 1], expected[1:])
    np.testing.assert_equal(nodes[["data", "right_edge"]][0], expected[["data", "right_edge"]][0])
    np.testing.assert_equal(nodes["parent"][n - 1:], v["leaf_parents"])
    np.testing.assert_equal(nodes["right_edge"][n - 1:], np.arange(n))
    np.testing.assert_equal(nodes["data"][n - 1:, 0], np.arange(n))


def test_problem_codes(hip_env, vectors, oracle):
    ctx, cq = hip_env
    codes = np.array(vectors["problem_codes"]["codes"], dtype=np.uint32)
    ids = np.arange(len(codes), dtype=np.uint32)
    _, nodes = _build(ctx, cq, codes, ids)
    assert set(nodes["parent"][1:].tolist()) == set(range(len(codes) - 1))
    ref = oracle.build_bvh(codes, ids)
    np.testing.assert_equal(nodes[1:len(codes) - 1], ref[1:len(codes) - 1])


@pytest.mark.parametrize("dt", DTYPES)
def test_compute_bounds(hip_env, vectors, dt):
    ctx, cq = hip_env
    v = vectors["compute_bounds"]
    coords = pad4(np.array(v["coords"], dtype=dt))
    radii = np.array(v["radii"], dtype=dt)
    nodes = np.array([(p, r, d) for p, r, d in v["nodes"]], dtype=Node)
    bounds_buf = hip.Buffer(ctx, len(nodes) * 8 * np.dtype(dt).itemsize)
    flags_buf = upload(ctx, np.zeros(len(nodes), np.uint32))
    coords_buf, radii_buf, nodes_buf = upload(ctx, coords), upload(ctx, radii), upload(ctx, nodes)
    call.col_bvh_refit(cq.stream, bounds_buf.ptr, flags_buf.ptr, coords_buf.ptr, radii_buf.ptr,
                       nodes_buf.ptr, len(coords), np.dtype(dt).itemsize)
    bounds = download(cq, bounds_buf, dt, (len(nodes), 2, 4))
    np.testing.assert_equal(bounds[:, :, :3], np.array(v["expected"], dtype=dt))


@pytest.mark.parametrize("dt", DTYPES)
def test_codes(hip_env, vectors, dt):
    ctx, cq = hip_env
    v = vectors["morton_codes"]
    coords = np.array(v["coords"], dtype=dt)
    rng = pad4(np.array([coords.min(axis=0), coords.max(axis=0)]))
    codes_buf, ids_buf = hip.Buffer(ctx, 16 * 4), hip.Buffer(ctx, 16 * 4)
    coords_buf, rng_buf = upload(ctx, pad4(coords)), upload(ctx, rng)
    call.col_morton(cq.stream, coords_buf.ptr, rng_buf.ptr, 6, 16, np.dtype(dt).itemsize,
                    codes_buf.ptr, ids_buf.ptr)
    codes = download(cq, codes_buf, np.uint32)
    np.testing.assert_equal(codes[:6], np.array(v["expected"], dtype=np.uint32))
    assert (codes[6:] == 0xFFFFFFFF).all()                          # collision.py:137-142
    np.testing.assert_equal(download(cq, ids_buf, np.uint32), np.arange(16))   # collision.cl:8-10


@pytest.mark.parametrize("dt", DTYPES)
def test_traverse(hip_env, vectors, dt):
    """The whole kernel chain with a host argsort in place of the radix sort (test_collision.py:302-422)."""
    ctx, cq = hip_env
    v = vectors["six_sphere_scene"]
    cb = np.dtype(dt).itemsize
    coords = np.array(v["coords"], dtype=dt)
    radii = np.array(v["radii"], dtype=dt)
    n = len(coords)
    coords_buf, radii_buf = upload(ctx, pad4(coords)), upload(ctx, radii)
    rng = pad4(np.array([coords.min(axis=0), coords.max(axis=0)]))
    codes_buf, rng_buf = hip.Buffer(ctx, n * 4), upload(ctx, rng)
    call.col_morton(cq.stream, coords_buf.ptr, rng_buf.ptr, n, n, cb, codes_buf.ptr, None)
    codes = download(cq, codes_buf, np.uint32)
    order = np.argsort(codes, kind="mergesort").astype(np.uint32)
    bounds_buf = hip.Buffer(ctx, (2 * n - 1) * 8 * cb)
    nodes_buf, _ = _build(ctx, cq, codes[order], order, bounds_buf, cb)
    flags_buf = upload(ctx, np.zeros(2 * n - 1, np.uint32))
    call.col_bvh_refit(cq.stream, bounds_buf.ptr, flags_buf.ptr, coords_buf.ptr, radii_buf.ptr, nodes_buf.ptr, n, cb)
    pairs_buf = upload(ctx, np.full(4, 0xFFFFFFFF, np.uint32))
    count_buf = upload(ctx, np.zeros(1, np.uint32))
    call.col_traverse(cq.stream, pairs_buf.ptr, count_buf.ptr, 2, nodes_buf.ptr, bounds_buf.ptr, n, cb)
    assert download(cq, count_buf, np.uint32)[0] == 2
    assert pair_set(download(cq, pairs_buf, np.uint32, (2, 2))) == pair_set(v["expected_pairs"])


def test_random_trees_match_oracle(hip_env, oracle):
    """Karras topology on random sorted codes with many duplicates, several sizes."""
    ctx, cq = hip_env
    rs = np.random.RandomState(4)
    for n, hi in ((2, 4), (3, 2), (17, 8), (1000, 300), (5000, 2 ** 30), (65537, 50000)):
        codes = np.sort(rs.randint(0, hi, size=n).astype(np.uint32))
        ids = rs.permutation(n).astype(np.uint32)
        _, nodes = _build(ctx, cq, codes, ids)
        ref = oracle.build_bvh(codes, ids)
        np.testing.assert_equal(nodes["right_edge"], ref["right_edge"])
        np.testing.assert_equal(nodes["parent"][1:], ref["parent"][1:])
        np.testing.assert_equal(nodes["data"][:n - 1], ref["data"][:n - 1])
        np.testing.assert_equal(nodes["data"][n - 1:, 0], ref["data"][n - 1:, 0])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 1000, 65536, 65537, 70000, 300000])
def test_fused_lbvh_equals_generic_kernels_and_oracle(hip_env, oracle, dt, n):
    """col_lbvh (range-query refit, csrc/lbvh.hip) == col_bvh_build + col_bvh_refit == oracle,
    on sorted codes with many duplicates; sizes straddle the 256-leaf chunk and 65536-leaf group."""
    ctx, cq = hip_env
    cb = np.dtype(dt).itemsize
    rs = np.random.RandomState(n)
    codes = np.sort(rs.randint(0, max(2, n // 3), size=n).astype(np.uint32))
    ids = rs.permutation(n).astype(np.uint32)
    coords = pad4(rs.uniform(-1, 1, size=(n, 3)).astype(dt))
    radii = rs.uniform(0.001, 0.05, size=n).astype(dt)
    codes_buf, ids_buf, coords_buf, radii_buf = (upload(ctx, a) for a in (codes, ids, coords, radii))
    nn = 2 * n - 1

    def fresh():
        return upload(ctx, np.full(nn, NO_NODE, np.uint32).repeat(4)), upload(ctx, np.zeros((nn, 2, 4), dt))

    nodes_a, bounds_a = fresh()
    scratch = hip.Buffer(ctx, call.col_lbvh_scratch_bytes(n, cb))
    call.col_lbvh(cq.stream, codes_buf.ptr, ids_buf.ptr, coords_buf.ptr, radii_buf.ptr, nodes_a.ptr, bounds_a.ptr,
                  scratch.ptr, n, cb)
    nodes_b, bounds_b = fresh()
    flags = upload(ctx, np.zeros(nn, np.uint32))
    call.col_bvh_build(cq.stream, codes_buf.ptr, ids_buf.ptr, nodes_b.ptr, bounds_b.ptr, n, cb)
    call.col_bvh_refit(cq.stream, bounds_b.ptr, flags.ptr, coords_buf.ptr, radii_buf.ptr, nodes_b.ptr, n, cb)
    na, nb = download(cq, nodes_a, Node, nn), download(cq, nodes_b, Node, nn)
    ba, bb = download(cq, bounds_a, dt, (nn, 2, 4)), download(cq, bounds_b, dt, (nn, 2, 4))
    np.testing.assert_array_equal(na, nb)                  # untouched fields keep the NO_NODE fill in both
    ua, ub = ba.view(np.uint32 if cb == 4 else np.uint64), bb.view(np.uint32 if cb == 4 else np.uint64)
    np.testing.assert_array_equal(ua[:, :, :3], ub[:, :, :3])               # boxes, bit for bit
    np.testing.assert_array_equal(ua[:, 0, 3], ub[:, 0, 3])                 # skip links
    # down links: equal, except where the fused build marked a LEAF BLOCK (a small dense node): the mark must name the
    # node's own leaf range -- first leaf (leftmost descendant) and count (col_common.h)
    da, db = ua[:, 1, 3].astype(np.uint64), ub[:, 1, 3].astype(np.uint64)
    marked = (da != db)
    assert not marked[n - 1:].any()                                         # never a leaf
    if marked.any():
        first = np.arange(nn, dtype=np.int64)
        for _ in range(64):                                                 # leftmost leaf of every node
            inner = first < n - 1
            if not inner.any():
                break
            first[inner] = nb["data"][first[inner], 0]
        lo = first - (n - 1)
        hi = np.where(np.arange(nn) < n - 1, nb["right_edge"], np.arange(nn) - (n - 1)).astype(np.int64)
        idx = np.nonzero(marked)[0]
        assert ((hi[idx] - lo[idx]) < 16).all()
        np.testing.assert_array_equal(da[idx], 0x80000000 | (lo[idx] << 4) | (hi[idx] - lo[idx]))
    if n >= 2:
        ref_nodes = oracle.build_bvh(codes, ids)
        ref_bounds = oracle.node_bounds(coords, radii, ref_nodes)
        np.testing.assert_array_equal(na["right_edge"], ref_nodes["right_edge"])
        np.testing.assert_array_equal(na["parent"][1:], ref_nodes["parent"][1:])
        np.testing.assert_array_equal(na["data"][:n - 1], ref_nodes["data"][:n - 1])
        np.testing.assert_array_equal(ba[:, :, :3], ref_bounds[:, :, :3])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", [2, 5, 64, 300, 5000])
def test_refit_on_a_degenerate_chain_tree(hip_env, oracle, dt, n):
    """col_bvh_refit takes ANY binary tree.  A caterpillar (internal node i = {leaf i, internal i+1})
    is n-1 levels deep: far beyond the level-synchronous sweeps, so the fence-walk finisher does
    most of the work.  Compared with the oracle's restatement of leafBounds + internalBounds."""
    ctx, cq = hip_env
    rs = np.random.RandomState(n)
    coords = pad4(rs.uniform(-1, 1, size=(n, 3)).astype(dt))
    radii = rs.uniform(0.01, 0.1, size=n).astype(dt)
    leaf = n - 1
    nodes = np.zeros(2 * n - 1, dtype=Node)
    nodes["parent"][:] = NO_NODE
    for i in range(n - 1):
        right = i + 1 if i + 1 < n - 1 else leaf + n - 1
        nodes["data"][i] = (leaf + i, right)
        nodes["right_edge"][i] = n - 1
        nodes["parent"][leaf + i] = i
        nodes["parent"][right] = i
    ids = rs.permutation(n).astype(np.uint32)
    nodes["data"][leaf:, 0] = ids
    nodes["right_edge"][leaf:] = np.arange(n)
    bounds_buf = upload(ctx, np.zeros((2 * n - 1, 2, 4), dt))
    flags_buf = upload(ctx, np.zeros(2 * n - 1, np.uint32))
    coords_buf, radii_buf, nodes_buf = upload(ctx, coords), upload(ctx, radii), upload(ctx, nodes)
    call.col_bvh_refit(cq.stream, bounds_buf.ptr, flags_buf.ptr, coords_buf.ptr, radii_buf.ptr, nodes_buf.ptr, n,
                       np.dtype(dt).itemsize)
    got = download(cq, bounds_buf, dt, (2 * n - 1, 2, 4))
    ref = oracle.node_bounds(coords, radii, nodes)
    np.testing.assert_array_equal(got[:, :, :3], ref[:, :, :3])

"""Pins the CPU oracle (oracle/) against the reference's golden vectors.

CPU only.  Every case cites the reference test it comes from; the vectors are
in tests/golden/ (see make_golden.py).
"""
import numpy as np
import pytest

NO = 0xFFFFFFFF


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_codes(oracle, vectors, dt):
    # tests/test_collision.py:251-299
    v = vectors["morton_codes"]
    coords = np.array(v["coords"], dtype=dt)
    rng = oracle.pad4(np.array([coords.min(axis=0), coords.max(axis=0)]))
    codes = oracle.morton(oracle.pad4(coords), rng)
    np.testing.assert_equal(codes, np.array(v["expected"], dtype=np.uint32))


def test_bounds_matches_numpy(oracle):
    # tests/test_bounds_py.py:18-148 (min/max exact vs NumPy)
    rng = np.random.RandomState(4)
    for dt, width in (("float32", 4), ("float64", 4), ("float32", 1)):
        vals = rng.normal(scale=1e8, size=(100, width)).astype(dt)
        out = oracle.bounds(vals)
        np.testing.assert_equal(out, np.stack([vals.min(axis=0), vals.max(axis=0)]))


def test_fill_internal(oracle, vectors):
    # tests/test_collision.py:50-75
    n = vectors["fill_internal"]["n"]
    ids = np.random.RandomState(4).permutation(n).astype(np.uint32)
    nodes = oracle.build_bvh(np.arange(n, dtype=np.uint32), ids)
    np.testing.assert_equal(nodes["data"][n - 1:, 0], ids)
    np.testing.assert_equal(nodes["right_edge"][n - 1:], np.arange(n))


@pytest.mark.parametrize("name", ["bvh_fig3_8", "bvh_fig3_7"])
def test_generate_bvh(oracle, vectors, name):
    # tests/test_collision.py:78-179
    v = vectors[name]
    codes = np.array(v["codes"], dtype=np.uint32)
    n = len(codes)
    nodes = oracle.build_bvh(codes, np.arange(n, dtype=np.uint32))
    for i, (parent, right_edge, children) in enumerate(v["internal"]):
        if i > 0:  # root parent is never written (test_collision.py:121-123)
            assert nodes["parent"][i] == parent
        assert nodes["right_edge"][i] == right_edge
        assert list(nodes["data"][i]) == children
    np.testing.assert_equal(nodes["parent"][n - 1:], v["leaf_parents"])
    np.testing.assert_equal(nodes["right_edge"][n - 1:], np.arange(n))
    np.testing.assert_equal(nodes["data"][n - 1:, 0], np.arange(n))


def test_problem_codes(oracle, vectors):
    # tests/test_collision.py:425-480
    codes = np.array(vectors["problem_codes"]["codes"], dtype=np.uint32)
    nodes = oracle.build_bvh(codes, np.arange(len(codes), dtype=np.uint32))
    assert set(nodes["parent"][1:].tolist()) == set(range(len(codes) - 1))


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_compute_bounds(oracle, vectors, dt):
    # tests/test_collision.py:182-248
    v = vectors["compute_bounds"]
    coords = oracle.pad4(np.array(v["coords"], dtype=dt))
    radii = np.array(v["radii"], dtype=dt)
    nodes = np.array([(p, r, d) for p, r, d in v["nodes"]], dtype=oracle.Node)
    b = oracle.node_bounds(coords, radii, nodes)
    np.testing.assert_equal(b[:, :, :3], np.array(v["expected"], dtype=dt))


def _pipeline(oracle, coords, radii, padded=None):
    coords4 = oracle.pad4(coords)
    n = len(coords4)
    cap = n * n
    return oracle.collide(coords4, radii, padded=padded or n, capacity=cap)


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_six_sphere_scene(oracle, vectors, generated, dt):
    # tests/test_collision.py:302-422, tests/test_collision_py.py:49-97 (orientation checked unsorted)
    v = vectors["six_sphere_scene"]
    r = _pipeline(oracle, np.array(v["coords"], dtype=dt), np.array(v["radii"], dtype=dt), padded=16)
    assert r["count"] == 2
    assert set(map(tuple, r["pairs"].tolist())) == set(map(tuple, v["expected_pairs"]))
    assert set(map(tuple, generated["six_pairs"].tolist())) == set(map(tuple, v["expected_pairs"]))


@pytest.mark.parametrize("dt", ["float32", "float64"])
@pytest.mark.parametrize("size", [8, 100, 120, 256, 317, 341, 351])
def test_random_scenes_vs_reference_find_collisions(oracle, generated, dt, size):
    # tests/test_collision_py.py:100-296; expected = reference's find_collisions (generated.npz)
    key = "scene_%s_%d" % (dt, size)
    coords, radii = generated[key + "_coords"], generated[key + "_radii"]
    expected = set(map(tuple, generated[key + "_pairs"].tolist()))
    r = _pipeline(oracle, coords, radii)
    assert r["count"] == len(expected)
    got = set(map(tuple, np.sort(r["pairs"], axis=1).tolist()))
    assert got == expected
    # the oracle's own restatements of find_collisions agree too
    assert oracle.find_collisions(coords, radii) == expected
    cnt, bf = oracle.brute_force(oracle.pad4(coords), radii)
    assert cnt == len(expected) and set(map(tuple, bf.tolist())) == expected
    if dt == "float32":  # the all-cores count used by bench.py's many-core CPU figure
        assert oracle.brute_force_count_all_cores(oracle.pad4(coords), radii)[0] == len(expected)


@pytest.mark.parametrize("size,r", [(2000, 0.005), (10000, 0.001)])
def test_config1_vs_reference_find_collisions(oracle, generated, size, r):
    # BASELINE config 1 generator; expected from the reference's find_collisions
    rng = np.random.RandomState(4)
    coords = rng.random_sample((size, 3)).astype("float32")
    radii = np.full(size, r, dtype="float32")
    expected = set(map(tuple, generated["config1_%d_pairs" % size].tolist()))
    res = oracle.collide(oracle.pad4(coords), radii, capacity=4 * len(expected) + 16)
    assert res["count"] == len(expected)
    assert set(map(tuple, np.sort(res["pairs"], axis=1).tolist())) == expected


def test_scan_goldens(oracle, vectors):
    # tests/test_scan.py:24-103
    v = vectors["local_scan"]
    out, sums = oracle.local_scan(v["values"], v["block"])
    np.testing.assert_equal(out, v["expected"])
    np.testing.assert_equal(sums, v["block_sums"])
    w = vectors["block_scan"]
    top, _ = oracle.local_scan(w["block_sums_in"], len(w["block_sums_in"]))
    np.testing.assert_equal(top, w["block_sums_scanned"])
    np.testing.assert_equal(oracle.block_scan(w["values"], w["block"], top), w["expected"])
    # composed == exclusive scan (tests/test_scan_py.py:46-65)
    vals = np.random.RandomState(4).randint(0, 1024, size=1024).astype(np.uint32)
    ex = oracle.exclusive_scan(vals)
    assert ex[0] == 0
    np.testing.assert_equal(ex[1:], np.cumsum(vals)[:-1])


def _radix_key(values, bits, rpass):
    # tests/test_radix.py:56-57
    return (values >> (rpass * bits)) & ((2 ** bits) - 1)


@pytest.mark.parametrize("key_dtype", ["uint32", "uint64"])
@pytest.mark.parametrize("ngroups,group_size", [(1, 8), (3, 8), (4, 8), (8, 32), (16, 128)])
def test_block_sort_and_scatter(oracle, key_dtype, ngroups, group_size):
    # tests/test_radix.py:61-182: per block stable digit sort, digit-major histogram, global stable scatter
    bits = 4
    rs = np.random.RandomState(4)
    keys = rs.randint(0, 64, size=(ngroups, group_size * 2)).astype(key_dtype)
    for rpass in range(np.dtype(key_dtype).itemsize * 8 // bits):
        digit = _radix_key(keys, bits, rpass)
        order = np.argsort(digit, kind="mergesort", axis=1)
        expected_blocks = np.take_along_axis(keys, order, axis=1)
        k, _, hist = oracle.block_sort(keys.reshape(-1), None, group_size * 2, bits, rpass)
        np.testing.assert_equal(k.reshape(keys.shape), expected_blocks)
        for b in range(ngroups):
            np.testing.assert_equal(hist[:, b], np.bincount(digit[b].astype(np.int64), minlength=16))
        offset = oracle.exclusive_scan(hist.reshape(-1)).reshape(hist.shape)
        out, _ = oracle.scatter(k, None, group_size * 2, bits, rpass, offset, hist)
        bdig = _radix_key(expected_blocks, bits, rpass)
        np.testing.assert_equal(out, expected_blocks.flat[np.argsort(bdig, axis=None, kind="mergesort")])


@pytest.mark.parametrize("key_dtype", ["uint32", "uint64"])
@pytest.mark.parametrize("size,group_size", [(32, 8), (15360, 32)])
def test_sorter(oracle, key_dtype, size, group_size):
    # tests/test_radix_py.py:83-201
    rs = np.random.RandomState(4)
    keys = rs.randint(500, size=size).astype(key_dtype)
    vals = rs.uniform(-1000, 1000, size=(size, 4))
    k, v = oracle.radix_sort(keys, vals, block=2 * group_size, bits=4)
    np.testing.assert_equal(k, np.sort(keys))
    np.testing.assert_equal(v, vals[np.argsort(keys, kind="mergesort")])


def test_offsets(oracle, generated_meta):
    # tests/test_offset_py.py:23-62
    lits = generated_meta["offset_literals"]
    assert lits[0][:2] == [0, 0]
    np.testing.assert_equal(oracle.find_offsets(lits[0], max(lits[0]) + 2), lits[1])
    np.testing.assert_equal(oracle.find_offsets(lits[2], 7), lits[3])


def test_host_size_helpers_match_reference(oracle, generated_meta):
    # collision/misc.py:31-35 outputs recorded from the reference
    for x, b, e in generated_meta["roundUp"]:
        assert oracle.lib().orc_round_up(x, b) == e

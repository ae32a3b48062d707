#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container only).

Part A writes ``reference_vectors.json``: the literal input/expected arrays the
reference's own tests hold for the hot path, transcribed as data with the
file:line each comes from.

Part B imports the reference's pure-NumPy helpers from /root/reference --
``find_collisions`` (tests/test_collision_py.py:30-37), ``roundUp`` /
``nextPowerOf2`` / ``dtype_sizeof`` (collision/misc.py:31-71) and the radix
test helpers (tests/test_radix.py:27-30,56-57) -- and records their outputs on
the seeded scenes the reference's tests use (``generated.npz`` +
``generated_meta.json``).  The reference's modules ``import pyopencl`` at the
top; PyOpenCL is not installed here and is only dereferenced inside functions
that are never called by this script, so an empty module object is registered
under that name for the import to succeed.  No reference file travels: the
outputs are plain arrays.

    python tests/golden/make_golden.py
"""
import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def part_a():
    NO = 0xFFFFFFFF
    six_spheres = [[0.0, 1.0, 3.0], [0.0, 1.0, 3.0], [4.0, 1.0, 8.0],
                   [-4.0, -6.0, 3.0], [-5.0, 0.0, -1.0], [-5.0, 0.5, -0.5]]
    v = {
        "_doc": "Literal vectors from /root/reference/tests (data only). 'leaf+k' is stored as n-1+k.",
        "morton_codes": {
            "src": "tests/test_collision.py:251-299 (f32 and f64)",
            "coords": six_spheres,
            "range": "min/max over the coords (test_collision.py:260-261)",
            "expected": [862940378, 862940378, 1073741823, 20332620, 302580864, 306295426],
        },
        "fill_internal": {
            "src": "tests/test_collision.py:50-75",
            "n": 8,
            "doc": "ids = any permutation; nodes[n-1+i].data[0] == ids[i], right_edge == i",
        },
        "bvh_fig3_8": {
            "src": "tests/test_collision.py:78-128",
            "codes": [0b00001, 0b00010, 0b00100, 0b00101, 0b10011, 0b11000, 0b11001, 0b11110],
            # (parent, right_edge, [child_a, child_b]) for internal nodes 0..6, leaf = 7
            "internal": [[NO, 7, [3, 4]], [3, 1, [7 + 0, 7 + 1]], [3, 3, [7 + 2, 7 + 3]], [0, 3, [1, 2]],
                         [0, 7, [7 + 4, 5]], [4, 7, [6, 7 + 7]], [5, 6, [7 + 5, 7 + 6]]],
            "leaf_parents": [1, 1, 2, 2, 4, 6, 6, 5],
        },
        "bvh_fig3_7": {
            "src": "tests/test_collision.py:131-179",
            "codes": [0b00001, 0b00010, 0b00100, 0b00101, 0b10011, 0b11000, 0b11001],
            "internal": [[NO, 6, [3, 4]], [3, 1, [6 + 0, 6 + 1]], [3, 3, [6 + 2, 6 + 3]], [0, 3, [1, 2]],
                         [0, 6, [6 + 4, 5]], [4, 6, [6 + 5, 6 + 6]]],
            "leaf_parents": [1, 1, 2, 2, 4, 5, 5],
        },
        "problem_codes": {
            "src": "tests/test_collision.py:425-480 (every internal node 0..n-2 appears as a parent)",
            "codes": [
                0b00000000000000000000000000000000, 0b00000000000000000000000000000000,
                0b00000110110000110100000100000010, 0b00001001001001001001001001001001,
                0b00001001001001001001001001001001, 0b00010010010010010010010010010010,
                0b00010010010010010010010010010010, 0b00010010011010010010011011011010,
                0b00011001001011001001011001001011, 0b00011011011011011011011011011011,
                0b00100100010000100010110100010110, 0b00100100100100100100100100100100,
                0b00100100100101101101100101100100, 0b00101001101001101101101101101001,
                0b00101101101101101101101101101101, 0b00110110110110110110110110110110,
                0b00110110110110110110110110110110, 0b00110110110110110110110110110110,
                0b00111111111111111111111111111111, 0b00111111111111111111111111111111,
                0b00111111111111111111111111111111],
        },
        "compute_bounds": {
            "src": "tests/test_collision.py:182-248",
            "coords": [[0.0, 1.0, 3.0], [4.0, 1.0, 8.0], [-4.0, -6.0, 3.0], [-5.0, 0.0, -1.0]],
            "radii": [1.0, 1.0, 1.0, 1.0],
            # (parent, right_edge, data[2]) for all 7 nodes, leaf = 3
            "nodes": [[NO, 3, [3 + 0, 1]], [0, 3, [3 + 3, 2]], [1, 2, [3 + 1, 3 + 2]],
                      [0, 0, [2, NO]], [2, 1, [0, NO]], [2, 2, [1, NO]], [1, 3, [3, NO]]],
            "expected": [[[-6.0, -7.0, -2.0], [5.0, 2.0, 9.0]], [[-6.0, -1.0, -2.0], [5.0, 2.0, 9.0]],
                         [[-1.0, 0.0, 2.0], [5.0, 2.0, 9.0]], [[-5.0, -7.0, 2.0], [-3.0, -5.0, 4.0]],
                         [[-1.0, 0.0, 2.0], [1.0, 2.0, 4.0]], [[3.0, 0.0, 7.0], [5.0, 2.0, 9.0]],
                         [[-6.0, -1.0, -2.0], [-4.0, 1.0, 0.0]]],
        },
        "six_sphere_scene": {
            "src": "tests/test_collision.py:302-422 and tests/test_collision_py.py:49-97",
            "coords": six_spheres,
            "radii": [1.0] * 6,
            "expected_pairs": [[0, 1], [4, 5]],
            "doc": "pair orientation is checked without a per-pair sort (test_collision_py.py:97)",
        },
        "local_scan": {
            "src": "tests/test_scan.py:24-60",
            "values": [17, 6, 24, 28, 18, 22, 2, 1, 25, 17, 7, 17, 3, 19, 8, 23],
            "block": 8,
            "expected": [0, 17, 23, 47, 75, 93, 115, 117, 0, 25, 42, 49, 66, 69, 88, 96],
            "block_sums": [118, 119],
        },
        "block_scan": {
            "src": "tests/test_scan.py:63-103",
            "values": [0, 17, 23, 47, 75, 93, 115, 117, 0, 25, 42, 49, 66, 69, 88, 96],
            "block": 8,
            "block_sums_in": [118, 119],
            "block_sums_scanned": [0, 118],
            "expected": [0, 17, 23, 47, 75, 93, 115, 117, 118, 143, 160, 167, 184, 187, 206, 214],
        },
        "block_lengths": {
            "src": "tests/test_scan_py.py:32-43",
            "cases": [[1024, 4, [128, 16, 2]], [20, 2, [8, 2]], [24, 4, [8]], [1032, 4, [136, 24, 4]],
                      [160, 4, [24, 4]], [320, 4, [40, 8]]],
        },
        "scanner_errs": {"src": "tests/test_scan_py.py:15-29", "cases": [[1023, 4], [20, 4], [96, 6]]},
        "padded_size": {
            "src": "tests/test_collision_py.py:40-46",
            "cases": [[48, 3, 8, 48], [47, 3, 8, 48], [49, 3, 8, 64]],
        },
        "sorter_errs": {
            "src": "tests/test_radix_py.py:33-42 (size, group_size, bits) -> ValueError",
            "cases": [[128, 8, 3], [128, 9, 4], [122, 8, 4], [128, 4, 4]],
        },
        "num_passes": {
            "src": "tests/test_radix_py.py:68-80 (bits, group_size, passes for u32; x2 for u64)",
            "cases": [[1, 4, 32], [2, 4, 16], [4, 8, 8], [8, 128, 4]],
        },
        "misc": {
            "src": "tests/test_misc.py:4-46",
            "roundUp": [[4, 5, 5], [5, 5, 5], [0, 5, 0], [4, 2, 4], [5, 2, 6], [0, 2, 0]],
            "nextPowerOf2": [[1, 1], [2, 2], [3, 4], [5, 8], [6, 8]],
        },
        "offsets": {
            "src": "tests/test_offset_py.py:23-62",
            "cases": "see generated_meta.json (read from the reference test at generation time)",
        },
    }
    (HERE / "reference_vectors.json").write_text(json.dumps(v, indent=1))


def part_b():
    sys.modules.setdefault("pyopencl", types.ModuleType("pyopencl"))
    sys.path.insert(0, str(REF))
    from tests.test_collision_py import find_collisions          # tests/test_collision_py.py:30-37
    from collision.misc import roundUp, nextPowerOf2, dtype_sizeof  # collision/misc.py:31-71
    from collision.scan import ceildiv                            # collision/scan.py:7

    out = {}
    meta = {"_doc": "outputs of the reference's own NumPy helpers; see make_golden.py"}

    # Seeded random scenes of tests/test_collision_py.py:100-296 (np.random.seed(4) inside each test).
    for dt in ("float32", "float64"):
        for size in (8, 100, 120, 256, 317, 341, 351):
            np.random.seed(4)
            coords = np.random.random((size, 3)).astype(dt)
            radius = 1 / (size ** 0.5)
            radii = np.random.uniform(0, radius, len(coords)).astype(dt)
            pairs = sorted(find_collisions(coords, radii))
            key = "scene_%s_%d" % (dt, size)
            out[key + "_coords"] = coords
            out[key + "_radii"] = radii
            out[key + "_pairs"] = np.array(pairs, dtype=np.uint32).reshape(-1, 2)
    meta["scenes"] = "tests/test_collision_py.py:100-296: seed 4, coords=random((n,3)), radii=uniform(0, n**-0.5)"

    # BASELINE config 1 generator at reduced sizes: RandomState(4), r const.
    for size, r in ((2000, 0.005), (10000, 0.001)):
        rng = np.random.RandomState(4)
        coords = rng.random_sample((size, 3)).astype("float32")
        radii = np.full(size, r, dtype="float32")
        pairs = sorted(find_collisions(coords, radii))
        out["config1_%d_pairs" % size] = np.array(pairs, dtype=np.uint32).reshape(-1, 2)
    meta["config1"] = "RandomState(4).random_sample((n,3)).astype(f32), radii=full(n, r): (2000, 0.005), (10000, 0.001)"

    # the six-sphere scene through the reference's find_collisions
    six = np.array([[0.0, 1.0, 3.0], [0.0, 1.0, 3.0], [4.0, 1.0, 8.0],
                    [-4.0, -6.0, 3.0], [-5.0, 0.0, -1.0], [-5.0, 0.5, -0.5]], dtype="float32")
    out["six_pairs"] = np.array(sorted(find_collisions(six, np.ones(6, "float32"))), dtype=np.uint32)

    meta["roundUp"] = [[x, b, int(roundUp(x, b))] for x in (0, 1, 47, 48, 49, 1000000, 307201)
                       for b in (1, 16, 128, 256, 512)]
    meta["nextPowerOf2"] = [[x, int(nextPowerOf2(x))] for x in (1, 2, 3, 5, 6, 127, 128, 129, 1000)]
    meta["ceildiv"] = [[a, b, int(ceildiv(a, b))] for a in (0, 1, 7, 8, 9, 1000) for b in (1, 8, 256)]
    meta["dtype_sizeof"] = [[repr(d), int(dtype_sizeof(np.dtype(d)))] for d in
                            ("uint32", "int64", ("float32", 3), ("float32", 4), ("float64", 3),
                             ("float32", (4, 3)))]

    # offset goldens are literals in tests/test_offset_py.py:23-62; read them as data.
    import ast
    src = (REF / "tests" / "test_offset_py.py").read_text()
    lits = [ast.literal_eval(n) for n in ast.walk(ast.parse(src))
            if isinstance(n, ast.List) and all(isinstance(e, ast.Constant) for e in n.elts) and len(n.elts) >= 6]
    meta["offset_literals"] = lits

    np.savez_compressed(HERE / "generated.npz", **out)
    (HERE / "generated_meta.json").write_text(json.dumps(meta, indent=1))


if __name__ == "__main__":
    part_a()
    if REF.exists():
        part_b()
    else:
        print("no /root/reference here: generated.npz left as committed")

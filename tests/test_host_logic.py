"""Host-side logic and the C-ABI surface; no GPU needed.

Mirrors the size/dtype/ValueError tests of the reference: tests/test_misc.py, and the
non-device parts of tests/test_scan_py.py, tests/test_radix_py.py, tests/test_collision_py.py.
"""
import os
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from collision_amd import _lib, hip
from collision_amd.misc import dtype_decl, dtype_sizeof, nextPowerOf2, product, roundUp

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def ctx():
    return hip.Context()


# ---------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "collision_hip.h").read_text()
    declared = set(re.findall(r"\b(col_[a-z0-9_]+)\s*\(", header))
    # diagnostics live in a header of their own: not part of the drop-in ABI, but exported for tools/
    debug = set(re.findall(r"\b(col_[a-z0-9_]+)\s*\(", (ROOT / "include" / "collision_hip_debug.h").read_text()))
    assert not any(name.startswith("col_debug") for name in declared) and not (declared & debug)
    lib = _lib.cdll()
    exported = {line.split()[-1] for line in
                subprocess.check_output(["nm", "-D", str(_lib.LIB_PATH)]).decode().splitlines()
                if " T " in line}
    assert declared, "no declarations parsed"
    # built with -fvisibility=hidden: the .so exports exactly what the header declares, internal helpers stay inside
    assert declared | debug == exported, (declared | debug) ^ exported
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert debug == set(_lib.DEBUG_EXPORTS), debug ^ set(_lib.DEBUG_EXPORTS)
    for name in declared | debug:
        assert hasattr(lib, name)
    assert lib.col_version() >= 100
    assert _lib.call.col_radix_tile(1 << 26, 4, 4) % 256 == 0 and _lib.call.col_radix_tile(1000, 4, 4) % 256 == 0
    assert lib.col_error_string(-1).decode().startswith("collision_hip")


def test_scratch_size_queries_are_monotonic():
    c = _lib.call
    assert c.col_scan_scratch_bytes(1) > 0
    assert c.col_scan_scratch_bytes(1 << 26) > c.col_scan_scratch_bytes(1 << 16)
    small, big = c.col_radix_scratch_bytes(1 << 10, 4, 4), c.col_radix_scratch_bytes(1 << 20, 4, 4)
    assert big > small and big >= (1 << 20) * 8
    assert c.col_collide_scratch_bytes(1000, 1024, 4) > c.col_radix_scratch_bytes(1024, 4, 4)


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_cdll", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libcollision_hip.so")
    with pytest.raises(ImportError):
        _lib.cdll()


def test_product_never_imports_the_oracle():
    for py in (ROOT / "collision_amd").glob("*.py"):
        assert "oracle" not in py.read_text(), py


# ---------------------------------------------------------------- misc (tests/test_misc.py:4-46)
def test_misc(vectors, generated_meta):
    for x, base, expected in vectors["misc"]["roundUp"]:
        assert roundUp(x, base) == expected
    for x, expected in vectors["misc"]["nextPowerOf2"]:
        assert nextPowerOf2(x) == expected
    for x, base, expected in generated_meta["roundUp"]:       # outputs of the reference's roundUp
        assert roundUp(x, base) == expected
    for x, expected in generated_meta["nextPowerOf2"]:
        assert nextPowerOf2(x) == expected
    assert product([1, 2, 3]) == 6 and product([]) == 1


def test_np_dtype():
    assert dtype_decl(np.dtype("uint32")) == "uint"
    assert dtype_decl(np.dtype("float16")) == "half"
    assert dtype_decl(np.dtype(("float16", 4))) == "half4"
    with pytest.raises(ValueError):
        dtype_decl(np.dtype(("float16", 5)))
    with pytest.raises(ValueError):
        dtype_decl(np.dtype(("float16", (3, 4))))


def test_dtype_sizeof():
    assert dtype_sizeof(np.dtype("uint32")) == 4
    assert dtype_sizeof(np.dtype("int64")) == 8
    assert dtype_sizeof(np.dtype(("float32", 3))) == 16
    assert dtype_sizeof(np.dtype(("float32", (4, 3)))) == 16 * 4
    assert dtype_sizeof(np.dtype((("float32", 3), 4))) == 16 * 4
    with pytest.raises(ValueError):
        dtype_sizeof(np.dtype(("float16", 5)))
    with pytest.raises(TypeError):
        dtype_sizeof(np.dtype([("foo", "float32")]))
    with pytest.raises(TypeError):
        dtype_sizeof(np.dtype(([("foo", "float32")], 4)))


# ---------------------------------------------------------------- scan (tests/test_scan_py.py:15-43)
def test_scanner_errs(ctx, vectors):
    from collision_amd.radix import PrefixScanner, PrefixScanProgram
    program = PrefixScanProgram(ctx)
    for size, group_size in vectors["scanner_errs"]["cases"]:
        with pytest.raises(ValueError):
            PrefixScanner(ctx, size, group_size, program=program)
    scanner = PrefixScanner(ctx, 1024, 4, program=program)
    with pytest.raises(ValueError):
        scanner.resize(1023, 4)
    assert (scanner.size, scanner.group_size) == (1024, 4)
    with pytest.raises(ValueError):
        PrefixScanner(ctx, 1024, 4, program=PrefixScanProgram(hip.Context(1)))


def test_block_levels(ctx, vectors):
    from collision_amd.scan import PrefixScanner
    for size, group_size, expected in vectors["block_lengths"]["cases"]:
        assert PrefixScanner(ctx, size, group_size).block_lengths == tuple(expected)


# ---------------------------------------------------------------- radix (tests/test_radix_py.py:33-80)
def test_sorter_errs(ctx, vectors):
    from collision_amd.radix import PrefixScanProgram, RadixProgram, RadixSorter
    prog, scan = RadixProgram(ctx), PrefixScanProgram(ctx)
    for size, group_size, bits in vectors["sorter_errs"]["cases"]:
        with pytest.raises(ValueError):
            RadixSorter(ctx, size, group_size, bits, program=prog, scan_program=scan)
    with pytest.raises(ValueError):
        RadixSorter(ctx, 128, 8, 4, key_dtype="uint16", program=prog, scan_program=scan)
    with pytest.raises(ValueError):
        RadixSorter(ctx, 128, 8, 4, value_dtype="uint16", program=prog, scan_program=scan)
    with pytest.raises(ValueError):
        RadixProgram(ctx, key_dtype="int32")
    sorter = RadixSorter(ctx, 64, 8, 4, program=prog, scan_program=scan)
    with pytest.raises(ValueError):
        sorter.resize(64, 5, 4)
    assert (sorter.size, sorter.group_size, sorter.radix_bits) == (64, 8, 4)   # rolled back


@pytest.mark.parametrize("key_dtype", ["uint32", "uint64"])
def test_num_passes(ctx, vectors, key_dtype):
    from collision_amd.radix import RadixProgram, RadixSorter
    prog = RadixProgram(ctx, key_dtype)
    for bits, group_size, expected in vectors["num_passes"]["cases"]:
        sorter = RadixSorter(ctx, 512, group_size, bits, key_dtype, program=prog)
        assert sorter.num_passes == expected * (2 if key_dtype == "uint64" else 1)


def test_histogram_len(ctx):
    from collision_amd.radix import RadixSorter
    # radix.py:113-116 and SURVEY.md section 8: P=1000448, gs=256 -> 31744; gs=128 -> 62720
    assert RadixSorter(ctx, 1000448, 256).histogram_len == 31744
    assert RadixSorter(ctx, 1000192, 128).histogram_len == 62720


# ---------------------------------------------------------------- collider (tests/test_collision_py.py)
def test_padded_size(ctx, vectors):
    from collision_amd.collision import Collider
    for size, ngroups, group_size, expected in vectors["padded_size"]["cases"]:
        assert Collider(ctx, size, ngroups, group_size, "float32").padded_size == expected


def test_count_err(ctx):
    # tests/test_collision_py.py:298-327: None pair buffer with capacity > 0 -> ValueError
    from collision_amd.collision import Collider
    collider = Collider(ctx, 100, 5, 8)
    with pytest.raises(ValueError):
        collider.get_collisions(None, None, None, None, None, 7)


@pytest.mark.parametrize("dt", ["float32", np.dtype("float32"), "float64", np.dtype("float64")])
def test_collider_dtype(ctx, dt):
    from collision_amd.collision import Collider
    collider = Collider(ctx, 100, 5, 8, coord_dtype=dt)
    assert collider.program.coord_dtype == np.dtype(dt)
    assert collider.reducer.program.value_dtype == np.dtype((dt, 3))


def test_program_mismatch_errs(ctx):
    from collision_amd.bounds import BoundsProgram
    from collision_amd.collision import Collider, CollisionProgram
    with pytest.raises(ValueError):
        CollisionProgram(ctx, "int32")
    with pytest.raises(ValueError):
        Collider(ctx, 100, 5, 8, "float32", program=CollisionProgram(ctx, "float64"))
    with pytest.raises(ValueError):
        Collider(ctx, 100, 5, 8, "float32", program=CollisionProgram(hip.Context(1), "float32"))
    with pytest.raises(ValueError):
        Collider(ctx, 100, 5, 8, "float32", reducer_program=BoundsProgram(ctx, ("float64", 3)))


def test_collider_resize_host_side(ctx):
    from collision_amd.collision import Collider
    collider = Collider(ctx, 350, 8, 64)
    collider.resize(351, 8, 64)
    assert (collider.size, collider.padded_size, collider.n_nodes) == (351, 384, 701)
    assert collider.sorter.size == 384
    collider.resize(130)                       # the reference raises here (see collision.py docstring)
    assert collider.sorter.size == collider.padded_size == 256


def test_drop_in_import_path():
    import collision.collision as cc
    import collision.radix as cr
    from collision_amd.collision import Collider
    assert cc.Collider is Collider and cc.NO_NODE == 0xFFFFFFFF and cc.Node.itemsize == 16
    assert hasattr(cr, "PrefixScanner") and hasattr(cr, "PrefixScanProgram")


def test_scratch_sizes_are_monotone_in_n():
    """A scratch buffer sized for n must serve every n' <= n (multi.py sizes the Collider once for its
    capacity and calls col_collide with the varying number of owned spheres).  The radix tile grows with
    n, so the histogram of a smaller input can be LARGER: the sizing functions take the maximum."""
    from collision_amd import _lib
    call = _lib.call
    for kb, vb in ((4, 4), (4, 0), (8, 8), (4, 32)):
        for thr in (1 << 20, 8 << 20, 16 << 20):
            sizes = [call.col_radix_scratch_bytes(n, kb, vb) for n in range(thr - 3000, thr + 3000, 256)]
            assert sizes == sorted(sizes), (kb, vb, thr)
    for cb in (4, 8):
        for thr in (1 << 20, 8 << 20, 16 << 20):
            ns = list(range(thr - 4096, thr + 4096, 512))
            sizes = [call.col_collide_scratch_bytes(n, n, cb) for n in ns]
            assert sizes == sorted(sizes), (cb, thr)
    # the two cases of the round-1 advisor note
    assert call.col_collide_scratch_bytes(1048064, 1048064, 4) <= call.col_collide_scratch_bytes(1049088, 1049088, 4)
    assert call.col_collide_scratch_bytes(16776704, 16776704, 4) <= call.col_collide_scratch_bytes(16777728, 16777728, 4)
    assert call.col_collide_scratch_bytes(8388096, 8388096, 4) <= call.col_collide_scratch_bytes(8389120, 8389120, 4)


def test_rendered_reductions_compile_without_a_device():
    """Any accumulator list (reduce.py:9-22) that is not compiled in is rendered into HIP and compiled by hiprtc at run
    time; the rendering and the compile step need no GPU (col_reduce_rtc_check).  Also: what the compiler says about
    a function it does not know reaches the caller."""
    import ctypes as C
    from collision_amd._lib import cdll
    from collision_amd.reduce import render_source
    from tests.test_bounds_py import RENDERED_LISTS
    lib = cdll()
    log = C.create_string_buffer(1 << 14)
    for name, acc in RENDERED_LISTS.items():
        for shape in ((), (3,)):
            src = render_source(np.dtype((name, shape)) if shape else np.dtype(name), acc)
            assert lib.col_reduce_rtc_check(src.encode(), b"gfx950", log, len(log)) == 0, log.value.decode()
    src = render_source(np.dtype("float32"), [("0", "no_such_function")])
    assert lib.col_reduce_rtc_check(src.encode(), b"gfx950", log, len(log)) != 0
    assert b"no_such_function" in log.value


def test_generic_reduction_restatement_agrees_with_numpy(oracle):
    """oracle.reduce_list (reduce.cl:5-58 for any accumulator list, the checker of the rendered reductions) against
    plain NumPy where the order of operations does not matter."""
    rs = np.random.RandomState(2)
    v = rs.randint(-1000, 1000, size=(5000, 3)).astype(np.int64)
    for geometry in ((8, 64), (5, 48), (1, 1)):
        out = oracle.reduce_list(v, [("0", "ADD"), ("LONG_MAX", "min"), ("LONG_MIN", "max")], *geometry)
        np.testing.assert_array_equal(out, np.stack([v.sum(0), v.min(0), v.max(0)]))
    f = rs.random_sample((4096, 2)).astype(np.float32)
    out = oracle.reduce_list(f, [("INFINITY", "fmin"), ("-INFINITY", "fmax"), ("0", "ADD")], 8, 64)
    np.testing.assert_array_equal(out[:2], np.stack([f.min(0), f.max(0)]))
    np.testing.assert_allclose(out[2], f.sum(0, dtype=np.float64), rtol=1e-5)


def test_another_build_of_the_library_can_be_selected():
    """COLLISION_AMD_LIB names another build of the same ABI (tools/ab_builds.sh times two builds on one box): the binding loads
    that file, and fails loudly when it does not exist -- there is no fallback."""
    import sys
    from collision_amd import _lib
    code = "from collision_amd import _lib; print(_lib.LIB_PATH); print(_lib.cdll().col_version())"
    env = dict(os.environ, COLLISION_AMD_LIB=str(_lib._HERE / "libcollision_hip.so"), COLLISION_AMD_NO_TORCH="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(_lib._HERE.parent))
    assert out.returncode == 0, out.stderr
    assert out.stdout.splitlines()[0].endswith("libcollision_hip.so") and int(out.stdout.splitlines()[1]) >= 1
    env["COLLISION_AMD_LIB"] = "/nonexistent/libcollision_hip.so"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(_lib._HERE.parent))
    assert out.returncode != 0


def test_climb_names_the_in_chunk_nodes_of_the_karras_tree(oracle):
    """csrc/lbvh.hip builds the nodes whose range lies inside one chunk of 256 sorted leaves bottom-up from the adjacent deltas
    instead of by Karras' searches (collision.cl:81-121).  The CPU restatement of that climb must name exactly those nodes of the
    oracle's tree, each with the oracle's other end and split -- distinct codes, heaps of equal codes, few distinct codes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("climb_prototype", Path(__file__).parent / "analysis" / "climb_prototype.py")
    proto = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(proto)
    rng = np.random.default_rng(17)
    named = sum(proto.check(oracle, codes, rng) for codes in proto.scenes(rng, 18, 2500))
    assert named > 10000

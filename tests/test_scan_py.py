"""PrefixScanner on the GPU; mirrors tests/test_scan_py.py:46-92 and tests/test_scan.py:24-103."""
import numpy as np
import pytest

from collision_amd import hip
from collision_amd._lib import call
from collision_amd.radix import PrefixScanner, PrefixScanProgram
from tests.util import download, upload

pytestmark = pytest.mark.gpu


def _check(cq, scanner, values):
    buf = upload(scanner.program.context, values)
    e = scanner.prefix_sum(cq, buf)
    out = download(cq, buf, np.uint32, wait_for=[e])
    assert out[0] == 0
    np.testing.assert_equal(out[1:], np.cumsum(values, dtype=np.uint32)[:-1])     # wraps mod 2^32


@pytest.mark.parametrize("size,group_size", [(20, 2), (24, 4), (1024, 4), (160, 4), (320, 4)])
def test_prefix_sum(hip_env, size, group_size):
    ctx, cq = hip_env
    scanner = PrefixScanner(ctx, size, group_size, program=PrefixScanProgram(ctx))
    _check(cq, scanner, np.random.RandomState(4).randint(0, size, size=size).astype(np.uint32))


@pytest.mark.parametrize("old_shape,new_shape", [((20, 2), (24, 4)), ((1024, 4), (160, 4)), ((24, 2), (None, 4)),
                                                 ((160, 4), (1024, None))])
def test_scanner_resized(hip_env, old_shape, new_shape):
    ctx, cq = hip_env
    scanner = PrefixScanner(ctx, *old_shape)
    _check(cq, scanner, np.ones(old_shape[0], np.uint32))
    scanner.resize(*new_shape)
    size = new_shape[0] or old_shape[0]
    _check(cq, scanner, np.random.RandomState(4).randint(0, 100, size=size).astype(np.uint32))


@pytest.mark.parametrize("size", [2048, 2304, 307200, 1536000, 3072000, 4196352, 33554432])
def test_prefix_sum_benchmark_sizes(hip_env, size):
    # tests/benchmarks/test_scan.py:29-53 sizes, one 2-level and one 3-level size, values in [0, 128)
    ctx, cq = hip_env
    scanner = PrefixScanner(ctx, size, 128)
    _check(cq, scanner, np.random.RandomState(4).randint(0, 128, size=size).astype(np.uint32))


def test_wraps_modulo_2_32(hip_env):
    ctx, cq = hip_env
    scanner = PrefixScanner(ctx, 8192, 64)
    _check(cq, scanner, np.full(8192, 0xFFFFFFF0, dtype=np.uint32))


def test_ragged_sizes_through_the_c_abi(hip_env):
    ctx, cq = hip_env
    rs = np.random.RandomState(4)
    for n in (1, 7, 2047, 2049, 100003):
        vals = rs.randint(0, 1000, size=n).astype(np.uint32)
        buf = upload(ctx, vals)
        scratch = hip.Buffer(ctx, call.col_scan_scratch_bytes(n))
        call.col_scan_u32(cq.stream, buf.ptr, n, scratch.ptr)
        exp = np.zeros(n, np.uint32)
        exp[1:] = np.cumsum(vals, dtype=np.uint32)[:-1]
        np.testing.assert_array_equal(download(cq, buf, np.uint32, n), exp)


def test_reference_kernels_golden(hip_env, vectors):
    # tests/test_scan.py:24-103 literal vectors through col_local_scan / col_block_scan
    ctx, cq = hip_env
    v = vectors["local_scan"]
    vals = np.array(v["values"], np.uint32)
    buf, sums = upload(ctx, vals), hip.Buffer(ctx, 8)
    call.col_local_scan(cq.stream, buf.ptr, len(vals), v["block"], sums.ptr)
    np.testing.assert_array_equal(download(cq, buf, np.uint32), v["expected"])
    np.testing.assert_array_equal(download(cq, sums, np.uint32), v["block_sums"])
    w = vectors["block_scan"]
    top = upload(ctx, np.array(w["block_sums_in"], np.uint32))
    call.col_local_scan(cq.stream, top.ptr, 2, 2, None)
    np.testing.assert_array_equal(download(cq, top, np.uint32), w["block_sums_scanned"])
    data = upload(ctx, np.array(w["values"], np.uint32))
    call.col_block_scan(cq.stream, data.ptr, 16, w["block"], top.ptr)
    np.testing.assert_array_equal(download(cq, data, np.uint32), w["expected"])

"""Kernel-level radix tests through the C ABI.

(a) the reference's own per-pass structure -- block_sort / scatter with block = 2*group_size and a
    digit-major histogram (tests/test_radix.py:61-351) -- via col_ref_block_sort / col_ref_scatter;
(b) the production pass -- col_radix_histogram -> col_scan_u32 -> col_radix_scatter -- checked
    per pass against a stable argsort by the digit, and its histogram against bincount per tile.
"""
import numpy as np
import pytest

from collision_amd import hip
from collision_amd._lib import call
from tests.util import download, upload

pytestmark = pytest.mark.gpu

SIZES = [(1, 8), (3, 8), (4, 8), (8, 32), (16, 128)]


def radix_key(values, bits, rpass):           # tests/test_radix.py:56-57
    return (values >> np.array(rpass * bits, values.dtype)) & np.array((1 << bits) - 1, values.dtype)


def excl(x):                                  # tests/test_radix.py:27-30
    r = np.zeros_like(x)
    r[1:] = np.cumsum(x)[:-1]
    return r


@pytest.mark.parametrize("key_dtype", ["uint32", "uint64"])
@pytest.mark.parametrize("ngroups,group_size", SIZES)
def test_ref_block_sort_and_scatter(hip_env, oracle, key_dtype, ngroups, group_size):
    ctx, cq = hip_env
    bits, block = 4, 2 * group_size
    kb = np.dtype(key_dtype).itemsize
    rs = np.random.RandomState(4)
    keys = rs.randint(0, 64, size=ngroups * block).astype(key_dtype)
    hist_buf = hip.Buffer(ctx, 16 * ngroups * 4)
    for rpass in range(kb * 8 // bits):
        keys_buf = upload(ctx, keys)
        call.col_ref_block_sort(cq.stream, keys_buf.ptr, None, len(keys), kb, 0, block, bits, rpass, hist_buf.ptr)
        got_keys = download(cq, keys_buf, key_dtype)
        got_hist = download(cq, hist_buf, np.uint32, (16, ngroups))
        exp_keys, _, exp_hist = oracle.block_sort(keys, None, block, bits, rpass)
        np.testing.assert_array_equal(got_keys, exp_keys)
        np.testing.assert_array_equal(got_hist, exp_hist)
        digit = radix_key(keys.reshape(ngroups, block), bits, rpass)
        order = np.argsort(digit, kind="mergesort", axis=1)
        np.testing.assert_array_equal(got_keys.reshape(ngroups, block),
                                      np.take_along_axis(keys.reshape(ngroups, block), order, axis=1))
        offs = excl(got_hist.reshape(-1)).astype(np.uint32)
        out_buf = hip.Buffer(ctx, keys.nbytes)
        offs_buf = upload(ctx, offs)
        call.col_ref_scatter(cq.stream, keys_buf.ptr, out_buf.ptr, None, None, len(keys), kb, 0, block, bits, rpass,
                             offs_buf.ptr, hist_buf.ptr)
        out = download(cq, out_buf, key_dtype)
        np.testing.assert_array_equal(out, keys[np.argsort(radix_key(keys, bits, rpass), kind="mergesort")])


@pytest.mark.parametrize("key_dtype", ["uint32", "uint64"])
@pytest.mark.parametrize("value_dtype", ["uint32", "float64"])
@pytest.mark.parametrize("ngroups,group_size", SIZES)
def test_ref_argsort_loop(hip_env, key_dtype, value_dtype, ngroups, group_size):
    # tests/test_radix.py:249-351: the whole multi-pass loop with a host-side scan, checked after every pass
    ctx, cq = hip_env
    bits, block = 4, 2 * group_size
    kb, vb = np.dtype(key_dtype).itemsize, np.dtype(value_dtype).itemsize
    rs = np.random.RandomState(4)
    keys = rs.randint(0, 64, size=ngroups * block).astype(key_dtype)
    values = rs.uniform(-1000, 1000, size=len(keys)).astype(value_dtype)
    sort_keys, sort_values = keys, values
    keys_buf, vals_buf = upload(ctx, keys), upload(ctx, values)
    out_k, out_v = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, values.nbytes)
    hist_buf, offs_buf = hip.Buffer(ctx, 16 * ngroups * 4), hip.Buffer(ctx, 16 * ngroups * 4)
    for rpass in range(kb * 8 // bits):
        call.col_ref_block_sort(cq.stream, keys_buf.ptr, vals_buf.ptr, len(keys), kb, vb, block, bits, rpass, hist_buf.ptr)
        hist = download(cq, hist_buf, np.uint32)
        hip.write_buffer(cq, offs_buf, excl(hist).astype(np.uint32))
        call.col_ref_scatter(cq.stream, keys_buf.ptr, out_k.ptr, vals_buf.ptr, out_v.ptr, len(keys), kb, vb, block,
                             bits, rpass, offs_buf.ptr, hist_buf.ptr)
        hip.enqueue_copy(cq, keys_buf, out_k)
        hip.enqueue_copy(cq, vals_buf, out_v)
        order = np.argsort(radix_key(sort_keys, bits, rpass), kind="mergesort")
        sort_keys, sort_values = sort_keys[order], sort_values[order]
        np.testing.assert_array_equal(download(cq, keys_buf, key_dtype), sort_keys)
        np.testing.assert_array_equal(download(cq, vals_buf, value_dtype), sort_values)
    order = np.argsort(keys, kind="mergesort")
    np.testing.assert_array_equal(download(cq, keys_buf, key_dtype), keys[order])
    np.testing.assert_array_equal(download(cq, vals_buf, value_dtype), values[order])


@pytest.mark.parametrize("key_dtype,val_bytes", [("uint32", 0), ("uint32", 4), ("uint32", 8), ("uint64", 4),
                                                 ("uint64", 16), ("uint32", 32)])
@pytest.mark.parametrize("n", [5000, 70001])
@pytest.mark.parametrize("force_tile", [0, 4096, 8192])
def test_production_pass(hip_env, key_dtype, val_bytes, n, force_tile):
    """One histogram + scatter pass at a time against NumPy; force_tile runs the 4096- and
    8192-pair kernels (normally chosen from 1 Mi / 8 Mi elements) on the same small inputs."""
    call.col_debug_radix_tile(force_tile)
    try:
        _production_pass(hip_env, key_dtype, val_bytes, n)
    finally:
        call.col_debug_radix_tile(0)


def _production_pass(hip_env, key_dtype, val_bytes, n):
    ctx, cq = hip_env
    kb = np.dtype(key_dtype).itemsize
    tile = call.col_radix_tile(n, kb, val_bytes)
    nblocks = -(-n // tile)
    rs = np.random.RandomState(4)
    keys = (rs.randint(0, 2 ** 32, size=n, dtype=np.uint64) * np.uint64(2654435761)).astype(key_dtype)
    vals = rs.randint(0, 255, size=(n, max(val_bytes, 1))).astype(np.uint8)
    keys_buf, vals_buf = upload(ctx, keys), upload(ctx, vals)
    out_k, out_v = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, vals.nbytes)
    hist_buf = hip.Buffer(ctx, 256 * nblocks * 4)
    scratch = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nblocks))
    for rpass in range(kb):
        call.col_radix_histogram(cq.stream, keys_buf.ptr, n, kb, val_bytes, rpass, hist_buf.ptr)
        hist = download(cq, hist_buf, np.uint32, (256, nblocks))
        digit = radix_key(keys, 8, rpass).astype(np.int64)
        for b in range(nblocks):                    # digit-major layout, radix.cl:99-100
            np.testing.assert_array_equal(hist[:, b], np.bincount(digit[b * tile:(b + 1) * tile], minlength=256))
        call.col_scan_u32(cq.stream, hist_buf.ptr, 256 * nblocks, scratch.ptr)
        np.testing.assert_array_equal(download(cq, hist_buf, np.uint32), excl(hist.reshape(-1)))
        call.col_radix_scatter(cq.stream, keys_buf.ptr, out_k.ptr, vals_buf.ptr if val_bytes else None,
                               out_v.ptr if val_bytes else None, n, kb, val_bytes, rpass, hist_buf.ptr)
        order = np.argsort(digit, kind="stable")
        np.testing.assert_array_equal(download(cq, out_k, key_dtype), keys[order])
        if val_bytes:
            np.testing.assert_array_equal(download(cq, out_v, np.uint8, vals.shape), vals[order])


@pytest.mark.parametrize("big", [1, 63, 64, 65, 4095, 8191, 8192, 8193, 20000, 70000])
def test_msd_sort_bucket_sizes(hip_env, big):
    """col_radix_sort_msd (col_collide's second sort plan): one bucket of exactly `big` codes next to
    ordinary ones and the 0xFFFFFFFF pads -- up to 8192 a bucket is finished in LDS, above it goes
    through the chunked global path and is reported in *oversize.  Result = stable sort on all 32 bits."""
    ctx, cq = hip_env
    rs = np.random.RandomState(big)
    n_real, pads = 300000 + big, 777
    low = rs.randint(0, 1 << 22, size=n_real, dtype=np.uint64)
    low[::5] = low[1]                                              # duplicates: stability matters
    digit = rs.randint(0, 256, size=n_real).astype(np.uint64)
    digit[digit == 5] = 6
    digit[rs.choice(n_real, size=big, replace=False)] = 5          # bucket 5 holds exactly `big` codes
    keys = np.concatenate([((digit << np.uint64(22)) | low).astype(np.uint32), np.full(pads, 0xFFFFFFFF, np.uint32)])
    n = len(keys)
    vals = np.arange(n, dtype=np.uint32)
    tile = call.col_radix_tile(n, 4, 4)
    assert tile == 1024
    nb = -(-n // tile)
    d8 = ((keys >> np.uint32(22)) & np.uint32(255)).astype(np.int64)
    hist = np.zeros((256, nb), np.uint32)
    np.add.at(hist, (d8, np.arange(n) // tile), 1)
    scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
    kb, vb = upload(ctx, keys), upload(ctx, vals)
    ko, vo = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, vals.nbytes)
    flag = upload(ctx, np.zeros(1, np.uint32))
    hip.write_buffer(cq, scratch, hist)                            # the bucket-digit histogram col_morton_tile would leave
    call.col_radix_sort_msd(cq.stream, kb.ptr, ko.ptr, vb.ptr, vo.ptr, n, scratch.ptr, flag.ptr)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(download(cq, ko, np.uint32, n), keys[order])
    np.testing.assert_array_equal(download(cq, vo, np.uint32, n), order.astype(np.uint32))
    assert int(download(cq, flag, np.uint32, 1)[0]) == (big if big > 8192 else 0)


@pytest.mark.parametrize("n_real,big", [(1200000, 8193), (1200000, 30000), (2500000, 16384), (2500000, 16385), (3900000, 100)])
def test_msd_sort_middle_tile(hip_env, n_real, big):
    """col_radix_sort_msd above 1 Mi codes: the global pass runs on the 4096-pair tile, and above 1.9 M codes
    a bucket is finished in a 16384-pair LDS image (8192 below); larger buckets take the chunked path and are
    reported.  Result = stable sort on all 32 bits."""
    ctx, cq = hip_env
    rs = np.random.RandomState(big)
    pads = 333
    low = rs.randint(0, 1 << 22, size=n_real, dtype=np.uint64)
    low[::5] = low[1]
    digit = rs.randint(0, 250, size=n_real).astype(np.uint64)
    digit[digit == 5] = 6
    digit[rs.choice(n_real, size=big, replace=False)] = 5          # bucket 5 holds exactly `big` codes
    keys = np.concatenate([((digit << np.uint64(22)) | low).astype(np.uint32), np.full(pads, 0xFFFFFFFF, np.uint32)])
    n = len(keys)
    vals = np.arange(n, dtype=np.uint32)
    tile = call.col_radix_tile(n, 4, 4)
    assert tile == 4096
    nb = -(-n // tile)
    d8 = ((keys >> np.uint32(22)) & np.uint32(255)).astype(np.int64)
    hist = np.zeros((256, nb), np.uint32)
    np.add.at(hist, (d8, np.arange(n) // tile), 1)
    scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
    kb, vb = upload(ctx, keys), upload(ctx, vals)
    ko, vo = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, vals.nbytes)
    flag = upload(ctx, np.zeros(1, np.uint32))
    hip.write_buffer(cq, scratch, hist)
    call.col_radix_sort_msd(cq.stream, kb.ptr, ko.ptr, vb.ptr, vo.ptr, n, scratch.ptr, flag.ptr)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(download(cq, ko, np.uint32, n), keys[order])
    np.testing.assert_array_equal(download(cq, vo, np.uint32, n), order.astype(np.uint32))
    cap = 8192 if n <= 1900000 else 16384
    biggest = max(big, int(np.bincount(d8[:n_real], minlength=256).max()))
    assert int(download(cq, flag, np.uint32, 1)[0]) == (big if big > cap else 0) or biggest > cap

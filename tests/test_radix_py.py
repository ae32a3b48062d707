"""RadixSorter on the GPU; mirrors tests/test_radix_py.py:83-201 (test_sorter, test_sorter_resized,
test_arg_sorter) and adds the BASELINE config-5 key distributions at larger sizes."""
import numpy as np
import pytest

from collision_amd import hip
from collision_amd.radix import PrefixScanProgram, RadixProgram, RadixSorter
from tests.util import download, upload

pytestmark = pytest.mark.gpu

VALUE_DTYPES = [np.dtype("uint32"), np.dtype("float64"), np.dtype(("float64", 3)), np.dtype(("float64", 4)),
                np.dtype(("float32", 3)),
                # widths the scatter passes do not move themselves (sorted as (key, index), gathered once):
                # the reference specialises for any NumPy dtype (radix.py:16-25)
                np.dtype("uint8"), np.dtype("int16"), np.dtype(("float64", 8)), np.dtype(("float64", 16)),
                np.dtype(("float32", 16))]


def _sort(ctx, cq, sorter, keys, values=None, value_dtype=None):
    keys_buf = upload(ctx, keys)
    out_keys = hip.Buffer(ctx, keys.nbytes)
    if values is None:
        e = sorter.sort(cq, keys_buf, out_keys)
        return download(cq, out_keys, keys.dtype, keys.shape, wait_for=[e]), None, download(cq, keys_buf, keys.dtype)
    if value_dtype.shape == (3,):                       # vec3 values are 4 wide on the device
        dev = np.zeros((len(values), 4), dtype=values.dtype)
        dev[:, :3] = values
    else:
        dev = values
    vals_buf = upload(ctx, dev)
    out_vals = hip.Buffer(ctx, dev.nbytes)
    e = sorter.sort(cq, keys_buf, out_keys, vals_buf, out_vals)
    k = download(cq, out_keys, keys.dtype, keys.shape, wait_for=[e])
    v = download(cq, out_vals, dev.dtype, dev.shape)
    if value_dtype.shape == (3,):
        v = v[:, :3]
    return k, v, download(cq, keys_buf, keys.dtype)


@pytest.mark.parametrize("key_dtype", ["uint32", "uint64"])
@pytest.mark.parametrize("size,group_size", [(32, 8), (15360, 32), (32, 16)])
def test_sorter(hip_env, key_dtype, size, group_size):
    ctx, cq = hip_env
    sorter = RadixSorter(ctx, size, group_size, key_dtype=key_dtype, program=RadixProgram(ctx, key_dtype),
                         scan_program=PrefixScanProgram(ctx))
    data = np.random.RandomState(4).randint(500, size=size).astype(key_dtype)
    out, _, untouched = _sort(ctx, cq, sorter, data)
    np.testing.assert_equal(out, np.sort(data))
    np.testing.assert_equal(untouched, data)            # inputs are left alone (see radix.py docstring)


@pytest.mark.parametrize("key_dtype", ["uint32", "uint64"])
@pytest.mark.parametrize("old_shape,new_shape", [((15360, 32), (32, 8)), ((32, 8), (15360, 32))])
def test_sorter_resized(hip_env, key_dtype, old_shape, new_shape):
    ctx, cq = hip_env
    sorter = RadixSorter(ctx, *old_shape, key_dtype=key_dtype, program=RadixProgram(ctx, key_dtype))
    rs = np.random.RandomState(4)
    _sort(ctx, cq, sorter, rs.randint(500, size=old_shape[0]).astype(key_dtype))
    sorter.resize(*new_shape)
    data = rs.randint(500, size=new_shape[0]).astype(key_dtype)
    out, _, _ = _sort(ctx, cq, sorter, data)
    np.testing.assert_equal(out, np.sort(data))


@pytest.mark.parametrize("key_dtype", ["uint32", "uint64"])
@pytest.mark.parametrize("value_dtype", VALUE_DTYPES, ids=str)
@pytest.mark.parametrize("size,group_size", [(32, 8), (15360, 32)])
def test_arg_sorter(hip_env, key_dtype, value_dtype, size, group_size):
    ctx, cq = hip_env
    sorter = RadixSorter(ctx, size, group_size, key_dtype=key_dtype, value_dtype=value_dtype,
                         program=RadixProgram(ctx, key_dtype, value_dtype))
    rs = np.random.RandomState(4)
    keys = rs.randint(500, size=size).astype(key_dtype)
    values = rs.uniform(-1000, 1000, size=(size,) + value_dtype.shape).astype(value_dtype.base)
    k, v, _ = _sort(ctx, cq, sorter, keys, values, value_dtype)
    np.testing.assert_equal(k, np.sort(keys))
    np.testing.assert_equal(v, values[np.argsort(keys, kind="mergesort")])       # stability


def test_keep_sorted_inputs_matches_reference_postcondition(hip_env):
    # radix.py:158-169: after sort() the reference's input buffers hold the sorted data too
    ctx, cq = hip_env
    sorter = RadixSorter(ctx, 4096, 64)
    sorter.keep_sorted_inputs = True
    rs = np.random.RandomState(4)
    keys = rs.randint(0, 2 ** 32, size=4096, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(4096, dtype=np.uint32)
    keys_buf, vals_buf = upload(ctx, keys), upload(ctx, vals)
    out_k, out_v = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, vals.nbytes)
    e = sorter.sort(cq, keys_buf, out_k, vals_buf, out_v)
    order = np.argsort(keys, kind="mergesort")
    for buf, expect in ((keys_buf, keys[order]), (out_k, keys[order]), (vals_buf, vals[order]), (out_v, vals[order])):
        np.testing.assert_equal(download(cq, buf, np.uint32, wait_for=[e]), expect)


def _config5_keys(kind, n, dtype):
    rs = np.random.RandomState(4)
    if kind == "morton30":
        return rs.randint(0, 2 ** 30, size=n).astype(dtype)
    if kind == "full":
        bits = 8 * np.dtype(dtype).itemsize
        return rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(dtype) if bits == 32 else \
            (rs.randint(0, 2 ** 32, size=n, dtype=np.uint64) << np.uint64(32) | rs.randint(0, 2 ** 32, size=n, dtype=np.uint64))
    if kind == "arange":
        return np.arange(n, dtype=dtype)
    if kind == "reversed":
        return np.arange(n, dtype=dtype)[::-1].copy()
    if kind == "few":
        return rs.randint(0, 1000, size=n).astype(dtype)       # tests/benchmarks/test_radix.py:51-55
    if kind == "constant":
        return np.full(n, 0xDEADBEEF, dtype=dtype)
    if kind == "allones":
        return np.full(n, np.iinfo(dtype).max, dtype=dtype)
    raise ValueError(kind)


@pytest.mark.parametrize("key_dtype", ["uint32", "uint64"])
@pytest.mark.parametrize("kind", ["morton30", "full", "arange", "reversed", "few", "constant", "allones"])
@pytest.mark.parametrize("size,group_size", [(307200, 128), (1000448, 256)])
def test_key_distributions_with_ids(hip_env, key_dtype, kind, size, group_size):
    ctx, cq = hip_env
    sorter = RadixSorter(ctx, size, group_size, key_dtype=key_dtype, program=RadixProgram(ctx, key_dtype))
    keys = _config5_keys(kind, size, key_dtype)
    vals = np.arange(size, dtype=np.uint32)
    k, v, _ = _sort(ctx, cq, sorter, keys, vals, np.dtype("uint32"))
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(k, keys[order])
    np.testing.assert_array_equal(v, order.astype(np.uint32))


def test_large_sort_properties(hip_env):
    """16 Mi random 32-bit keys + ids: sortedness, permutation, stability, key/value pairing."""
    ctx, cq = hip_env
    n = 1 << 24
    sorter = RadixSorter(ctx, n, 256)
    keys = np.random.RandomState(4).randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    k, v, _ = _sort(ctx, cq, sorter, keys, np.arange(n, dtype=np.uint32), np.dtype("uint32"))
    assert (np.diff(k.astype(np.int64)) >= 0).all()
    np.testing.assert_array_equal(keys[v], k)
    eq = k[1:] == k[:-1]
    assert (v[1:][eq] > v[:-1][eq]).all()
    assert (np.bincount(v, minlength=n) == 1).all()


def test_sizes_around_the_tile(hip_env):
    """The C ABI takes any n (the 2*group_size rule is the Python class's); ragged tails."""
    from collision_amd._lib import call
    ctx, cq = hip_env
    tile = call.col_radix_tile(1000, 4, 4)             # small inputs use the small tile ...
    mid, big = 1 << 20, 8 << 20                        # ... then 4096 pairs from here, 8192 from there
    assert tile < call.col_radix_tile(mid, 4, 4) < call.col_radix_tile(big, 4, 4)
    assert call.col_radix_tile(mid - 1, 4, 4) == tile and call.col_radix_tile(big - 1, 4, 4) == call.col_radix_tile(mid, 4, 4)
    assert call.col_radix_tile(big, 8, 4) == call.col_radix_tile(mid, 4, 4)      # 8-byte keys stop at the middle tile
    rs = np.random.RandomState(4)
    for n in (1, 2, 63, 64, 65, tile - 1, tile, tile + 1, 3 * tile + 17, mid - 1, mid, mid + 4097, big - 1, big,
              big + 8193):
        keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        kb, vb = upload(ctx, keys), upload(ctx, vals)
        ko, vo = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, vals.nbytes)
        scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
        call.col_radix_sort(cq.stream, kb.ptr, ko.ptr, vb.ptr, vo.ptr, n, 4, 4, scratch.ptr, 0)
        order = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(download(cq, ko, np.uint32, n), keys[order])
        np.testing.assert_array_equal(download(cq, vo, np.uint32, n), order.astype(np.uint32))


@pytest.mark.parametrize("n", [1000, 4099, 70001, (1 << 20) + 3, (8 << 20) + 3, (16 << 20) + 3])
def test_constant_and_blocky_keys_with_ragged_tail(hip_env, n):
    """Keys with one digit per wave (constant, or long constant runs) and a ragged last tile: the
    histogram's wave-uniform shortcut must count only the lanes that are in range."""
    from collision_amd._lib import call
    ctx, cq = hip_env
    for keys in (np.full(n, 3, np.uint32), (np.arange(n, dtype=np.uint32) // 777) % 5, np.zeros(n, np.uint32)):
        keys = keys.astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        kb, vb = upload(ctx, keys), upload(ctx, vals)
        ko, vo = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, vals.nbytes)
        scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
        call.col_radix_sort(cq.stream, kb.ptr, ko.ptr, vb.ptr, vo.ptr, n, 4, 4, scratch.ptr, 0)
        order = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(download(cq, ko, np.uint32, n), keys[order])
        np.testing.assert_array_equal(download(cq, vo, np.uint32, n), order.astype(np.uint32))


@pytest.mark.parametrize("key_dtype,val_bytes", [("uint32", 4), ("uint64", 4), ("uint64", 8), ("uint32", 8), ("uint32", 16),
                                                 ("uint64", 32), ("uint32", 0), ("uint64", 0), ("uint32", 1), ("uint64", 2),
                                                 ("uint32", 64), ("uint64", 128)])
@pytest.mark.parametrize("n", [(1 << 20) + 12345, (16 << 20) + 12345])
def test_big_tiles_all_type_combinations(hip_env, key_dtype, val_bytes, n):
    """From 1 Mi elements the sort uses the 4096-pair tile, from 8 Mi the 8192-pair one (4-byte
    keys): every key/value width once in each."""
    from collision_amd._lib import call
    ctx, cq = hip_env
    rs = np.random.RandomState(4)
    kbytes = np.dtype(key_dtype).itemsize
    keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)
    if kbytes == 8:
        keys = (keys << np.uint64(17)) ^ rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)
    keys = keys.astype(key_dtype)
    keys[::7] = keys[3]                                  # duplicates: stability matters
    vals = rs.randint(0, 255, size=(n, max(val_bytes, 1)), dtype=np.uint8)
    kb, vb = upload(ctx, keys), upload(ctx, vals)
    ko, vo = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, vals.nbytes)
    scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, kbytes, val_bytes))
    call.col_radix_sort(cq.stream, kb.ptr, ko.ptr, vb.ptr if val_bytes else None, vo.ptr if val_bytes else None, n, kbytes,
                        val_bytes, scratch.ptr, 0)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(download(cq, ko, key_dtype, n), keys[order])
    if val_bytes:
        np.testing.assert_array_equal(download(cq, vo, np.uint8, vals.shape), vals[order])


@pytest.mark.parametrize("key_dtype", ["uint8", "uint16"])
@pytest.mark.parametrize("value_dtype", [None, np.dtype("uint32"), np.dtype(("float32", 4)), np.dtype("uint8"), np.dtype(("float64", 8))])
@pytest.mark.parametrize("n", [512, 200 * 1024])
def test_narrow_keys(hip_env, key_dtype, value_dtype, n):
    """uint8 / uint16 keys (radix.py:16-20 accepts every unsigned dtype): widened to u32 on the device, one 8-bit pass
    per key byte, narrowed on the way out; values of every width follow stably."""
    ctx, cq = hip_env
    rs = np.random.RandomState(4)
    keys = rs.randint(0, 2 ** (8 * np.dtype(key_dtype).itemsize), size=n).astype(key_dtype)
    sorter = RadixSorter(ctx, n, 64, 4, key_dtype=key_dtype, value_dtype=value_dtype or np.dtype("uint32"))
    assert sorter.num_passes == 2 * np.dtype(key_dtype).itemsize
    kb, ko = upload(ctx, keys), hip.Buffer(ctx, keys.nbytes)
    order = np.argsort(keys, kind="stable")
    if value_dtype is None:
        e = sorter.sort(cq, kb, ko)
        np.testing.assert_array_equal(download(cq, ko, key_dtype, n, wait_for=[e]), keys[order])
        return
    vals = rs.randint(0, 255, size=(n, value_dtype.itemsize), dtype=np.uint8)
    vb, vo = upload(ctx, vals), hip.Buffer(ctx, vals.nbytes)
    e = sorter.sort(cq, kb, ko, vb, vo)
    np.testing.assert_array_equal(download(cq, ko, key_dtype, n, wait_for=[e]), keys[order])
    np.testing.assert_array_equal(download(cq, vo, np.uint8, vals.shape), vals[order])
    np.testing.assert_array_equal(download(cq, kb, key_dtype, n), keys)          # inputs untouched


def test_huge_tile_kernel(hip_env):
    """The 16384-key tile (u32 keys WITHOUT values; automatic from 32 Mi keys) has a kernel of its own
    (k_scatter_huge) and a separate instance for the input's partial last tile: forced here on small inputs --
    only a partial tile, exact multiples, ragged tails, constant and blocky keys -- and at its automatic size.
    Sorts with values keep the 8192-pair tile (a forced 16384 falls back to it)."""
    from collision_amd._lib import call, cdll
    ctx, cq = hip_env
    huge = 32 << 20
    assert call.col_radix_tile(huge - 1, 4, 0) == 8192 and call.col_radix_tile(huge, 4, 0) == 16384
    assert call.col_radix_tile(huge, 4, 4) == 8192 and call.col_radix_tile(huge, 4, 1) == 8192 and call.col_radix_tile(huge, 8, 0) == 4096
    rs = np.random.RandomState(11)

    def check(keys, with_values=False):
        n = len(keys)
        vals = np.arange(n, dtype=np.uint32)
        kb, vb = upload(ctx, keys), upload(ctx, vals)
        ko, vo = hip.Buffer(ctx, keys.nbytes), hip.Buffer(ctx, vals.nbytes)
        vbytes = 4 if with_values else 0
        scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, vbytes))
        call.col_radix_sort(cq.stream, kb.ptr, ko.ptr, vb.ptr if with_values else None, vo.ptr if with_values else None,
                            n, 4, vbytes, scratch.ptr, 0)
        out = download(cq, ko, np.uint32, n)
        if with_values:
            _check_stable_sort(keys, out, download(cq, vo, np.uint32, n))
        else:
            np.testing.assert_array_equal(out, np.sort(keys))

    assert cdll().col_debug_radix_tile(16384) == 0
    try:
        assert call.col_radix_tile(1000, 4, 0) == 16384 and call.col_radix_tile(1000, 4, 4) == 8192
        for n in (1, 100, 16383, 16384, 16385, 3 * 16384, 3 * 16384 + 77, 40 * 16384 + 16383, 1000003):
            check(rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32))
        n = 5 * 16384 + 4001
        for keys in (np.full(n, 3, np.uint32), (np.arange(n, dtype=np.uint32) // 777) % 5, np.zeros(n, np.uint32),
                     np.arange(n, dtype=np.uint32)[::-1].copy(), rs.randint(0, 2 ** 30, size=n).astype(np.uint32)):
            check(keys.astype(np.uint32))
        check(rs.randint(0, 2 ** 32, size=100001, dtype=np.uint64).astype(np.uint32), with_values=True)
    finally:
        cdll().col_debug_radix_tile(0)
    for n in (huge, huge + 5):                                     # automatic: exact multiple, 5-key last tile
        k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
        k[::7] = k[3]
        check(k)


def _check_stable_sort(keys, out_keys, out_vals):
    """out_vals must be THE stable argsort of keys (values were arange), out_keys the sorted keys -- checked
    without an argsort: a permutation that gathers the sorted keys and ascends inside runs of equal keys."""
    n = len(keys)
    np.testing.assert_array_equal(out_keys, np.sort(keys))
    assert (np.bincount(out_vals, minlength=n) == 1).all()                  # a permutation of arange(n)
    np.testing.assert_array_equal(keys[out_vals], out_keys)                  # values travelled with their keys
    same = out_keys[1:] == out_keys[:-1]
    assert (out_vals[1:][same] > out_vals[:-1][same]).all()                  # stable


@pytest.mark.parametrize("kind", ["morton30", "uniform32", "arange"])
@pytest.mark.parametrize("with_values", [True, False])
def test_config5_full_size(hip_env, kind, with_values):
    """BASELINE config 5 at full size: 64 Mi uint32 keys (30-bit Morton-like, full 32-bit uniform, already
    sorted -- tests/benchmarks/test_radix.py:51-55 shapes) with uint32 ids, and key-only; complete check
    (radix.py:118-170: sorted keys, values follow their keys stably)."""
    from collision_amd._lib import call
    ctx, cq = hip_env
    n = 1 << 26
    rng = np.random.RandomState(4)
    if kind == "morton30":
        keys = rng.randint(0, 2 ** 30, size=n).astype(np.uint32)
    elif kind == "uniform32":
        keys = rng.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    else:
        keys = np.arange(n, dtype=np.uint32)
    kb = upload(ctx, keys)
    ko = hip.Buffer(ctx, n * 4)
    vb = upload(ctx, np.arange(n, dtype=np.uint32)) if with_values else None
    vo = hip.Buffer(ctx, n * 4) if with_values else None
    scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4 if with_values else 0))
    call.col_radix_sort(cq.stream, kb.ptr, ko.ptr, vb.ptr if with_values else None, vo.ptr if with_values else None,
                        n, 4, 4 if with_values else 0, scratch.ptr, 0)
    out_keys = download(cq, ko, np.uint32, n)
    if with_values:
        _check_stable_sort(keys, out_keys, download(cq, vo, np.uint32, n))
    else:
        np.testing.assert_array_equal(out_keys, np.sort(keys))
    np.testing.assert_array_equal(download(cq, kb, np.uint32, n), keys)     # inputs untouched (copy_back = 0)

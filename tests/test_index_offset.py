"""Indexer and OffsetFinder on the GPU; mirrors tests/test_index.py:18-79 and
tests/test_offset_py.py:23-62 of the reference, plus an oracle comparison on larger inputs."""
import numpy as np
import pytest

from collision_amd import hip
from collision_amd.index import Indexer, IndexProgram
from collision_amd.offset import OffsetFinder, OffsetProgram
from tests.util import download, upload

pytestmark = pytest.mark.gpu

VALUE_DTYPES = [np.dtype("uint32"), np.dtype(("float64", 2)), np.dtype(("float32", 4))]
INDEX_DTYPES = [np.dtype("uint32"), np.dtype("uint64"), np.dtype("uint8"), np.dtype("uint16")]     # index.py:13-17: any unsigned dtype


@pytest.mark.parametrize("value_dtype", VALUE_DTYPES, ids=str)
@pytest.mark.parametrize("index_dtype", INDEX_DTYPES, ids=str)
def test_gather(hip_env, value_dtype, index_dtype):
    ctx, cq = hip_env
    size, nindices = 240, 30
    rs = np.random.RandomState(4)
    indexer = Indexer(ctx, value_dtype, index_dtype)
    values = rs.uniform(0, 1000, (size,) + value_dtype.shape).astype(value_dtype.base)
    indices = rs.choice(size, size=nindices, replace=False).astype(index_dtype)
    values_buf, index_buf = upload(ctx, values), upload(ctx, indices)
    out_buf = hip.Buffer(ctx, nindices * value_dtype.itemsize)
    e = indexer.gather(cq, nindices, values_buf, index_buf, out_buf)
    out = download(cq, out_buf, value_dtype.base, (nindices,) + value_dtype.shape, wait_for=[e])
    np.testing.assert_equal(out, values[indices])


@pytest.mark.parametrize("value_dtype", VALUE_DTYPES, ids=str)
@pytest.mark.parametrize("index_dtype", INDEX_DTYPES, ids=str)
def test_scatter(hip_env, value_dtype, index_dtype):
    ctx, cq = hip_env
    size, nindices = 240, 30
    rs = np.random.RandomState(4)
    indexer = Indexer(ctx, value_dtype, index_dtype)
    values = rs.uniform(0, 1000, (nindices,) + value_dtype.shape).astype(value_dtype.base)
    indices = rs.choice(size, size=nindices, replace=False).astype(index_dtype)
    values_buf, index_buf = upload(ctx, values), upload(ctx, indices)
    out_buf = hip.Buffer(ctx, size * value_dtype.itemsize)
    e = hip.enqueue_fill_buffer(cq, out_buf, np.full(1, 1.0, value_dtype.base), 0, size * value_dtype.itemsize)
    e = indexer.scatter(cq, nindices, values_buf, index_buf, out_buf, wait_for=[e])
    out = download(cq, out_buf, value_dtype.base, (size,) + value_dtype.shape, wait_for=[e])
    selection = np.zeros(size, dtype=bool)
    selection[indices] = True
    np.testing.assert_equal(out[indices], values)
    np.testing.assert_equal(out[~selection], 1.0)


def test_index_program_errs(hip_env):
    ctx, cq = hip_env
    with pytest.raises(ValueError):
        IndexProgram(ctx, "uint32", "int32")
    with pytest.raises(ValueError):
        Indexer(ctx, "uint32", "uint32", program=IndexProgram(ctx, "uint32", "uint64"))


@pytest.mark.parametrize("value_dtype", ["uint32", "uint64", "uint8", "uint16"])          # offset.py:14-19: any unsigned dtype
@pytest.mark.parametrize("offset_dtype", ["uint32", "uint64", "uint8", "uint16"])
def test_offset_goldens(hip_env, generated_meta, value_dtype, offset_dtype):
    ctx, cq = hip_env
    lits = generated_meta["offset_literals"]        # tests/test_offset_py.py:27-28,48-49
    finder = OffsetFinder(ctx, value_dtype, offset_dtype, OffsetProgram(ctx, value_dtype, offset_dtype))
    for values, expected, n_offsets in ((lits[0], lits[1], max(lits[0]) + 2), (lits[2], lits[3], 7)):
        values = np.array(values, dtype=value_dtype)
        values_buf = upload(ctx, values)
        out_buf = hip.Buffer(ctx, len(expected) * np.dtype(offset_dtype).itemsize)
        e = finder.find_offsets(cq, values_buf, len(values), out_buf, n_offsets)
        np.testing.assert_equal(download(cq, out_buf, offset_dtype, len(expected), wait_for=[e]), expected)


def test_offsets_match_oracle_and_searchsorted(hip_env, oracle):
    # tests/benchmarks/test_offset.py:24-39 shape: 2^21 sorted values
    ctx, cq = hip_env
    rs = np.random.RandomState(4)
    for maxval in (2000, 2000000):
        values = np.sort(rs.randint(0, maxval, size=1 << 21).astype(np.uint32))
        n_offsets = int(values.max()) + 2
        out_buf = hip.Buffer(ctx, n_offsets * 4)
        values_buf = upload(ctx, values)
        e = OffsetFinder(ctx).find_offsets(cq, values_buf, len(values), out_buf, n_offsets)
        out = download(cq, out_buf, np.uint32, n_offsets, wait_for=[e])
        np.testing.assert_array_equal(out, oracle.find_offsets(values, n_offsets))
        np.testing.assert_array_equal(out, np.searchsorted(values, np.arange(n_offsets), side="left"))


@pytest.mark.parametrize("value_dtype", ["uint8", "uint16"])
def test_narrow_offsets_reach_the_type_maximum(hip_env, value_dtype):
    """Values up to the type's maximum (the reference's own comment warns of the wrap at a == b == VALUE_TYPE_MAX:
    the loop counter is 64 bits wide here)."""
    ctx, cq = hip_env
    top = int(np.iinfo(value_dtype).max)
    rs = np.random.RandomState(4)
    values = np.sort(rs.randint(0, top + 1, size=5000)).astype(value_dtype)
    values[-3:] = top
    n_offsets = top + 1
    out_buf = hip.Buffer(ctx, n_offsets * 4)
    e = OffsetFinder(ctx, value_dtype, "uint32").find_offsets(cq, upload(ctx, values), len(values), out_buf, n_offsets)
    out = download(cq, out_buf, np.uint32, n_offsets, wait_for=[e])
    np.testing.assert_array_equal(out, np.searchsorted(values, np.arange(n_offsets), side="left"))

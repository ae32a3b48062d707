"""Bounds / Summer reducers on the GPU; mirrors tests/test_bounds_py.py:18-148 and
tests/test_summer_py.py."""
import numpy as np
import pytest

from collision_amd import hip
from collision_amd.bounds import Bounds, BoundsProgram
from collision_amd.misc import dtype_sizeof
from collision_amd.summer import Summer
from tests.util import download, upload

pytestmark = pytest.mark.gpu

COORD_DTYPES = [np.dtype(("float32", 3)), np.dtype(("float64", 4)), np.dtype("float32")]


def _device_dtype(coord_dtype):
    return np.dtype((coord_dtype.base, 4)) if coord_dtype.shape == (3,) else coord_dtype


def _reduce(ctx, cq, reducer, coord_dtype, values):
    value_dtype = _device_dtype(coord_dtype)
    out_buf = hip.Buffer(ctx, 2 * dtype_sizeof(coord_dtype))
    e = reducer.reduce(cq, len(values), upload(ctx, values), out_buf)
    out = download(cq, out_buf, value_dtype.base, (2,) + value_dtype.shape, wait_for=[e])
    expected = np.stack([values.min(axis=0), values.max(axis=0)])
    if coord_dtype.shape == (3,):
        out, expected = out[..., :3], expected[..., :3]
    np.testing.assert_equal(out, expected)


@pytest.mark.parametrize("coord_dtype", COORD_DTYPES, ids=str)
def test_negative_bounds(hip_env, coord_dtype):
    ctx, cq = hip_env
    vd = _device_dtype(coord_dtype)
    values = np.random.RandomState(4).normal(-10, 1, size=(24,) + vd.shape).astype(vd.base).clip(max=-1.0)
    _reduce(ctx, cq, Bounds(ctx, 2, 4, coord_dtype, BoundsProgram(ctx, coord_dtype)), coord_dtype, values)


@pytest.mark.parametrize("coord_dtype", COORD_DTYPES, ids=str)
@pytest.mark.parametrize("size,ngroups,group_size", [(24, 2, 4), (100, 4, 8), (100, 5, 8), (1536000, 64, 128)])
def test_bounds(hip_env, coord_dtype, size, ngroups, group_size):
    ctx, cq = hip_env
    vd = _device_dtype(coord_dtype)
    values = np.random.RandomState(4).normal(scale=1e8, size=(size,) + vd.shape).astype(vd.base)
    _reduce(ctx, cq, Bounds(ctx, ngroups, group_size, coord_dtype), coord_dtype, values)


@pytest.mark.parametrize("coord_dtype", COORD_DTYPES, ids=str)
def test_bounds_resized(hip_env, coord_dtype):
    ctx, cq = hip_env
    reducer = Bounds(ctx, 2, 4, coord_dtype)
    reducer.resize(4, 8)
    vd = _device_dtype(coord_dtype)
    values = np.random.RandomState(4).normal(size=(100,) + vd.shape).astype(vd.base)
    _reduce(ctx, cq, reducer, coord_dtype, values)


@pytest.mark.parametrize("value_dtype", [np.dtype(("float64", 3)), np.dtype(("uint32", 4)), np.dtype("int32")], ids=str)
def test_summer(hip_env, value_dtype):
    ctx, cq = hip_env
    vd = _device_dtype(value_dtype)
    rs = np.random.RandomState(4)
    values = rs.randint(0, 100, size=(1000,) + vd.shape).astype(vd.base)     # integer-valued: sums are exact
    out_buf = hip.Buffer(ctx, dtype_sizeof(value_dtype))
    e = Summer(ctx, 4, 8, value_dtype).reduce(cq, len(values), upload(ctx, values), out_buf)
    out = download(cq, out_buf, vd.base, vd.shape or (1,), wait_for=[e])
    expected = values.sum(axis=0)
    if value_dtype.shape == (3,):
        out, expected = out[:3], expected[:3]
    np.testing.assert_equal(out.reshape(-1), np.asarray(expected).reshape(-1))

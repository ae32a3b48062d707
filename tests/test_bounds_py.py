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


@pytest.mark.parametrize("value_dtype", [np.dtype("float32"), np.dtype(("float64", 4)), np.dtype(("int32", 2)), np.dtype("uint64")], ids=str)
def test_generic_accumulator_lists(hip_env, value_dtype):
    """reduce.py:9-22 renders ANY list of (init, fn) accumulators into its kernel template; beyond the two
    lists the package itself uses (Bounds, Summer) that is the table-driven col_reduce_list: here
    [sum, min, max, product-of-ones] in one pass, each accumulator one output row (reduce.cl:34-37)."""
    from collision_amd.reduce import ReductionProgram, Reducer
    ctx, cq = hip_env
    vd = _device_dtype(value_dtype)

    class StatsProgram(ReductionProgram):
        accumulator = [("0", "ADD"), ("INFINITY", "fmin" if vd.base.kind == "f" else "min"), ("-INFINITY", "max"), ("1", "MUL")]

    class Stats(Reducer):
        program_type = StatsProgram

    rs = np.random.RandomState(4)
    n = 100003
    values = rs.randint(1, 16, size=(n,) + vd.shape).astype(vd.base)           # small integers: float32 sums stay exact
    ones = np.ones_like(values)
    for data, prod in ((values, None), (ones, 1)):
        out_buf = hip.Buffer(ctx, 4 * dtype_sizeof(value_dtype))
        e = Stats(ctx, 8, 64, value_dtype).reduce(cq, n, upload(ctx, data), out_buf)
        out = download(cq, out_buf, vd.base, (4,) + (vd.shape or (1,)), wait_for=[e])
        d2 = data.reshape(n, -1)
        np.testing.assert_equal(out[0].reshape(-1), d2.sum(axis=0))
        np.testing.assert_equal(out[1].reshape(-1), d2.min(axis=0))
        np.testing.assert_equal(out[2].reshape(-1), d2.max(axis=0))
        if prod is not None:
            np.testing.assert_equal(out[3].reshape(-1), np.full(d2.shape[1], prod, vd.base))


def test_generic_accumulator_errors(hip_env):
    """A list the compiler cannot make sense of is a ValueError that carries its messages (the reference fails in
    pyopencl's build the same way); so is a geometry whose accumulators do not fit a group's local memory."""
    from collision_amd.reduce import ReductionProgram, Reducer
    ctx, cq = hip_env

    class Bad(ReductionProgram):
        accumulator = [("0", "no_such_function")]

    with pytest.raises(ValueError, match="no_such_function"):
        Bad(ctx, np.dtype("float32"))

    class Wide(ReductionProgram):
        accumulator = [("0", "hypot")] * 40

    class WideReducer(Reducer):
        program_type = Wide

    r = WideReducer(ctx, 8, 1024, np.dtype(("float64", 4)))                 # 40 x 4 x 8 bytes x 1024 work-items
    with pytest.raises(ValueError, match="too large"):
        r.reduce(cq, 10, upload(ctx, np.zeros((10, 4))), hip.Buffer(ctx, 40 * 32))


RENDERED_LISTS = {
    # float ADD on arbitrary floats: the sum depends on the order -- the reference's order is the contract
    "float32": [("0", "ADD"), ("INFINITY", "fmin"), ("-INFINITY", "fmax"), ("1", "copysign"), ("0", "maxmag"),
                ("INFINITY", "minmag"), ("0", "fdim")],
    "float64": [("0", "ADD"), ("-INFINITY", "max"), ("INFINITY", "min"), ("0", "maxmag"), ("1", "copysign")],
    "uint8": [("0", "add_sat"), ("0", "ADD"), ("0", "hadd"), ("255", "min"), ("0", "rhadd"), ("0", "abs_diff"), ("7", "mul_hi")],
    "int16": [("SHRT_MAX", "min"), ("0", "add_sat"), ("0", "sub_sat"), ("-3", "mul_hi"), ("0", "abs_diff")],
    "int64": [("LONG_MIN", "max"), ("0", "add_sat"), ("0x7fffffffffffffff", "min"), ("0", "rhadd"), ("0", "hadd")],
    "uint32": [("0", "ADD"), ("0", "max"), ("UINT_MAX", "min"), ("0", "rhadd"), ("0", "sub_sat")],
}


@pytest.mark.parametrize("geometry", [(8, 64), (5, 48), (1, 1), (16, 256)], ids=str)
@pytest.mark.parametrize("value_dtype", [np.dtype("float32"), np.dtype(("float64", 3)), np.dtype(("uint8", 4)), np.dtype("int16"),
                                         np.dtype(("int64", 2)), np.dtype("uint32")], ids=str)
def test_rendered_accumulator_lists(hip_env, oracle, value_dtype, geometry):
    """reduce.py:9-22: ANY accumulator list.  What the compiled-in kernels and the table-driven reducer do not cover is
    rendered into HIP with the structure of reduce.cl and compiled at run time (collision_amd/reduce.py render_source,
    csrc/rtc_reduce.hip); it runs on the caller's ngroups x group_size work-items, so every function -- also one that is
    not associative, like a float ADD on arbitrary values -- gives the bits of the reference's order of operations
    (oracle.reduce_list restates it).  Geometries that are not powers of two fold every partial."""
    from collision_amd.reduce import ReductionProgram, Reducer
    ctx, cq = hip_env
    vd = _device_dtype(value_dtype)
    acc = RENDERED_LISTS[vd.base.name]
    program = type("P", (ReductionProgram,), {"accumulator": acc})
    reducer = type("R", (Reducer,), {"program_type": program})
    assert program(ctx, value_dtype).rtc is not None                         # not one of the table's lists
    ngroups, group_size = geometry
    rs = np.random.RandomState(5)
    for n in (0, 1, 777, 100003):
        shape = (n,) + (vd.shape or (1,))
        if vd.base.kind == "f":
            values = ((rs.random_sample(shape) - 0.5) * 8).astype(vd.base)
        else:
            info = np.iinfo(vd.base)
            values = rs.randint(max(info.min, -(1 << 40)), min(info.max, 1 << 40), size=shape, dtype=np.int64).astype(vd.base)
        out_buf = hip.Buffer(ctx, len(acc) * dtype_sizeof(value_dtype))
        src = upload(ctx, values) if n else hip.Buffer(ctx, 16)
        e = reducer(ctx, ngroups, group_size, value_dtype).reduce(cq, n, src, out_buf)
        out = download(cq, out_buf, vd.base, (len(acc),) + (vd.shape or (1,)), wait_for=[e])
        want = oracle.reduce_list(values.reshape(n, shape[1]), acc, ngroups, group_size)
        if value_dtype.shape == (3,):                                        # the fourth lane of a 3-vector is padding
            out, want = out[:, :3], want[:, :3]
        np.testing.assert_array_equal(out.view(np.uint8), want.view(np.uint8), err_msg="n=%d" % n)


def test_integer_initial_values_are_exact(hip_env):
    """Initial values of integer lists beyond 2^53 (a double cannot carry them): UINT64_MAX as the start of a min
    list, 2^63 + 1 as the start of a max list over smaller data; a negative start for an unsigned dtype is refused.
    (An initial value is the start of EVERY work-item's accumulator, reduce.cl:9-11, so only min / max lists give a
    geometry-independent meaning to one that is not the identity; ADD starts from a zero spelt in hex here.)"""
    from collision_amd.reduce import ReductionProgram, Reducer
    ctx, cq = hip_env
    big = (1 << 63) + 1

    class LimitsProgram(ReductionProgram):
        accumulator = [(str((1 << 64) - 1), "min"), (str(big), "max"), ("0x0", "ADD")]

    class Limits(Reducer):
        program_type = LimitsProgram

    values = np.array([(1 << 64) - 5, (1 << 63) - 1, 12345], dtype=np.uint64)
    out_buf = hip.Buffer(ctx, 3 * 8)
    e = Limits(ctx, 8, 64, np.dtype("uint64")).reduce(cq, len(values), upload(ctx, values), out_buf)
    out = download(cq, out_buf, np.uint64, 3, wait_for=[e])
    assert int(out[0]) == 12345 and int(out[1]) == (1 << 64) - 5 and int(out[2]) == int(values.sum(dtype=np.uint64))
    e = Limits(ctx, 8, 64, np.dtype("uint64")).reduce(cq, 2, upload(ctx, values[1:]), out_buf)      # max stays at its start
    out = download(cq, out_buf, np.uint64, 3, wait_for=[e])
    assert int(out[1]) == big

    class Negative(ReductionProgram):
        accumulator = [("-1", "max")]

    with pytest.raises(ValueError):
        Negative(ctx, np.dtype("uint32"))
    Negative(ctx, np.dtype("int32"))          # fine for a signed dtype

"""CPU oracle for the collision hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the timed CPU baseline.
The product (``collision_amd``) never imports it.

Two restatements live here:

* ``collision_oracle.c`` (``liboracle.so``, built by ``oracle/Makefile``): every
  stage of the path, each function citing the reference file:line it follows.
* :func:`find_collisions` below: NumPy restatement of the reference's own
  brute-force test oracle (``/root/reference/tests/test_collision_py.py:30-37``).

Parity status: PINNED -- ``tests/test_oracle_golden.py`` checks both against
the literal vectors of the reference's tests and against outputs of the
reference's ``find_collisions`` (``tests/golden/make_golden.py``).
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_here = Path(__file__).resolve().parent
_lib = None

# collision.py:9
Node = np.dtype([('parent', 'uint32'), ('right_edge', 'uint32'), ('data', 'uint32', 2)])
NO_NODE = 0xFFFFFFFF


def build():
    """Compile liboracle.so with gcc (building the checker is not using it)."""
    subprocess.run(["make", "-s", "-C", str(_here)], check=True)


def lib():
    global _lib
    if _lib is None:
        so = _here / "liboracle.so"
        if not so.exists():
            build()
        _lib = C.CDLL(str(so))
        _lib.orc_traverse_f32.restype = C.c_uint64
        _lib.orc_traverse_f64.restype = C.c_uint64
        _lib.orc_brute_force_f32.restype = C.c_uint64
        _lib.orc_brute_force_f64.restype = C.c_uint64
        _lib.orc_collide_f32.restype = C.c_int64
        _lib.orc_collide_f64.restype = C.c_int64
        _lib.orc_round_up.restype = C.c_uint64
        _lib.orc_brute_force_count_mt_f32.restype = C.c_uint64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _suf(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return "f32"
    if dt == np.float64:
        return "f64"
    raise ValueError(dt)


def pad4(coords):
    """(n,3) -> (n,4) rows; float3 is 4 scalars wide on the device (misc.py:62-71)."""
    coords = np.asarray(coords)
    out = np.zeros((len(coords), 4), dtype=coords.dtype)
    out[:, :3] = coords[:, :3]
    return out


# ---------------------------------------------------------------- stages
def bounds(values):
    """reduce.cl:5-58 + bounds.py:5. values (n, width) -> (2, width) [min, max]."""
    values = np.ascontiguousarray(values)
    if values.ndim == 1:
        values = values.reshape(-1, 1)
    out = np.empty((2, values.shape[1]), dtype=values.dtype)
    getattr(lib(), "orc_bounds_" + _suf(values.dtype))(
        _p(values), C.c_uint64(len(values)), C.c_uint32(values.shape[1]), _p(out))
    return out


def _int_info(dt):
    info = np.iinfo(dt)
    return int(info.min), int(info.max)


def _cl_binary(name, dt):
    """OpenCL's binary built-in `name` (or reduce.cl:3's ADD) on arrays of scalar dtype `dt`, as NumPy arithmetic in that
    dtype -- the functions whose results NumPy can give exactly."""
    dt = np.dtype(dt)
    if dt.kind == "f":
        table = {
            "ADD": lambda a, b: (a + b).astype(dt), "MUL": lambda a, b: (a * b).astype(dt),
            "min": lambda a, b: np.where(b < a, b, a), "max": lambda a, b: np.where(a < b, b, a),
            "fmin": np.fmin, "fmax": np.fmax, "copysign": np.copysign,
            "maxmag": lambda a, b: np.where(np.abs(a) > np.abs(b), a, np.where(np.abs(b) > np.abs(a), b, np.fmax(a, b))),
            "minmag": lambda a, b: np.where(np.abs(a) < np.abs(b), a, np.where(np.abs(b) < np.abs(a), b, np.fmin(a, b))),
            "fdim": lambda a, b: np.where(np.isnan(a) | np.isnan(b), np.nan, np.where(a > b, a - b, 0)).astype(dt),
        }
        return table[name]
    lo, hi = _int_info(dt)
    bits = 8 * dt.itemsize

    def big(f):          # exact integer arithmetic on Python ints, wrapped / clipped back into dt
        def g(a, b):
            out = [f(int(x), int(y)) for x, y in zip(np.ravel(a), np.ravel(b))]
            return np.array([((v - lo) % (1 << bits)) + lo for v in out], dtype=object).astype(dt).reshape(np.shape(a))
        return g
    table = {
        "ADD": big(lambda x, y: x + y), "MUL": big(lambda x, y: x * y),
        "min": lambda a, b: np.where(b < a, b, a), "max": lambda a, b: np.where(a < b, b, a),
        "add_sat": big(lambda x, y: min(max(x + y, lo), hi)), "sub_sat": big(lambda x, y: min(max(x - y, lo), hi)),
        "hadd": big(lambda x, y: (x + y) >> 1), "rhadd": big(lambda x, y: (x + y + 1) >> 1),
        "abs_diff": big(lambda x, y: abs(x - y)), "mul_hi": big(lambda x, y: (x * y) >> bits),
    }
    return table[name]


def reduce_list(values, accumulator, ngroups, group_size):
    """reduce.cl:5-58 for ANY accumulator list [(initial value, function name), ...] on ngroups x group_size work-items
    (reduce.py:62-76): every work-item folds values[gid], values[gid + G], ... into its accumulators in that order
    (reduce.cl:13-17), a group folds its work-items' rows by the halving tree of reduce.cl:22-33 (row l + o into row l),
    bounds2 folds the group rows the same way (reduce.cl:40-52).  Sizes that are not powers of two: every row is folded
    (the tree starts at the next power of two; the reference's `o = size / 2` loop would drop rows).
    values (n, width) -> (len(accumulator), width).  NumPy arithmetic in the value dtype: exact for the functions of
    _cl_binary."""
    values = np.ascontiguousarray(values)
    if values.ndim == 1:
        values = values.reshape(-1, 1)
    dt, width, n = values.dtype, values.shape[1], len(values)
    G = ngroups * group_size
    out = np.empty((len(accumulator), width), dtype=dt)
    consts = {"INFINITY": np.inf, "-INFINITY": -np.inf, "FLT_MAX": float(np.finfo(np.float32).max), "DBL_MAX": float(np.finfo(np.float64).max),
              "SHRT_MAX": 32767, "SHRT_MIN": -32768, "INT_MAX": 2 ** 31 - 1, "INT_MIN": -2 ** 31, "UINT_MAX": 2 ** 32 - 1,
              "LONG_MAX": 2 ** 63 - 1, "LONG_MIN": -2 ** 63, "ULONG_MAX": 2 ** 64 - 1, "UCHAR_MAX": 255, "USHRT_MAX": 65535}

    def tree(rows):                                     # rows: (groups, size, width) -> (groups, width)
        size = rows.shape[1]
        top = 1
        while top < size:
            top <<= 1
        o = top // 2
        while o > 0:
            m = min(o, size - o)                        # l < o and l + o < size
            if m > 0:
                rows[:, :m] = fn(rows[:, :m], rows[:, o:o + m])
            o //= 2
        return rows[:, 0]

    with np.errstate(all="ignore"):
        for k, (init, name) in enumerate(accumulator):
            fn = _cl_binary(name, dt)
            start = consts[str(init)] if str(init) in consts else (float(init) if dt.kind == "f" else int(str(init), 0))
            acc = np.full((G, width), start, dtype=dt)
            for i in range(0, n, G):
                rows = values[i:i + G]
                acc[:len(rows)] = fn(acc[:len(rows)], rows)
            groups = tree(acc.reshape(ngroups, group_size, width).copy())
            out[k] = tree(groups.reshape(1, ngroups, width).copy())[0]
    return out


def morton(coords4, rng):
    """collision.cl:22-40. coords4 (n,4), rng (2,4) -> uint32 codes."""
    coords4 = np.ascontiguousarray(coords4)
    rng = np.ascontiguousarray(rng, dtype=coords4.dtype)
    codes = np.empty(len(coords4), dtype=np.uint32)
    getattr(lib(), "orc_morton_" + _suf(coords4.dtype))(
        _p(coords4), _p(rng), C.c_uint32(len(coords4)), _p(codes))
    return codes


def local_scan(data, block):
    """scan.cl:5-30 -> (scanned copy, block sums)."""
    data = np.array(data, dtype=np.uint32)
    sums = np.empty(len(data) // block, dtype=np.uint32)
    lib().orc_local_scan(_p(data), C.c_uint64(len(data)), C.c_uint32(block), _p(sums))
    return data, sums


def block_scan(data, block, sums):
    """scan.cl:32-36."""
    data = np.array(data, dtype=np.uint32)
    sums = np.ascontiguousarray(sums, dtype=np.uint32)
    lib().orc_block_scan(_p(data), C.c_uint64(len(data)), C.c_uint32(block), _p(sums))
    return data


def exclusive_scan(data):
    """scan.py:75-112 composed result."""
    data = np.array(data, dtype=np.uint32)
    lib().orc_exclusive_scan(_p(data), C.c_uint64(len(data)))
    return data


def _val_bytes(values):
    return 0 if values is None else values.dtype.itemsize * int(np.prod(values.shape[1:], dtype=np.int64))


def block_sort(keys, values, block, bits, rpass):
    """radix.cl:48-102 -> (keys, values, histogram[(2**bits, nblocks)])."""
    keys = np.array(keys)
    values = None if values is None else np.array(values)
    nblocks = len(keys) // block
    hist = np.empty((1 << bits, nblocks), dtype=np.uint32)
    rc = lib().orc_block_sort(_p(keys), _p(values), C.c_uint64(len(keys)), C.c_uint32(keys.dtype.itemsize),
                              C.c_uint32(_val_bytes(values)), C.c_uint32(block), C.c_uint32(bits),
                              C.c_uint32(rpass), _p(hist))
    assert rc == 0
    return keys, values, hist


def scatter(keys, values, block, bits, rpass, offset, hist):
    """radix.cl:104-139 -> (out_keys, out_values)."""
    keys = np.ascontiguousarray(keys)
    out_keys = np.empty_like(keys)
    out_values = None if values is None else np.empty_like(values)
    offset = np.ascontiguousarray(offset, dtype=np.uint32)
    hist = np.ascontiguousarray(hist, dtype=np.uint32)
    lib().orc_scatter(_p(keys), _p(out_keys), _p(values), _p(out_values), C.c_uint64(len(keys)),
                      C.c_uint32(keys.dtype.itemsize), C.c_uint32(_val_bytes(values)), C.c_uint32(block),
                      C.c_uint32(bits), C.c_uint32(rpass), _p(offset), _p(hist))
    return out_keys, out_values


def radix_sort(keys, values=None, block=None, bits=4):
    """radix.py:118-170 -> (sorted keys, sorted values)."""
    keys = np.array(keys)
    values = None if values is None else np.array(values)
    if block is None:
        block = len(keys)
    out_keys = np.empty_like(keys)
    out_values = None if values is None else np.empty_like(values)
    rc = lib().orc_radix_sort(_p(keys), _p(out_keys), _p(values), _p(out_values), C.c_uint64(len(keys)),
                              C.c_uint32(keys.dtype.itemsize), C.c_uint32(_val_bytes(values)),
                              C.c_uint32(block), C.c_uint32(bits))
    assert rc == 0
    return out_keys, out_values


def build_bvh(codes, ids):
    """collision.cl:55-121 -> Node[2n-1] (root parent left as NO_NODE, leaf data[1] as NO_NODE)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint32)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    n = len(codes)
    nodes = np.full(2 * n - 1, NO_NODE, dtype=np.uint32).repeat(4).view(Node)
    nodes = np.ascontiguousarray(nodes)
    lib().orc_fill_leaves(_p(nodes), _p(ids), C.c_uint32(n))
    lib().orc_generate_bvh(_p(codes), _p(nodes), C.c_uint32(n))
    return nodes


def node_bounds(coords4, radii, nodes):
    """collision.cl:128-162 -> bounds (2n-1, 2, 4); lane w is NaN-filled (undefined on device)."""
    coords4 = np.ascontiguousarray(coords4)
    radii = np.ascontiguousarray(radii, dtype=coords4.dtype)
    nodes = np.ascontiguousarray(nodes)
    n = (len(nodes) + 1) // 2
    b = np.full((len(nodes), 2, 4), np.nan, dtype=coords4.dtype)
    flags = np.zeros(len(nodes), dtype=np.uint32)
    s = _suf(coords4.dtype)
    getattr(lib(), "orc_leaf_bounds_" + s)(_p(b), _p(coords4), _p(radii), _p(nodes), C.c_uint32(n))
    getattr(lib(), "orc_internal_bounds_" + s)(_p(b), _p(flags), _p(nodes), C.c_uint32(n))
    return b


def traverse(nodes, bnds, capacity=None):
    """collision.cl:174-226 -> (total count, pairs[min(count, capacity), 2])."""
    nodes = np.ascontiguousarray(nodes)
    bnds = np.ascontiguousarray(bnds)
    n = (len(nodes) + 1) // 2
    fn = getattr(lib(), "orc_traverse_" + _suf(bnds.dtype))
    if capacity is None:
        capacity = int(fn(None, C.c_uint64(0), _p(nodes), _p(bnds), C.c_uint32(n)))
    pairs = np.empty((max(capacity, 1), 2), dtype=np.uint32)
    count = int(fn(_p(pairs), C.c_uint64(capacity), _p(nodes), _p(bnds), C.c_uint32(n)))
    return count, pairs[:min(count, capacity)]


def brute_force(coords4, radii, capacity=None):
    """C double loop of tests/test_collision_py.py:30-37 -> (count, pairs (i<j))."""
    coords4 = np.ascontiguousarray(coords4)
    radii = np.ascontiguousarray(radii, dtype=coords4.dtype)
    fn = getattr(lib(), "orc_brute_force_" + _suf(coords4.dtype))
    n = len(coords4)
    if capacity is None:
        capacity = int(fn(None, C.c_uint64(0), _p(coords4), _p(radii), C.c_uint32(n)))
    pairs = np.empty((max(capacity, 1), 2), dtype=np.uint32)
    count = int(fn(_p(pairs), C.c_uint64(capacity), _p(coords4), _p(radii), C.c_uint32(n)))
    return count, pairs[:min(count, capacity)]


def brute_force_count_all_cores(coords4, radii):
    """Pair count of the O(n^2) brute force on every host core (OpenMP) -> (count, threads)."""
    coords4 = np.ascontiguousarray(coords4, dtype=np.float32)
    radii = np.ascontiguousarray(radii, dtype=np.float32)
    nt = C.c_int(0)
    count = lib().orc_brute_force_count_mt_f32(_p(coords4), _p(radii), C.c_uint32(len(coords4)), C.byref(nt))
    return int(count), nt.value


def collide(coords4, radii, padded=None, capacity=0, want=True):
    """collision.py:130-198 end to end.

    Returns dict(count, pairs, codes, ids, nodes, bounds); the arrays are only
    produced when ``want`` is true (the timed CPU baseline passes False).
    """
    coords4 = np.ascontiguousarray(coords4)
    radii = np.ascontiguousarray(radii, dtype=coords4.dtype)
    n = len(coords4)
    if padded is None:
        padded = n
    codes = np.empty(padded, np.uint32) if want else None
    ids = np.empty(padded, np.uint32) if want else None
    nodes = np.full(2 * n - 1, NO_NODE, np.uint32).repeat(4).view(Node).copy() if want else None
    bnds = np.full((2 * n - 1, 2, 4), np.nan, coords4.dtype) if want else None
    pairs = np.empty((max(capacity, 1), 2), np.uint32)
    count = getattr(lib(), "orc_collide_" + _suf(coords4.dtype))(
        _p(coords4), _p(radii), C.c_uint32(n), C.c_uint32(padded), _p(codes), _p(ids), _p(nodes), _p(bnds),
        _p(pairs), C.c_uint64(capacity))
    assert count >= 0
    return dict(count=int(count), pairs=pairs[:min(int(count), capacity)], codes=codes, ids=ids,
                nodes=nodes, bounds=bnds)


def gather(values, idx):
    """index.cl:1-6."""
    return np.ascontiguousarray(values)[np.asarray(idx)]


def find_offsets(values, n_offsets):
    """offset.py:37-49 + offset.cl:3-12."""
    values = np.ascontiguousarray(values, dtype=np.uint32)
    out = np.empty(n_offsets, dtype=np.uint32)
    lib().orc_find_offsets(_p(values), C.c_uint64(len(values)), _p(out), C.c_uint64(n_offsets))
    return out


# ------------------------------------------------- NumPy brute force
def find_collisions(coords, radii):
    """NumPy restatement of tests/test_collision_py.py:30-37.

    AABB (not sphere) overlap with strict inequalities; every unordered pair
    once, as the set {(lower index, higher index)}.
    """
    coords = np.asarray(coords)[:, :3]
    radii = np.asarray(radii)
    lo = coords - radii[:, None]
    hi = coords + radii[:, None]
    hit = ((hi[:, None, :] > lo[None, :, :]) & (lo[:, None, :] < hi[None, :, :])).all(axis=-1)
    hit = np.tril(hit, -1)
    rows, cols = np.nonzero(hit)
    return set(zip(cols.tolist(), rows.tolist()))

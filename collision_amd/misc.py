"""Host-side helpers of the path: size arithmetic and dtype bookkeeping.

Mirrors the public names of the reference's ``collision/misc.py`` (roundUp, nextPowerOf2,
product, dtype_decl, dtype_sizeof: misc.py:31-71, pinned by tests/test_misc.py:4-46).  The
OpenCL program-compilation half of that module (misc.py:6-28) has no counterpart here: the
kernels are precompiled HIP behind the C ABI, so a "program" is only a typed handle
(:class:`ProgramHandle`).
"""
import math

import numpy as np

from ._lib import cdll

# element type codes of include/collision_hip.h
COL_F32, COL_F64, COL_U32, COL_I32, COL_U64, COL_I64 = range(6)
COL_OP_MINMAX, COL_OP_SUM = 0, 1

np_integer_dtypes = ["int8", "int16", "int32", "int64"]
np_unsigned_dtypes = ["u" + name for name in np_integer_dtypes]
np_float_dtypes = ["float16", "float32", "float64"]
np_dtypes = np_integer_dtypes + np_unsigned_dtypes + np_float_dtypes

_C_NAMES = dict(zip(np_dtypes, ["char", "short", "int", "long", "uchar", "ushort", "uint", "ulong",
                                "half", "float", "double"]))
_VECTOR_WIDTHS = (2, 3, 4, 8, 16)
_TYPE_CODES = {"float32": COL_F32, "float64": COL_F64, "uint32": COL_U32, "int32": COL_I32,
               "uint64": COL_U64, "int64": COL_I64}


def roundUp(x, base=1):
    """Smallest multiple of `base` that is >= x (misc.py:31-32)."""
    return -(-x // base) * base


def nextPowerOf2(x):
    """Smallest power of two >= x (misc.py:34-35)."""
    return 1 << (x - 1).bit_length()


def product(xs):
    """Product of an iterable, 1 when empty (misc.py:37-38)."""
    return math.prod(xs)


def dtype_decl(dt):
    """Device-side type name of a NumPy dtype, e.g. ``('float16', 4) -> 'half4'`` (misc.py:51-60).
    Kept for API parity; nothing is compiled from it here."""
    dt = np.dtype(dt)
    if dt.shape == ():
        return _C_NAMES[dt.name]
    if len(dt.shape) != 1:
        raise ValueError("Too many vector dimensions: {}".format(dt.shape))
    if dt.shape[0] not in _VECTOR_WIDTHS:
        raise ValueError("Invalid vector size: {}".format(dt.shape[0]))
    return _C_NAMES[dt.base.name] + str(dt.shape[0])


def dtype_sizeof(dt):
    """Bytes one element occupies on the device: 3-vectors are 4 wide (misc.py:62-71)."""
    dt = np.dtype(dt)
    if dt.base.name not in np_dtypes:
        if dt.subdtype is None:
            raise TypeError("Unsupported dtype: {}".format(dt))
        sub, shape = dt.subdtype
        return product(shape) * dtype_sizeof(sub)
    *outer, width = dt.shape or (1,)
    if width != 1 and width not in _VECTOR_WIDTHS:
        raise ValueError("Invalid vector size: {}".format(width))
    return dt.base.itemsize * product(outer) * (4 if width == 3 else width)


def device_width(dt):
    """Scalars per device row of a (possibly vector) dtype: () -> 1, (3,) -> 4, (n,) -> n."""
    dt = np.dtype(dt)
    if dt.shape == ():
        return 1
    if len(dt.shape) != 1 or dt.shape[0] not in _VECTOR_WIDTHS:
        raise ValueError("Invalid vector shape: {}".format(dt.shape))
    return 4 if dt.shape[0] == 3 else dt.shape[0]


def type_code(dt):
    """C-ABI element type code of the scalar base of `dt`; ValueError if the kernels lack it."""
    name = np.dtype(dt).base.name
    if name not in _TYPE_CODES:
        raise ValueError("Unsupported element dtype on this device path: {}".format(name))
    return _TYPE_CODES[name]


class ProgramHandle:
    """What remains of the reference's ``Program`` (misc.py:6-22) once the kernels are
    precompiled: the context it belongs to (plus, in subclasses, the dtypes it was
    specialised for).  Creating one makes sure the shared library is loadable."""

    def __init__(self, ctx):
        cdll()
        self._context = ctx

    @property
    def context(self):
        return self._context


# names the reference exports
Program = ProgramHandle
SimpleProgram = ProgramHandle

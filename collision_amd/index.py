"""Elementwise gather / scatter by an index buffer (``collision/index.py:8-55``,
``collision/index.cl:1-13``); not used by Collider."""
import numpy as np

from . import hip
from ._lib import call
from .misc import ProgramHandle, dtype_sizeof, np_unsigned_dtypes

_UNSIGNED = {np.dtype(name) for name in np_unsigned_dtypes}


class IndexProgram(ProgramHandle):
    def __init__(self, ctx, value_dtype=np.dtype("uint32"), index_dtype=np.dtype("uint32")):
        self.value_dtype = np.dtype(value_dtype)
        self.index_dtype = np.dtype(index_dtype)
        if self.index_dtype not in _UNSIGNED:
            raise ValueError("Invalid index dtype: {}".format(self.index_dtype))
        if dtype_sizeof(self.value_dtype) not in (1, 2, 4, 8, 16, 32):
            raise ValueError("Unsupported value dtype on this device path: {}".format(self.value_dtype))
        super().__init__(ctx)


class Indexer:
    def __init__(self, ctx, value_dtype=np.dtype("uint32"), index_dtype=np.dtype("uint32"), program=None):
        if program is None:
            program = IndexProgram(ctx, value_dtype, index_dtype)
        else:
            if program.context != ctx:
                raise ValueError("Sorter and program contexts must match")
            if program.index_dtype != np.dtype(index_dtype):
                raise ValueError("Sorter and program index dtypes must match")
            if program.value_dtype != np.dtype(value_dtype):
                raise ValueError("Sorter and program value dtypes must match")
        self.program = program

    def _run(self, fn, cq, size, in_values_buf, indices_buf, out_values_buf, wait_for):
        cq.wait_for(wait_for)
        fn(cq.stream, in_values_buf.ptr, indices_buf.ptr, out_values_buf.ptr, size,
           dtype_sizeof(self.program.value_dtype), self.program.index_dtype.itemsize)
        return hip.Event(cq)

    def gather(self, cq, size, in_values_buf, indices_buf, out_values_buf, wait_for=None):
        """out[i] = in[indices[i]] (index.cl:1-6)."""
        return self._run(call.col_gather, cq, size, in_values_buf, indices_buf, out_values_buf, wait_for)

    def scatter(self, cq, size, in_values_buf, indices_buf, out_values_buf, wait_for=None):
        """out[indices[i]] = in[i] (index.cl:8-13)."""
        return self._run(call.col_scatter, cq, size, in_values_buf, indices_buf, out_values_buf, wait_for)

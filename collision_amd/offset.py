"""First index of each value in a sorted array -- a cell-start table
(``collision/offset.py:8-49``, ``collision/offset.cl:3-12``); not used by Collider."""
import numpy as np

from . import hip
from ._lib import call
from .misc import ProgramHandle, np_unsigned_dtypes

_UNSIGNED = {np.dtype(name) for name in np_unsigned_dtypes}


class OffsetProgram(ProgramHandle):
    def __init__(self, ctx, value_dtype=np.dtype("uint32"), offset_dtype=np.dtype("uint32")):
        self.value_dtype = np.dtype(value_dtype)
        self.offset_dtype = np.dtype(offset_dtype)
        for what, dt in (("value", self.value_dtype), ("offset", self.offset_dtype)):
            if dt not in _UNSIGNED:
                raise ValueError("Invalid {} dtype: {}".format(what, dt))
        super().__init__(ctx)


class OffsetFinder:
    def __init__(self, ctx, value_dtype=np.dtype("uint32"), offset_dtype=np.dtype("uint32"), program=None):
        if program is None:
            program = OffsetProgram(ctx, value_dtype, offset_dtype)
        else:
            if program.context != ctx:
                raise ValueError("Sorter and program contexts must match")
            if program.value_dtype != np.dtype(value_dtype):
                raise ValueError("Sorter and program value dtypes must match")
            if program.offset_dtype != np.dtype(offset_dtype):
                raise ValueError("Sorter and program offset dtypes must match")
        self.program = program

    def find_offsets(self, cq, values_buf, n_values, offsets_buf, n_offsets, wait_for=None):
        """offsets[v] = first i with values[i] >= v; entries past the largest value hold n_values."""
        cq.wait_for(wait_for)
        call.col_find_offsets(cq.stream, values_buf.ptr, n_values, offsets_buf.ptr, n_offsets,
                              self.program.value_dtype.itemsize, self.program.offset_dtype.itemsize)
        return hip.Event(cq)

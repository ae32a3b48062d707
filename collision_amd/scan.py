"""Exclusive prefix sum of uint32 on the device.

Mirrors ``collision/scan.py`` (PrefixScanProgram :10-13, PrefixScanner :15-112): same
constructor, ``check_size`` / ``resize`` / ``block_lengths`` / ``prefix_sum`` and the same
``ValueError`` contract.  The device work is one C-ABI call, ``col_scan_u32`` (reduce-then-scan,
see csrc/scan.hip); ``group_size`` only governs the size rules and ``block_lengths``, which are
kept because callers and tests read them (tests/test_scan_py.py:15-43).
"""
import numpy as np

from . import hip
from ._lib import call
from .misc import ProgramHandle, nextPowerOf2, roundUp


def ceildiv(a, b):
    return -(-a // b)


class PrefixScanProgram(ProgramHandle):
    """Typed handle (scan.py:10-13); uint32 only, as in the reference."""


class PrefixScanner:
    block_sums_dtype = np.dtype("uint32")

    def __init__(self, ctx, size, group_size, program=None):
        self.check_size(size, group_size)
        if program is None:
            program = PrefixScanProgram(ctx)
        elif program.context != ctx:
            raise ValueError("Scanner and program context must match")
        self.program = program
        self.size = size
        self.group_size = group_size
        self._scratch = None          # device scratch, allocated at first use
        self._scratch_for = None

    @staticmethod
    def check_size(size, group_size):
        # scan.py:34-39
        if group_size != nextPowerOf2(group_size):
            raise ValueError("Group size ({}) must be a power of two".format(group_size))
        if size % (2 * group_size):
            raise ValueError("Size must be multiple of 2 * group_size ({})".format(group_size))

    def _ensure_scratch(self):
        if self._scratch_for != self.size:
            nbytes = call.col_scan_scratch_bytes(self.size)
            self._scratch = hip.Buffer(self.program.context, nbytes)
            self._scratch_for = self.size

    def resize(self, size=None, group_size=None):
        size = self.size if size is None else size
        group_size = self.group_size if group_size is None else group_size
        self.check_size(size, group_size)       # raises before any state changes
        self.size, self.group_size = size, group_size

    @property
    def block_lengths(self):
        """Level sizes of the reference's multi-level scan (scan.py:62-73), pinned by
        tests/test_scan_py.py:32-43."""
        step = 2 * self.group_size
        levels = []
        length = roundUp(ceildiv(self.size, step), step)
        while length > step:
            length = roundUp(length, step)
            levels.append(length)
            length = ceildiv(length, step)
        levels.append(nextPowerOf2(length))
        return tuple(levels)

    def prefix_sum(self, cq, values_buf, wait_for=None):
        """In-place exclusive scan of the first ``size`` uint32 of values_buf (scan.py:75-112)."""
        self._ensure_scratch()
        cq.wait_for(wait_for)
        call.col_scan_u32(cq.stream, values_buf.ptr, self.size, self._scratch.ptr)
        return hip.Event(cq)

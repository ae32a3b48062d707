"""Stable LSD radix sort of keys (+ values) on the device.

Mirrors ``collision/radix.py`` (RadixProgram :9-25, RadixSorter :27-170): constructor,
``check_size``, ``resize`` with roll-back, ``num_passes``, ``histogram_len`` and
``sort(cq, keys_buf, out_keys_buf, in_values_buf, out_values_buf, wait_for)``, with the same
``ValueError`` contract.  ``PrefixScanProgram`` / ``PrefixScanner`` are re-exported because the
reference's callers import them from here (tests/test_collision_py.py:19).

The device work is ``col_radix_sort``: 8-bit digits whatever ``radix_bits`` says -- the result
of a stable sort over all key bits does not depend on the digit width -- so ``radix_bits`` and
``group_size`` only govern the size rules below.  Unlike the reference (radix.py:158-169) the
input buffers are left untouched unless ``keep_sorted_inputs`` is set.
"""
import numpy as np

from . import hip
from ._lib import call
from .misc import ProgramHandle, device_width, nextPowerOf2, np_unsigned_dtypes, roundUp
from .scan import PrefixScanProgram, PrefixScanner  # noqa: F401  (re-exported)

_UNSIGNED = {np.dtype(name) for name in np_unsigned_dtypes}
_KEY_BYTES = (1, 2, 4, 8)             # uint8 / uint16 keys are widened to u32 on the device, one pass per key byte
_VALUE_BYTES = (1, 2, 4, 8, 16, 32, 64, 128)     # 1 / 2 / 64 / 128: sorted as (key, index), gathered once


def _value_bytes(value_dtype):
    value_dtype = np.dtype(value_dtype)
    return value_dtype.base.itemsize * device_width(value_dtype)


class RadixProgram(ProgramHandle):
    """Typed handle (radix.py:9-25): the key/value dtypes the sorter is specialised for."""

    def __init__(self, ctx, key_dtype=np.dtype("uint32"), value_dtype=np.dtype("uint32")):
        self.key_dtype = np.dtype(key_dtype)
        self.value_dtype = np.dtype(value_dtype)
        if self.key_dtype not in _UNSIGNED:
            raise ValueError("Invalid key dtype: {}".format(self.key_dtype))
        if self.key_dtype.itemsize not in _KEY_BYTES:
            raise ValueError("Unsupported key dtype on this device path: {}".format(self.key_dtype))
        if _value_bytes(self.value_dtype) not in _VALUE_BYTES:
            raise ValueError("Unsupported value dtype on this device path: {}".format(self.value_dtype))
        super().__init__(ctx)


class RadixSorter:
    histogram_dtype = np.dtype("uint32")

    def __init__(self, ctx, size, group_size, radix_bits=4, key_dtype=np.dtype("uint32"),
                 value_dtype=np.dtype("uint32"), program=None, scan_program=None):
        self.check_size(size, group_size, radix_bits, key_dtype)
        if program is None:
            program = RadixProgram(ctx, key_dtype, value_dtype)
        else:
            if program.context != ctx:
                raise ValueError("Sorter and program contexts must match")
            if program.key_dtype != np.dtype(key_dtype):
                raise ValueError("Sorter and program key dtypes must match")
            if program.value_dtype != np.dtype(value_dtype):
                raise ValueError("Sorter and program value dtypes must match")
        self.program = program
        self.size, self.group_size, self.radix_bits = size, group_size, radix_bits
        self.keep_sorted_inputs = False
        if scan_program is None:
            scan_program = PrefixScanProgram(ctx)
        self.scanner = PrefixScanner(ctx, self.histogram_len, group_size, scan_program)
        self._scratch = None          # device scratch, allocated at first use
        self._scratch_for = None

    @staticmethod
    def check_size(size, group_size, radix_bits, key_dtype):
        # radix.py:61-74
        key_bits = np.dtype(key_dtype).itemsize * 8
        if group_size != nextPowerOf2(group_size):
            raise ValueError("Group size ({}) must be a power of two".format(group_size))
        if size % (2 * group_size):
            raise ValueError("Size ({}) must be multiple of 2 * group_size ({})".format(size, group_size))
        if key_bits % radix_bits:
            raise ValueError("Radix bits ({}) must evenly divide item-size ({})".format(radix_bits, key_bits))
        if 2 ** radix_bits > 2 * group_size:
            raise ValueError("2 ^ radix_bits ({}) must be less than 2 * group_size ({})"
                             .format(radix_bits, group_size))

    @property
    def key_bytes(self):
        return self.program.key_dtype.itemsize

    @property
    def value_bytes(self):
        return _value_bytes(self.program.value_dtype)

    def _ensure_scratch(self):
        if self._scratch_for != self.size:
            nbytes = call.col_radix_scratch_bytes(self.size, self.key_bytes, self.value_bytes)
            self._scratch = hip.Buffer(self.program.context, nbytes)
            self._scratch_for = self.size

    def resize(self, size=None, group_size=None, radix_bits=None):
        new = (self.size if size is None else size,
               self.group_size if group_size is None else group_size,
               self.radix_bits if radix_bits is None else radix_bits)
        self.check_size(*new, self.program.key_dtype)
        old = (self.size, self.group_size, self.radix_bits)
        self.size, self.group_size, self.radix_bits = new
        try:
            self.scanner.resize(self.histogram_len, self.group_size)
        except Exception:
            self.size, self.group_size, self.radix_bits = old     # radix.py:93-97
            raise

    @property
    def num_passes(self):
        """Passes the reference runs for this radix_bits (radix.py:109-111)."""
        return self.key_bytes * 8 // self.radix_bits

    @property
    def histogram_len(self):
        """Length of the reference's digit-major block histogram (radix.py:113-116)."""
        blocks = self.size // (2 * self.group_size)
        return roundUp(2 ** self.radix_bits * blocks, 2 * self.group_size)

    def sort(self, cq, keys_buf, out_keys_buf, in_values_buf=None, out_values_buf=None, wait_for=None):
        """Sorted keys land in out_keys_buf, values (if both value buffers are given) follow their
        keys stably into out_values_buf (radix.py:118-170)."""
        self._ensure_scratch()
        cq.wait_for(wait_for)
        with_values = in_values_buf is not None and out_values_buf is not None
        call.col_radix_sort(
            cq.stream, keys_buf.ptr, out_keys_buf.ptr,
            in_values_buf.ptr if with_values else None, out_values_buf.ptr if with_values else None,
            self.size, self.key_bytes, self.value_bytes if with_values else 0,
            self._scratch.ptr, 1 if self.keep_sorted_inputs else 0)
        return hip.Event(cq)

"""Two-stage device reductions over rows of scalars.

Mirrors ``collision/reduce.py`` (ReductionProgram :9-22, Reducer :24-76).  The reference renders
its kernel from a Jinja2 template with a list of ``(init, fn)`` accumulators; here the two
accumulator lists the package uses are compiled in (``COL_OP_MINMAX`` for bounds.py:5,
``COL_OP_SUM`` for summer.py:5) and any other list of up to four accumulators goes through the
table-driven ``col_reduce_list`` -- ``fn`` one of ``min`` / ``max`` / ``fmin`` / ``fmax`` / ``ADD`` /
``MUL`` (the binary functions the template can name), ``init`` a number or ``[-]INFINITY``.  A
subclass names its list through ``accumulator``.
``ngroups`` / ``group_size`` are accepted for API parity; the launch geometry is the kernel's own
(csrc/reduce.hip) and, unlike reduce.cl:40-52, every partial is folded whatever ``ngroups`` is.
"""
import ctypes as C

import numpy as np

from . import hip
from ._lib import call
from .misc import COL_OP_MINMAX, COL_OP_SUM, ProgramHandle, device_width, type_code

_OPS = {
    (("INFINITY", "min"), ("-INFINITY", "max")): COL_OP_MINMAX,
    (("0", "ADD"),): COL_OP_SUM,
}


_ACC_FN = {"min": 0, "fmin": 0, "max": 1, "fmax": 1, "ADD": 2, "MUL": 3}
MAX_ACCUMULATORS = 4


def _parse_init(text, dtype):
    """(value as a double, value as an exact 64-bit integer or None).  The reference renders the text verbatim into
    the kernel; here it must be a number or [-]INFINITY.  Integer dtypes take integer initial values exactly
    (a double is inexact beyond 2^53: UINT64_MAX as the start of a min list), and an unsigned dtype refuses a
    negative one."""
    text = str(text).strip().replace("(", "").replace(")", "")
    if text in ("INFINITY", "+INFINITY"):
        return float("inf"), 0
    if text == "-INFINITY":
        return float("-inf"), 0
    base = np.dtype(dtype).base
    if base.kind not in "ui":
        return float(text), None  # ValueError for anything that is not a number
    try:
        exact = int(text, 0)      # decimal, 0x..., 0b...: as C spells integer literals
    except ValueError:
        value = float(text)
        if value != int(value):
            raise ValueError("Initial value {} is not an integer ({} accumulator)".format(text, base))
        exact = int(value)
    value = float(exact)
    info = np.iinfo(base)
    if not info.min <= exact <= info.max:
        raise ValueError("Initial value {} outside the range of {}".format(text, base))
    return value, exact - (1 << 64) if exact >= (1 << 63) else exact


class ReductionProgram(ProgramHandle):
    accumulator = None      # set by subclasses, as in the reference (bounds.py:5, summer.py:5)

    def __init__(self, ctx, value_dtype):
        self.value_dtype = np.dtype(value_dtype)
        key = tuple(tuple(a) for a in (self.accumulator or ()))
        if not key:
            raise ValueError("Unsupported accumulator list: {}".format(self.accumulator))
        self.op = _OPS.get(key)
        if self.op is None:       # not one of the compiled-in lists: the table-driven reducer
            if len(key) > MAX_ACCUMULATORS or any(fn not in _ACC_FN for _, fn in key):
                raise ValueError("Unsupported accumulator list: {}".format(self.accumulator))
            self.acc_ops = (C.c_int * len(key))(*[_ACC_FN[fn] for _, fn in key])
            parsed = [_parse_init(init, self.value_dtype) for init, _ in key]
            self.acc_inits = (C.c_double * len(key))(*[v for v, _ in parsed])
            self.acc_int_inits = None              # exact integer initial values for the integer dtypes
            if all(e is not None for _, e in parsed):
                self.acc_int_inits = (C.c_int64 * len(key))(*[e for _, e in parsed])
        self.acc_dtype = np.dtype((self.value_dtype, len(key)))
        self.type_code = type_code(self.value_dtype)
        self.width = device_width(self.value_dtype)
        super().__init__(ctx)


class Reducer:
    program_type = ReductionProgram

    def __init__(self, ctx, ngroups, group_size, value_dtype, program=None):
        if program is None:
            program = self.program_type(ctx, value_dtype)
        else:
            if program.context != ctx:
                raise ValueError("Collider and program context must match")
            if program.value_dtype != np.dtype(value_dtype):
                raise ValueError("Reducer and program value dtypes must match")
        self.program = program
        self.ngroups = ngroups
        self.group_size = group_size
        self._scratch = None          # device scratch, allocated at first use

    def resize(self, ngroups=None, group_size=None):
        if ngroups is not None:
            self.ngroups = ngroups
        if group_size is not None:
            self.group_size = group_size

    def reduce(self, cq, size, values_buf, output_buf, wait_for=None):
        """Fold the first ``size`` rows of values_buf; output_buf receives one row per accumulator
        (min row then max row for Bounds) at its start (reduce.py:62-76)."""
        p = self.program
        if self._scratch is None:
            self._scratch = hip.Buffer(p.context, call.col_reduce_scratch_bytes(p.type_code, p.width))
        cq.wait_for(wait_for)
        if p.op is not None:
            call.col_reduce(cq.stream, values_buf.ptr, size, p.type_code, p.width, p.op,
                            self._scratch.ptr, output_buf.ptr)
        else:
            call.col_reduce_list(cq.stream, values_buf.ptr, size, p.type_code, p.width, len(p.acc_ops), p.acc_ops,
                                 p.acc_inits, p.acc_int_inits, self._scratch.ptr, output_buf.ptr)
        return hip.Event(cq)


def specialise(program_name, reducer_name, accumulator, default_dtype=None, dtype_keyword="value_dtype"):
    """Build the (Program, Reducer) pair for one accumulator list -- what the reference spells as
    two small subclasses per module (bounds.py:4-15, summer.py:4-8).  `dtype_keyword` is the name
    the constructors give their dtype argument; `default_dtype` makes it optional."""
    accumulator = [tuple(a) for a in accumulator]

    def program_init(self, ctx, *args, **kwargs):
        dt = _pick_dtype(args, kwargs, 0, dtype_keyword, default_dtype)
        ReductionProgram.__init__(self, ctx, dt)

    program_cls = type(program_name, (ReductionProgram,), {"accumulator": accumulator, "__init__": program_init,
                                                           "__module__": __name__})

    def reducer_init(self, ctx, ngroups, group_size, *args, **kwargs):
        dt = _pick_dtype(args, kwargs, 0, dtype_keyword, default_dtype)
        program = args[1] if len(args) > 1 else kwargs.get("program")
        Reducer.__init__(self, ctx, ngroups, group_size, dt, program)

    reducer_cls = type(reducer_name, (Reducer,), {"program_type": program_cls, "__init__": reducer_init,
                                                  "__module__": __name__})
    return program_cls, reducer_cls


def _pick_dtype(args, kwargs, position, keyword, default):
    if len(args) > position:
        return args[position]
    if keyword in kwargs:
        return kwargs[keyword]
    if "value_dtype" in kwargs:
        return kwargs["value_dtype"]
    if default is None:
        raise TypeError("missing dtype argument %r" % keyword)
    return np.dtype(default)

"""Two-stage device reductions over rows of scalars.

Mirrors ``collision/reduce.py`` (ReductionProgram :9-22, Reducer :24-76).  The reference renders
its kernel from a Jinja2 template with a list of ``(init, fn)`` accumulators; here the two
accumulator lists the package uses are compiled in (``COL_OP_MINMAX`` for bounds.py:5,
``COL_OP_SUM`` for summer.py:5), a list of up to four accumulators with ``fn`` one of ``min`` / ``max`` /
``fmin`` / ``fmax`` / ``ADD`` / ``MUL`` and ``init`` a number or ``[-]INFINITY`` goes through the
table-driven ``col_reduce_list``, and ANY other list is rendered into a HIP kernel with the structure
of reduce.cl:5-58 and compiled at run time (hiprtc; ``render_source`` below, csrc/rtc_reduce.hip) -- as
the reference does with OpenCL.  A subclass names its list through ``accumulator``.
``ngroups`` / ``group_size``: the compiled-in and table-driven kernels take them for API parity only
(their launch geometry is their own, csrc/reduce.hip); a rendered kernel runs on exactly
``ngroups x group_size`` work-items like reduce.py:62-76, so a function that is not associative (a float
``ADD``) meets its operands in the reference's order.  Unlike reduce.cl:40-52, every partial is folded
whatever ``ngroups`` and ``group_size`` are (the reference's halving loop drops some when they are not
powers of two).
"""
import ctypes as C
import sys

import numpy as np

from . import hip
from ._lib import call, cdll
from .misc import COL_OP_MINMAX, COL_OP_SUM, ProgramHandle, device_width, type_code

_OPS = {
    (("INFINITY", "min"), ("-INFINITY", "max")): COL_OP_MINMAX,
    (("0", "ADD"),): COL_OP_SUM,
}


_ACC_FN = {"min": 0, "fmin": 0, "max": 1, "fmax": 1, "ADD": 2, "MUL": 3}
MAX_ACCUMULATORS = 4

# ---- rendered kernels (any accumulator list) ----
_C_TYPES = {"int8": "signed char", "uint8": "unsigned char", "int16": "short", "uint16": "unsigned short",
            "int32": "int", "uint32": "unsigned int", "int64": "long", "uint64": "unsigned long",
            "float32": "float", "float64": "double"}
# OpenCL's binary built-ins (and reduce.cl:3's ADD) by the names the template may carry -> functions of the prelude; a name
# that is not listed is rendered as it stands (HIP device code knows most of OpenCL's math functions by the same name)
_CL_NAMES = ("min", "max", "fmin", "fmax", "maxmag", "minmag", "fdim", "copysign", "hypot", "pow", "fmod", "atan2",
             "remainder", "nextafter", "add_sat", "sub_sat", "hadd", "rhadd", "abs_diff", "mul_hi")
_PRELUDE = r"""
#define ADD(x, y) ((x) + (y))                         /* reduce.cl:3 */
#define MUL(x, y) ((x) * (y))
#undef INFINITY
#undef NAN
#define INFINITY (__builtin_inff())
#define NAN (__builtin_nanf(""))
#define HUGE_VALF (__builtin_inff())
#define HUGE_VAL (__builtin_inf())
#define MAXFLOAT __FLT_MAX__
#ifndef FLT_MAX
#define FLT_MAX __FLT_MAX__
#define FLT_MIN __FLT_MIN__
#define FLT_EPSILON __FLT_EPSILON__
#define DBL_MAX __DBL_MAX__
#define DBL_MIN __DBL_MIN__
#define DBL_EPSILON __DBL_EPSILON__
#endif
#ifndef INT_MAX
#define CHAR_BIT 8
#define SCHAR_MAX 127
#define SCHAR_MIN (-128)
#define CHAR_MAX 127
#define CHAR_MIN (-128)
#define UCHAR_MAX 255
#define SHRT_MAX 32767
#define SHRT_MIN (-32768)
#define USHRT_MAX 65535
#define INT_MAX 2147483647
#define INT_MIN (-2147483647 - 1)
#define UINT_MAX 0xffffffffU
#define LONG_MAX 0x7fffffffffffffffL                  /* OpenCL's long is 64 bits wide */
#define LONG_MIN (-0x7fffffffffffffffL - 1)
#define ULONG_MAX 0xffffffffffffffffUL
#endif
template <typename X> struct cl_unsigned { typedef X type; };
template <> struct cl_unsigned<signed char> { typedef unsigned char type; };
template <> struct cl_unsigned<short> { typedef unsigned short type; };
template <> struct cl_unsigned<int> { typedef unsigned int type; };
template <> struct cl_unsigned<long> { typedef unsigned long type; };
template <typename X> struct cl_is_signed { static constexpr bool value = (X)(-1) < (X)0; };
template <typename X> __device__ inline X cl_min(X a, X b) { return b < a ? b : a; }        /* OpenCL 6.15.4 / 6.15.3 */
template <typename X> __device__ inline X cl_max(X a, X b) { return a < b ? b : a; }
__device__ inline float cl_fmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ inline double cl_fmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ inline float cl_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ inline double cl_fmax(double a, double b) { return __builtin_fmax(a, b); }
template <typename X> __device__ inline X cl_fabs(X a) { return a < (X)0 ? -a : a; }
template <typename X> __device__ inline X cl_maxmag(X a, X b) { const X x = cl_fabs(a), y = cl_fabs(b); return x > y ? a : (y > x ? b : cl_fmax(a, b)); }
template <typename X> __device__ inline X cl_minmag(X a, X b) { const X x = cl_fabs(a), y = cl_fabs(b); return x < y ? a : (y < x ? b : cl_fmin(a, b)); }
template <typename X> __device__ inline X cl_fdim(X a, X b) { return (a != a || b != b) ? (X)NAN : (a > b ? a - b : (X)0); }
__device__ inline float cl_copysign(float a, float b) { return __builtin_copysignf(a, b); }
__device__ inline double cl_copysign(double a, double b) { return __builtin_copysign(a, b); }
__device__ inline float cl_hypot(float a, float b) { return hypotf(a, b); }
__device__ inline double cl_hypot(double a, double b) { return hypot(a, b); }
__device__ inline float cl_pow(float a, float b) { return powf(a, b); }
__device__ inline double cl_pow(double a, double b) { return pow(a, b); }
__device__ inline float cl_fmod(float a, float b) { return fmodf(a, b); }
__device__ inline double cl_fmod(double a, double b) { return fmod(a, b); }
__device__ inline float cl_atan2(float a, float b) { return atan2f(a, b); }
__device__ inline double cl_atan2(double a, double b) { return atan2(a, b); }
__device__ inline float cl_remainder(float a, float b) { return remainderf(a, b); }
__device__ inline double cl_remainder(double a, double b) { return remainder(a, b); }
__device__ inline float cl_nextafter(float a, float b) { return nextafterf(a, b); }
__device__ inline double cl_nextafter(double a, double b) { return nextafter(a, b); }
template <typename X> __device__ inline X cl_add_sat(X a, X b) {
    X r;
    if (!__builtin_add_overflow(a, b, &r)) return r;
    if (!cl_is_signed<X>::value) return (X)~(X)0;
    typedef typename cl_unsigned<X>::type U;
    const X hi = (X)((U)~(U)0 >> 1);
    return b < (X)0 ? (X)(-hi - 1) : hi;
}
template <typename X> __device__ inline X cl_sub_sat(X a, X b) {
    X r;
    if (!__builtin_sub_overflow(a, b, &r)) return r;
    if (!cl_is_signed<X>::value) return (X)0;
    typedef typename cl_unsigned<X>::type U;
    const X hi = (X)((U)~(U)0 >> 1);
    return b > (X)0 ? (X)(-hi - 1) : hi;
}
template <typename X> __device__ inline X cl_hadd(X a, X b) { return (X)((a & b) + ((a ^ b) >> 1)); }
template <typename X> __device__ inline X cl_rhadd(X a, X b) { return (X)((a | b) - ((a ^ b) >> 1)); }
template <typename X> __device__ inline X cl_abs_diff(X a, X b) {
    typedef typename cl_unsigned<X>::type U;
    return (X)(a > b ? (U)a - (U)b : (U)b - (U)a);
}
template <typename X> __device__ inline X cl_mul_hi(X a, X b) {
    if (sizeof(X) < 8) {
        if (cl_is_signed<X>::value) return (X)(((long)a * (long)b) >> (8 * sizeof(X)));
        return (X)(((unsigned long)a * (unsigned long)b) >> (8 * sizeof(X)));
    }
    if (cl_is_signed<X>::value) return (X)(((__int128)a * (__int128)b) >> 64);
    return (X)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
}
"""


def render_source(value_dtype, accumulator):
    """The HIP text of reduce.cl:5-58 for one accumulator list: kernels ``bounds1(values, n, group_accs)`` and
    ``bounds2(group_accs, output)``.  VALDTYPE may be a vector: OpenCL applies its built-ins component by component, so
    the rendered code loops over the W scalars of a row.  The tree over a group's scratch rows folds row ``l + o`` into
    row ``l`` for o = top / 2 ... 1 (reduce.cl:22-33 when the size is a power of two; otherwise every row still gets
    folded, where the reference's ``o = size / 2`` loop drops some)."""
    vd = np.dtype(value_dtype)
    if vd.base.name not in _C_TYPES:
        raise ValueError("Unsupported element dtype for a rendered reduction: {}".format(vd.base.name))
    width = device_width(vd)
    fns = [("cl_" + fn) if fn in _CL_NAMES else fn for _, fn in accumulator]
    inits = [str(init) for init, _ in accumulator]
    k = len(accumulator)
    fold_value = "\n".join("            acc[%d][c] = %s(acc[%d][c], v);" % (a, fns[a], a) for a in range(k))
    fold_rows = "\n".join("                s[l][%d][c] = %s(s[l][%d][c], s[l + o][%d][c]);" % (a, fns[a], a, a) for a in range(k))
    init_rows = "\n".join("        acc[%d][c] = (S)(%s);" % (a, inits[a]) for a in range(k))
    body = r"""
typedef %(ctype)s S;
#define W %(width)d
#define NACC %(k)d
extern "C" __global__ void bounds1(const S *__restrict__ values, unsigned long long n, S *__restrict__ group_accs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    S (*s)[NACC][W] = (S (*)[NACC][W])lds_raw;                       /* local VALDTYPE (*scratch)[ACC_SIZE], reduce.cl:8 */
    S acc[NACC][W];
    for (int c = 0; c < W; c++) {
%(init_rows)s
    }
    const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
        for (int c = 0; c < W; c++) {
            const S v = values[i * W + c];
%(fold_value)s
        }
    const unsigned l = threadIdx.x;
    for (int a = 0; a < NACC; a++)
        for (int c = 0; c < W; c++) s[l][a][c] = acc[a][c];
    __syncthreads();
    unsigned top = 1;
    while (top < blockDim.x) top <<= 1;
    for (unsigned o = top / 2; o > 0; o /= 2) {
        if (l < o && l + o < blockDim.x)
            for (int c = 0; c < W; c++) {
%(fold_rows)s
            }
        __syncthreads();
    }
    if (l == 0)
        for (int a = 0; a < NACC; a++)
            for (int c = 0; c < W; c++) group_accs[((unsigned long long)blockIdx.x * NACC + a) * W + c] = s[0][a][c];
}

extern "C" __global__ void bounds2(const S *__restrict__ group_accs, S *__restrict__ output) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    S (*s)[NACC][W] = (S (*)[NACC][W])lds_raw;
    const unsigned l = threadIdx.x;
    for (int a = 0; a < NACC; a++)
        for (int c = 0; c < W; c++) s[l][a][c] = group_accs[((unsigned long long)l * NACC + a) * W + c];
    __syncthreads();
    unsigned top = 1;
    while (top < blockDim.x) top <<= 1;
    for (unsigned o = top / 2; o > 0; o /= 2) {
        if (l < o && l + o < blockDim.x)
            for (int c = 0; c < W; c++) {
%(fold_rows)s
            }
        __syncthreads();
    }
    if (l == 0)
        for (int a = 0; a < NACC; a++)
            for (int c = 0; c < W; c++) output[a * W + c] = s[0][a][c];
}
""" % dict(ctype=_C_TYPES[vd.base.name], width=width, k=k, init_rows=init_rows, fold_value=fold_value, fold_rows=fold_rows)
    return _PRELUDE + body


class _BadInitialValue(ValueError):
    """A NUMERIC initial value the accumulator's dtype cannot hold (out of range / not an integer): rejected, where a
    value that is not a number at all is an expression for the compiler."""


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
            raise _BadInitialValue("Initial value {} is not an integer ({} accumulator)".format(text, base))
        exact = int(value)
    value = float(exact)
    info = np.iinfo(base)
    if not info.min <= exact <= info.max:
        raise _BadInitialValue("Initial value {} outside the range of {}".format(text, base))
    return value, exact - (1 << 64) if exact >= (1 << 63) else exact


class ReductionProgram(ProgramHandle):
    accumulator = None      # set by subclasses, as in the reference (bounds.py:5, summer.py:5)

    def __init__(self, ctx, value_dtype):
        self.value_dtype = np.dtype(value_dtype)
        key = tuple(tuple(a) for a in (self.accumulator or ()))
        if not key:
            raise ValueError("Unsupported accumulator list: {}".format(self.accumulator))
        self.op = _OPS.get(key)
        self.rtc = None
        self.width = device_width(self.value_dtype)
        self.acc_dtype = np.dtype((self.value_dtype, len(key)))
        super().__init__(ctx)
        if self.op is None and not self._table_driven(key):
            # anything else: rendered and compiled now, like the reference's Program (misc.py:6-22)
            self.source = render_source(self.value_dtype, key)
            log = C.create_string_buffer(1 << 14)
            handle = C.c_void_p()
            if cdll().col_reduce_rtc_create(self.source.encode(), log, len(log), C.byref(handle)) != 0:
                raise ValueError("Unsupported accumulator list: {}\n{}".format(self.accumulator, log.value.decode(errors="replace")))
            self.rtc = handle
            self.acc_bytes = len(key) * self.width * self.value_dtype.base.itemsize
            return
        self.type_code = type_code(self.value_dtype)

    def _table_driven(self, key):
        """Fills in the table-driven reducer's arguments if the list fits it (<= 4 accumulators, the six names, numeric
        initial values, a dtype the compiled kernels have)."""
        if len(key) > MAX_ACCUMULATORS or any(fn not in _ACC_FN for _, fn in key):
            return False
        if self.value_dtype.base.name not in ("float32", "float64", "uint32", "int32", "uint64", "int64"):
            return False
        try:
            parsed = [_parse_init(init, self.value_dtype) for init, _ in key]
        except _BadInitialValue:
            raise
        except ValueError:
            return False                           # an initial value that is an expression: let the compiler read it
        self.acc_ops = (C.c_int * len(key))(*[_ACC_FN[fn] for _, fn in key])
        self.acc_inits = (C.c_double * len(key))(*[v for v, _ in parsed])
        self.acc_int_inits = None              # exact integer initial values for the integer dtypes
        if all(e is not None for _, e in parsed):
            self.acc_int_inits = (C.c_int64 * len(key))(*[e for _, e in parsed])
        return True

    def __del__(self):
        rtc, self.rtc = getattr(self, "rtc", None), None
        if rtc and not sys.is_finalizing():        # (at interpreter shutdown the HIP runtime may be gone already)
            try:
                cdll().col_reduce_rtc_destroy(rtc)
            except Exception:
                pass


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
        if p.rtc is not None:
            if self.ngroups > 1024 or self.group_size > 1024 or p.acc_bytes * max(self.ngroups, self.group_size) > 65536:
                raise ValueError("ngroups / group_size too large for a rendered reduction "
                                 "(at most 1024 each and 64 KB of accumulators per group)")
            need = self.ngroups * p.acc_bytes
            if self._scratch is None or self._scratch.size < need:
                self._scratch = hip.Buffer(p.context, need)
            cq.wait_for(wait_for)
            call.col_reduce_rtc(cq.stream, p.rtc, values_buf.ptr, size, self.ngroups, self.group_size, p.acc_bytes,
                                self._scratch.ptr, output_buf.ptr)
            return hip.Event(cq)
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

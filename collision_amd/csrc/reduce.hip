// Two-stage reduction (scene bounds, sums).
// Replaces Reducer.reduce = bounds1 + bounds2 (collision/reduce.py:62-76, collision/reduce.cl:5-58)
// with the accumulator lists of collision/bounds.py:5 (min,max) and collision/summer.py:5 (sum).
//
// HBM-bound: one 16/32-byte row load per sphere.  Stage 1: grid-stride rows, register
// accumulators, wave shuffle reduce, LDS across the block's 4 waves, one partial per block.
// Stage 2: one block folds the partials.  min/max are exact, so the split does not matter.
#include "col_common.h"
#include <math.h>

namespace {

constexpr int RT = 256;          // threads per block
constexpr int RMAX_BLOCKS = 1024;

template <typename T, int W> struct Row { T v[W]; };

template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int o) { return __shfl_xor(v, o, COL_WAVE); }

template <typename T> __device__ __forceinline__ T pos_inf();
template <> __device__ __forceinline__ float pos_inf<float>() { return INFINITY; }
template <> __device__ __forceinline__ double pos_inf<double>() { return (double)INFINITY; }
// integer "infinities" only matter for COL_OP_MINMAX on integer types
template <> __device__ __forceinline__ uint32_t pos_inf<uint32_t>() { return 0xFFFFFFFFu; }
template <> __device__ __forceinline__ int32_t pos_inf<int32_t>() { return 0x7FFFFFFF; }
template <> __device__ __forceinline__ uint64_t pos_inf<uint64_t>() { return ~0ull; }
template <> __device__ __forceinline__ int64_t pos_inf<int64_t>() { return 0x7FFFFFFFFFFFFFFFll; }
template <typename T> __device__ __forceinline__ T neg_inf();
template <> __device__ __forceinline__ float neg_inf<float>() { return -INFINITY; }
template <> __device__ __forceinline__ double neg_inf<double>() { return -(double)INFINITY; }
template <> __device__ __forceinline__ uint32_t neg_inf<uint32_t>() { return 0u; }
template <> __device__ __forceinline__ int32_t neg_inf<int32_t>() { return (int32_t)0x80000000; }
template <> __device__ __forceinline__ uint64_t neg_inf<uint64_t>() { return 0ull; }
template <> __device__ __forceinline__ int64_t neg_inf<int64_t>() { return (int64_t)0x8000000000000000ll; }

// Accumulator: A[0..NACC*W). MINMAX: [mins, maxes]; SUM: [sums].
template <typename T, int W, int OP> struct Acc {
    static constexpr int N = (OP == COL_OP_MINMAX ? 2 : 1) * W;
    T a[N];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (OP == COL_OP_MINMAX) { a[i] = pos_inf<T>(); a[W + i] = neg_inf<T>(); }
            else a[i] = (T)0;
        }
    }
    __device__ __forceinline__ void add_row(const Row<T, W> &r) {
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (OP == COL_OP_MINMAX) {
                a[i] = r.v[i] < a[i] ? r.v[i] : a[i];
                a[W + i] = r.v[i] > a[W + i] ? r.v[i] : a[W + i];
            } else a[i] += r.v[i];
        }
    }
    __device__ __forceinline__ void merge(const T *o) {
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (OP == COL_OP_MINMAX) {
                a[i] = o[i] < a[i] ? o[i] : a[i];
                a[W + i] = o[W + i] > a[W + i] ? o[W + i] : a[W + i];
            } else a[i] += o[i];
        }
    }
    __device__ __forceinline__ void wave_reduce() {
#pragma unroll
        for (int o = COL_WAVE / 2; o > 0; o >>= 1) {
            T t[N];
#pragma unroll
            for (int i = 0; i < N; i++) t[i] = shfl_xor_t(a[i], o);
            merge(t);
        }
    }
};

template <typename T, int W, int OP>
__device__ __forceinline__ void block_fold(Acc<T, W, OP> &acc, T *out) {
    constexpr int N = Acc<T, W, OP>::N;
    __shared__ T s[(RT / COL_WAVE) * N];
    acc.wave_reduce();
    const u32 lane = lane_id(), w = threadIdx.x / COL_WAVE;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; i++) s[w * N + i] = acc.a[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < RT / COL_WAVE; k++) acc.merge(&s[k * N]);
#pragma unroll
        for (int i = 0; i < N; i++) out[i] = acc.a[i];
    }
}

template <typename T, int W, int OP>
__global__ __launch_bounds__(RT) void k_reduce1(const Row<T, W> *__restrict__ rows, uint64_t n, T *partials) {
    Acc<T, W, OP> acc;
    acc.init();
    const uint64_t stride = (uint64_t)gridDim.x * RT;
    uint64_t i = (uint64_t)blockIdx.x * RT + threadIdx.x;
    // EIGHT rows in flight per thread for the [min, max] of 4-wide rows -- the scene bounds of the whole path, whose grid is at
    // most COL_MINMAX_PARTS = 256 blocks (one per CU, four waves): with four rows a 16 M-sphere scene read at 4.7 TB/s.  min / max
    // are exact and order-independent; the other accumulators keep their order of additions (four rows, then one at a time).
    if constexpr (OP == COL_OP_MINMAX && W == 4) {
        for (; i + 7 * stride < n; i += 8 * stride) {
            Row<T, W> r[8];
#pragma unroll
            for (int u = 0; u < 8; u++) r[u] = rows[i + u * stride];
#pragma unroll
            for (int u = 0; u < 8; u++) acc.add_row(r[u]);
        }
    }
    // 4 independent rows in flight per thread
    for (; i + 3 * stride < n; i += 4 * stride) {
        Row<T, W> r0 = rows[i], r1 = rows[i + stride], r2 = rows[i + 2 * stride], r3 = rows[i + 3 * stride];
        acc.add_row(r0); acc.add_row(r1); acc.add_row(r2); acc.add_row(r3);
    }
    for (; i < n; i += stride) acc.add_row(rows[i]);
    block_fold<T, W, OP>(acc, partials + (size_t)blockIdx.x * Acc<T, W, OP>::N);
}

// k_reduce1 for [min row, max row] of 4-wide rows with the row count read from DEVICE memory (clamped to n_max): the
// multi-GPU path enqueues it right behind the launch that produces the count, before the host knows it (multi.py)
template <typename T>
__global__ __launch_bounds__(RT) void k_minmax4_dev(const Row<T, 4> *__restrict__ rows, const u32 *__restrict__ n_dev, u32 n_max,
                                                    T *partials) {
    Acc<T, 4, COL_OP_MINMAX> acc;
    acc.init();
    const uint64_t n = min(*n_dev, n_max);
    const uint64_t stride = (uint64_t)gridDim.x * RT;
    uint64_t i = (uint64_t)blockIdx.x * RT + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {           // (eight rows in flight: see k_reduce1)
        Row<T, 4> r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] = rows[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) acc.add_row(r[u]);
    }
    for (; i + 3 * stride < n; i += 4 * stride) {
        Row<T, 4> r0 = rows[i], r1 = rows[i + stride], r2 = rows[i + 2 * stride], r3 = rows[i + 3 * stride];
        acc.add_row(r0); acc.add_row(r1); acc.add_row(r2); acc.add_row(r3);
    }
    for (; i < n; i += stride) acc.add_row(rows[i]);
    block_fold<T, 4, COL_OP_MINMAX>(acc, partials + (size_t)blockIdx.x * Acc<T, 4, COL_OP_MINMAX>::N);
}

template <typename T, int W, int OP>
__global__ __launch_bounds__(RT) void k_reduce2(const T *partials, uint32_t nparts, T *out) {
    constexpr int N = Acc<T, W, OP>::N;
    Acc<T, W, OP> acc;
    acc.init();
    for (uint32_t i = threadIdx.x; i < nparts; i += RT) acc.merge(partials + (size_t)i * N);
    block_fold<T, W, OP>(acc, out);
}

template <typename T, int W, int OP>
int launch(void *stream, const void *values, uint64_t n, void *scratch, void *out) {
    uint64_t blocks = col_ceil_div(n, (uint64_t)RT * 4);
    if (blocks > RMAX_BLOCKS) blocks = RMAX_BLOCKS;
    if (blocks == 0) blocks = 1;
    k_reduce1<T, W, OP><<<dim3((unsigned)blocks), dim3(RT), 0, col_stream(stream)>>>(
        (const Row<T, W> *)values, n, (T *)scratch);
    COL_LAUNCH_OK();
    k_reduce2<T, W, OP><<<dim3(1), dim3(RT), 0, col_stream(stream)>>>((const T *)scratch, (uint32_t)blocks, (T *)out);
    COL_LAUNCH_OK();
    return COL_OK;
}

template <typename T, int OP>
int by_width(void *stream, const void *values, uint64_t n, int width, void *scratch, void *out) {
    switch (width) {
    case 1: return launch<T, 1, OP>(stream, values, n, scratch, out);
    case 2: return launch<T, 2, OP>(stream, values, n, scratch, out);
    case 4: return launch<T, 4, OP>(stream, values, n, scratch, out);
    case 8: return launch<T, 8, OP>(stream, values, n, scratch, out);
    case 16: return launch<T, 16, OP>(stream, values, n, scratch, out);
    default: return COL_EINVAL;
    }
}

template <typename T>
int by_op(void *stream, const void *values, uint64_t n, int width, int op, void *scratch, void *out) {
    if (op == COL_OP_MINMAX) return by_width<T, COL_OP_MINMAX>(stream, values, n, width, scratch, out);
    if (op == COL_OP_SUM) return by_width<T, COL_OP_SUM>(stream, values, n, width, scratch, out);
    return COL_EINVAL;
}

// ---- generic accumulator lists (reduce.py:9-22 renders any list of (init, fn) pairs into reduce.cl) ----
// Table-driven instead of rendered: up to COL_REDUCE_MAX_ACC accumulators, each an op code
// (COL_ACC_MIN / MAX / ADD / MUL -- the binary functions the template can name: min, max, fmin, fmax, the
// ADD macro, and MUL as its obvious sibling) and an initial value.  Output: one row of `width` scalars per
// accumulator, in list order (ACC_SIZE consecutive VALDTYPEs, reduce.cl:34-37).  HBM-bound like the two
// compiled-in lists; the op dispatch is a wave-uniform switch.
struct AccList { int n; int op[COL_REDUCE_MAX_ACC]; double init[COL_REDUCE_MAX_ACC]; long long iinit[COL_REDUCE_MAX_ACC]; int exact[COL_REDUCE_MAX_ACC]; };

// `exact`: the initial value of an integer list arrived as a 64-bit integer (a double cannot carry values beyond
// 2^53, e.g. UINT64_MAX as the start of a min list); u64 values travel as their two's-complement bit pattern
template <typename T> __device__ __forceinline__ T init_value(const AccList &L, int k) {
    if (k >= L.n) return (T)0;
    const double v = L.init[k];
    if (v == (double)INFINITY) return pos_inf<T>();
    if (v == -(double)INFINITY) return neg_inf<T>();
    if constexpr (!__is_floating_point(T)) { if (L.exact[k]) return (T)(unsigned long long)L.iinit[k]; }
    return (T)v;
}
template <typename T> __device__ __forceinline__ T apply_op(int op, T a, T b) {
    switch (op) {
    case COL_ACC_MIN: return b < a ? b : a;
    case COL_ACC_MAX: return b > a ? b : a;
    case COL_ACC_MUL: return a * b;
    default: return a + b;
    }
}

template <typename T, int W>
__device__ __forceinline__ void fold_list(const AccList &L, T (&a)[COL_REDUCE_MAX_ACC][W], T *out) {
    __shared__ T s[(RT / COL_WAVE) * COL_REDUCE_MAX_ACC * W];
    const u32 lane = lane_id(), w = threadIdx.x / COL_WAVE;
    for (int k = 0; k < L.n; k++) {
#pragma unroll
        for (int o = COL_WAVE / 2; o > 0; o >>= 1) {
#pragma unroll
            for (int i = 0; i < W; i++) a[k][i] = apply_op(L.op[k], a[k][i], shfl_xor_t(a[k][i], o));
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < W; i++) s[(w * COL_REDUCE_MAX_ACC + k) * W + i] = a[k][i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < L.n; k++) {
            for (int i = 0; i < W; i++) {
                T v = a[k][i];
                for (int ww = 1; ww < RT / COL_WAVE; ww++) v = apply_op(L.op[k], v, s[(ww * COL_REDUCE_MAX_ACC + k) * W + i]);
                out[k * W + i] = v;
            }
        }
    }
}

template <typename T, int W>
__global__ __launch_bounds__(RT) void k_reduce_list1(const Row<T, W> *__restrict__ rows, uint64_t n, AccList L, T *partials) {
    T a[COL_REDUCE_MAX_ACC][W];
    for (int k = 0; k < COL_REDUCE_MAX_ACC; k++)
#pragma unroll
        for (int i = 0; i < W; i++) a[k][i] = init_value<T>(L, k);
    const uint64_t stride = (uint64_t)gridDim.x * RT;
    for (uint64_t i = (uint64_t)blockIdx.x * RT + threadIdx.x; i < n; i += stride) {
        const Row<T, W> r = rows[i];
        for (int k = 0; k < L.n; k++)
#pragma unroll
            for (int c = 0; c < W; c++) a[k][c] = apply_op(L.op[k], a[k][c], r.v[c]);
    }
    fold_list<T, W>(L, a, partials + (size_t)blockIdx.x * L.n * W);
}

template <typename T, int W>
__global__ __launch_bounds__(RT) void k_reduce_list2(const T *partials, uint32_t nparts, AccList L, T *out) {
    T a[COL_REDUCE_MAX_ACC][W];
    for (int k = 0; k < COL_REDUCE_MAX_ACC; k++)
#pragma unroll
        for (int i = 0; i < W; i++) a[k][i] = init_value<T>(L, k);
    for (uint32_t p = threadIdx.x; p < nparts; p += RT)
        for (int k = 0; k < L.n; k++)
#pragma unroll
            for (int c = 0; c < W; c++) a[k][c] = apply_op(L.op[k], a[k][c], partials[((size_t)p * L.n + k) * W + c]);
    fold_list<T, W>(L, a, out);
}

template <typename T, int W>
int launch_list(void *stream, const void *values, uint64_t n, const AccList &L, void *scratch, void *out) {
    uint64_t blocks = col_ceil_div(n, (uint64_t)RT * 4);
    if (blocks > RMAX_BLOCKS) blocks = RMAX_BLOCKS;
    if (blocks == 0) blocks = 1;
    k_reduce_list1<T, W><<<dim3((unsigned)blocks), dim3(RT), 0, col_stream(stream)>>>((const Row<T, W> *)values, n, L, (T *)scratch);
    COL_LAUNCH_OK();
    k_reduce_list2<T, W><<<dim3(1), dim3(RT), 0, col_stream(stream)>>>((const T *)scratch, (uint32_t)blocks, L, (T *)out);
    COL_LAUNCH_OK();
    return COL_OK;
}

template <typename T>
int list_by_width(void *stream, const void *values, uint64_t n, int width, const AccList &L, void *scratch, void *out) {
    switch (width) {
    case 1: return launch_list<T, 1>(stream, values, n, L, scratch, out);
    case 2: return launch_list<T, 2>(stream, values, n, L, scratch, out);
    case 4: return launch_list<T, 4>(stream, values, n, L, scratch, out);
    case 8: return launch_list<T, 8>(stream, values, n, L, scratch, out);
    case 16: return launch_list<T, 16>(stream, values, n, L, scratch, out);
    default: return COL_EINVAL;
    }
}

}  // namespace

extern "C" {

int col_minmax4_stage1(void *stream, const void *rows, uint64_t n, int coord_bytes, void *partials, uint32_t *parts) {
    uint64_t blocks = col_ceil_div(n, (uint64_t)RT * 16);
    if (blocks > COL_MINMAX_PARTS) blocks = COL_MINMAX_PARTS;
    if (blocks == 0) blocks = 1;
    *parts = (uint32_t)blocks;
    if (coord_bytes == 4)
        k_reduce1<float, 4, COL_OP_MINMAX><<<dim3((unsigned)blocks), dim3(RT), 0, col_stream(stream)>>>(
            (const Row<float, 4> *)rows, n, (float *)partials);
    else if (coord_bytes == 8)
        k_reduce1<double, 4, COL_OP_MINMAX><<<dim3((unsigned)blocks), dim3(RT), 0, col_stream(stream)>>>(
            (const Row<double, 4> *)rows, n, (double *)partials);
    else
        return COL_EINVAL;
    COL_LAUNCH_OK();
    return COL_OK;
}

// The same partials with the row count in device memory (*n_dev, at most n_max): the grid is sized for n_max, blocks
// without rows leave the identity, and min / max being exact and order-independent the fold gives the bits of
// col_minmax4_stage1 on the true count.
int col_minmax4_stage1_dev(void *stream, const void *rows, const uint32_t *n_dev, uint32_t n_max, int coord_bytes, void *partials,
                           uint32_t *parts) {
    if (!n_dev || !partials || !parts) return COL_EINVAL;
    uint64_t blocks = col_ceil_div(n_max, (uint64_t)RT * 16);
    if (blocks > COL_MINMAX_PARTS) blocks = COL_MINMAX_PARTS;
    if (blocks == 0) blocks = 1;
    *parts = (uint32_t)blocks;
    if (coord_bytes == 4)
        k_minmax4_dev<float><<<dim3((unsigned)blocks), dim3(RT), 0, col_stream(stream)>>>((const Row<float, 4> *)rows, n_dev, n_max, (float *)partials);
    else if (coord_bytes == 8)
        k_minmax4_dev<double><<<dim3((unsigned)blocks), dim3(RT), 0, col_stream(stream)>>>((const Row<double, 4> *)rows, n_dev, n_max, (double *)partials);
    else
        return COL_EINVAL;
    COL_LAUNCH_OK();
    return COL_OK;
}

size_t col_reduce_scratch_bytes(int dtype, int width) {
    size_t eb = (dtype == COL_F32 || dtype == COL_U32 || dtype == COL_I32) ? 4 : 8;
    return (size_t)RMAX_BLOCKS * COL_REDUCE_MAX_ACC * (size_t)width * eb;
}

int col_reduce_list(void *stream, const void *values, uint64_t n, int dtype, int width, int n_acc, const int *ops,
                    const double *inits, const int64_t *int_inits, void *scratch, void *out) {
    if (!scratch) return COL_ENOSCRATCH;
    if (n_acc < 1 || n_acc > COL_REDUCE_MAX_ACC || !ops || !inits) return COL_EINVAL;
    AccList L;
    L.n = n_acc;
    for (int k = 0; k < COL_REDUCE_MAX_ACC; k++) {
        L.op[k] = k < n_acc ? ops[k] : COL_ACC_ADD;
        L.init[k] = k < n_acc ? inits[k] : 0.0;
        L.exact[k] = k < n_acc && int_inits != nullptr;
        L.iinit[k] = L.exact[k] ? (long long)int_inits[k] : 0;
        if (L.op[k] < COL_ACC_MIN || L.op[k] > COL_ACC_MUL) return COL_EINVAL;
    }
    switch (dtype) {
    case COL_F32: return list_by_width<float>(stream, values, n, width, L, scratch, out);
    case COL_F64: return list_by_width<double>(stream, values, n, width, L, scratch, out);
    case COL_U32: return list_by_width<uint32_t>(stream, values, n, width, L, scratch, out);
    case COL_I32: return list_by_width<int32_t>(stream, values, n, width, L, scratch, out);
    case COL_U64: return list_by_width<uint64_t>(stream, values, n, width, L, scratch, out);
    case COL_I64: return list_by_width<int64_t>(stream, values, n, width, L, scratch, out);
    default: return COL_EINVAL;
    }
}

int col_reduce(void *stream, const void *values, uint64_t n, int dtype, int width, int op,
               void *scratch, void *out) {
    if (!scratch) return COL_ENOSCRATCH;
    switch (dtype) {
    case COL_F32: return by_op<float>(stream, values, n, width, op, scratch, out);
    case COL_F64: return by_op<double>(stream, values, n, width, op, scratch, out);
    case COL_U32: return by_op<uint32_t>(stream, values, n, width, op, scratch, out);
    case COL_I32: return by_op<int32_t>(stream, values, n, width, op, scratch, out);
    case COL_U64: return by_op<uint64_t>(stream, values, n, width, op, scratch, out);
    case COL_I64: return by_op<int64_t>(stream, values, n, width, op, scratch, out);
    default: return COL_EINVAL;
    }
}

}  // extern "C"

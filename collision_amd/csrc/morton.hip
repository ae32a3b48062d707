// 30-bit Morton codes + id ramp + padding.
// Replaces `range`, the 0xFFFFFFFF padding fill and `calculateCodes`
// (collision/collision.py:137-146,161-165; collision/collision.cl:8-40) with one launch.
//
// HBM-bound: one 16-byte (f32) / 32-byte (f64) row load, 4+4 bytes stored per sphere.
// Arithmetic must be IEEE with one rounding per operation (this file is compiled with
// -ffp-contract=off and correctly rounded division) so codes match the oracle bit for bit.
#include "col_common.h"

namespace {

// expand_bits / quantize (collision.cl:14-31) live in col_common.h: multi.hip computes the same codes

template <typename T> struct alignas(4 * sizeof(T)) Vec4 { T x, y, z, w; };
template <> struct alignas(16) Vec4<double> { double x, y, z, w; };

// The word that is about to be zeroed is the pair counter of the PREVIOUS call on the same buffers: publish it in a
// host-visible word (0x80000000 | count) before it goes -- the host reads it when it makes its next call and chooses the
// traversal's allocation scheme by it (dense scenes: chunked, bvh.hip), with no launch and no sync spent on it.
__device__ __forceinline__ void publish_count(u32 *publish, u32 count) {
    if (publish) __hip_atomic_store(publish, 0x80000000u | (count < 0x7FFFFFFFu ? count : 0x7FFFFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
__global__ __launch_bounds__(256) void k_morton(const Vec4<T> *__restrict__ coords,
                                                 const Vec4<T> *__restrict__ range, u32 n, u32 padded,
                                                 u32 *__restrict__ codes, u32 *__restrict__ ids,
                                                 const T *__restrict__ radii, Vec4<T> *__restrict__ packed,
                                                 u32 *__restrict__ zero_word, u32 *publish) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && zero_word) {                          // the pair counter (collision.py:151-154), no extra launch
        publish_count(publish, *zero_word);
        *zero_word = 0;
    }
    if (i >= padded) return;
    u32 code = 0xFFFFFFFFu;   // collision.py:137-142
    if (i < n) {
        const Vec4<T> mn = range[0], mx = range[1];
        const Vec4<T> c = coords[i];
        if (packed) {                                   // (x, y, z, r) rows: the refit gathers ONE row per leaf
            Vec4<T> pr = c;
            pr.w = radii[i];
            packed[i] = pr;
        }
        code = (expand_bits(quantize(c.x, mn.x, mx.x)) << 2) + (expand_bits(quantize(c.y, mn.y, mx.y)) << 1) +
               expand_bits(quantize(c.z, mn.z, mx.z));
    }
    codes[i] = code;
    if (ids) ids[i] = i;      // collision.cl:8-10
}

// The same codes, for col_collide's fused front end: one block per tile of the radix sort (1024 codes
// below 1 Mi spheres, 4096 up to 8 Mi, 8192 above: col_radix_tile), 4 rows per thread and HALVES passes over them
// (the 8192-code tile is two passes of 1024 threads: the register footprint of the 4096-code one).
// The block folds the `parts` partial [min row, max row] results of the bounds reduction itself
// (min/max are exact and order-independent, so every block gets the bits of a stage-2 launch), and
// counts its tile's digits for the sort's first pass in LDS while the codes are in registers.
constexpr int MT_ROWS = 4;
template <typename T, int NT, int HALVES>
__global__ __launch_bounds__(NT) void k_morton_tile(const Vec4<T> *__restrict__ coords, const T *__restrict__ partials,
                                                     u32 parts, u32 n_bound, u32 padded, u32 *__restrict__ codes,
                                                     u32 *__restrict__ ids, const T *__restrict__ radii,
                                                     Vec4<T> *__restrict__ packed, u32 *__restrict__ zero_word,
                                                     u32 *__restrict__ hist0, u32 nblocks, int hist_shift, u32 *publish,
                                                     const u32 *__restrict__ n_dev) {
    constexpr int TILE = NT * MT_ROWS * HALVES, NW = NT / 64;
    const u32 n = count_of(n_bound, n_dev);       // rows from n on are pads (0xFFFFFFFF), whatever the bound
    __shared__ T s_fold[NW][8];
    __shared__ u32 s_hist[256];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (blockIdx.x == 0 && tid == 0 && zero_word) {
        publish_count(publish, *zero_word);
        *zero_word = 0;
    }
    if (tid < 256) s_hist[tid] = 0;
    // this thread's rows first: their loads overlap the fold of the partials below
    Vec4<T> c[MT_ROWS];
    T rad[MT_ROWS];
#pragma unroll
    for (int k = 0; k < MT_ROWS; k++) {
        const u32 i = blockIdx.x * TILE + k * NT + tid;
        if (i < n) {
            c[k] = coords[i];
            if (packed) rad[k] = radii[i];
        }
    }
    T acc[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { acc[k] = (T)INFINITY; acc[4 + k] = -(T)INFINITY; }
    for (u32 i = tid; i < parts; i += NT) {
        const T *q = partials + (size_t)i * 8;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            acc[k] = q[k] < acc[k] ? q[k] : acc[k];
            acc[4 + k] = q[4 + k] > acc[4 + k] ? q[4 + k] : acc[4 + k];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const T lo = __shfl_xor(acc[k], o, 64), hi = __shfl_xor(acc[4 + k], o, 64);
            acc[k] = lo < acc[k] ? lo : acc[k];
            acc[4 + k] = hi > acc[4 + k] ? hi : acc[4 + k];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) s_fold[w][k] = acc[k];
    }
    __syncthreads();
    Vec4<T> mn, mx;
    {
        T r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = s_fold[0][k];
        for (int ww = 1; ww < NW; ww++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                r[k] = s_fold[ww][k] < r[k] ? s_fold[ww][k] : r[k];
                r[4 + k] = s_fold[ww][4 + k] > r[4 + k] ? s_fold[ww][4 + k] : r[4 + k];
            }
        }
        mn.x = r[0]; mn.y = r[1]; mn.z = r[2]; mn.w = r[3];
        mx.x = r[4]; mx.y = r[5]; mx.z = r[6]; mx.w = r[7];
    }
#pragma unroll
    for (int h = 0; h < HALVES; h++) {
        if (h > 0) {              // the next MT_ROWS rows of this thread (the first ones were loaded above)
#pragma unroll
            for (int k = 0; k < MT_ROWS; k++) {
                const u32 i = blockIdx.x * TILE + (h * MT_ROWS + k) * NT + tid;
                if (i < n) {
                    c[k] = coords[i];
                    if (packed) rad[k] = radii[i];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MT_ROWS; k++) {
            const u32 i = blockIdx.x * TILE + (h * MT_ROWS + k) * NT + tid;
            if (i >= padded) continue;
            u32 code = 0xFFFFFFFFu;   // collision.py:137-142
            if (i < n) {
                if (packed) {
                    Vec4<T> pr = c[k];
                    pr.w = rad[k];
                    packed[i] = pr;
                }
                code = (expand_bits(quantize(c[k].x, mn.x, mx.x)) << 2) + (expand_bits(quantize(c[k].y, mn.y, mx.y)) << 1) +
                       expand_bits(quantize(c[k].z, mn.z, mx.z));
            }
            codes[i] = code;
            if (ids) ids[i] = i;
            atomicAdd(&s_hist[(code >> hist_shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (tid < 256) hist0[(uint64_t)tid * nblocks + blockIdx.x] = s_hist[tid];
}

template <typename T>
int launch_morton_tile(hipStream_t s, u32 tile, const void *coords, const void *radii, const void *partials, uint32_t parts,
                       uint32_t n, uint32_t padded, uint32_t *codes, uint32_t *ids, void *packed, uint32_t *zero_word,
                       uint32_t *hist0, uint32_t nblocks, int hist_shift, uint32_t *publish, const uint32_t *n_dev) {
    dim3 grid(nblocks);
    if (tile == 1024)
        k_morton_tile<T, 256, 1><<<grid, dim3(256), 0, s>>>((const Vec4<T> *)coords, (const T *)partials, parts, n, padded, codes, ids,
                                                             (const T *)radii, (Vec4<T> *)packed, zero_word, hist0, nblocks, hist_shift, publish, n_dev);
    else if (tile == 4096)
        k_morton_tile<T, 1024, 1><<<grid, dim3(1024), 0, s>>>((const Vec4<T> *)coords, (const T *)partials, parts, n, padded, codes, ids,
                                                               (const T *)radii, (Vec4<T> *)packed, zero_word, hist0, nblocks, hist_shift, publish, n_dev);
    else
        k_morton_tile<T, 1024, 2><<<grid, dim3(1024), 0, s>>>((const Vec4<T> *)coords, (const T *)partials, parts, n, padded, codes, ids,
                                                               (const T *)radii, (Vec4<T> *)packed, zero_word, hist0, nblocks, hist_shift, publish, n_dev);
    COL_LAUNCH_OK();
    return COL_OK;
}

}  // namespace

extern "C" {

int col_morton_tile(void *stream, const void *coords, const void *radii, const void *partials, uint32_t parts,
                    uint32_t n, uint32_t padded, int coord_bytes, uint32_t *codes, uint32_t *ids, void *packed,
                    uint32_t *zero_word, uint32_t *hist0, uint32_t tile, uint32_t nblocks, int hist_shift, uint32_t *publish,
                    const uint32_t *n_dev) {
    if (padded < n || parts == 0 || !hist0 || hist_shift < 0 || hist_shift > 24) return COL_EINVAL;
    if (tile != 1024 && tile != 4096 && tile != 8192) return COL_EINVAL;
    if (padded == 0) return COL_OK;
    if (packed && !radii) return COL_EINVAL;
    if (nblocks != (uint32_t)col_ceil_div(padded, tile)) return COL_EINVAL;
    if (coord_bytes == 4)
        return launch_morton_tile<float>(col_stream(stream), tile, coords, radii, partials, parts, n, padded, codes, ids, packed,
                                         zero_word, hist0, nblocks, hist_shift, publish, n_dev);
    if (coord_bytes == 8)
        return launch_morton_tile<double>(col_stream(stream), tile, coords, radii, partials, parts, n, padded, codes, ids, packed,
                                          zero_word, hist0, nblocks, hist_shift, publish, n_dev);
    return COL_EINVAL;
}

// col_morton plus two by-products for col_collide: packed (x, y, z, r) rows and a zeroed word.
int col_morton_ex(void *stream, const void *coords, const void *radii, const void *range, uint32_t n, uint32_t padded,
                  int coord_bytes, uint32_t *codes, uint32_t *ids, void *packed, uint32_t *zero_word, uint32_t *publish) {
    if (padded < n) return COL_EINVAL;
    if (padded == 0) return COL_OK;
    if (packed && !radii) return COL_EINVAL;
    dim3 grid((unsigned)col_ceil_div(padded, 256)), block(256);
    if (coord_bytes == 4)
        k_morton<float><<<grid, block, 0, col_stream(stream)>>>((const Vec4<float> *)coords, (const Vec4<float> *)range, n, padded,
                                                                codes, ids, (const float *)radii, (Vec4<float> *)packed, zero_word, publish);
    else if (coord_bytes == 8)
        k_morton<double><<<grid, block, 0, col_stream(stream)>>>((const Vec4<double> *)coords, (const Vec4<double> *)range, n,
                                                                 padded, codes, ids, (const double *)radii, (Vec4<double> *)packed,
                                                                 zero_word, publish);
    else
        return COL_EINVAL;
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_morton(void *stream, const void *coords, const void *range, uint32_t n, uint32_t padded, int coord_bytes,
               uint32_t *codes, uint32_t *ids) {
    return col_morton_ex(stream, coords, nullptr, range, n, padded, coord_bytes, codes, ids, nullptr, nullptr, nullptr);
}

}  // extern "C"

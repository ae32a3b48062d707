// 30-bit Morton codes + id ramp + padding.
// Replaces `range`, the 0xFFFFFFFF padding fill and `calculateCodes`
// (collision/collision.py:137-146,161-165; collision/collision.cl:8-40) with one launch.
//
// HBM-bound: one 16-byte (f32) / 32-byte (f64) row load, 4+4 bytes stored per sphere.
// Arithmetic must be IEEE with one rounding per operation (this file is compiled with
// -ffp-contract=off and correctly rounded division) so codes match the oracle bit for bit.
#include "col_common.h"

namespace {

__device__ __forceinline__ u32 expand_bits(u32 v) {   // collision.cl:14-20
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

template <typename T> struct alignas(4 * sizeof(T)) Vec4 { T x, y, z, w; };
template <> struct alignas(16) Vec4<double> { double x, y, z, w; };

template <typename T> __device__ __forceinline__ T tmax(T a, T b);
template <> __device__ __forceinline__ float tmax(float a, float b) { return fmaxf(a, b); }
template <> __device__ __forceinline__ double tmax(double a, double b) { return fmax(a, b); }
template <typename T> __device__ __forceinline__ T tmin(T a, T b);
template <> __device__ __forceinline__ float tmin(float a, float b) { return fminf(a, b); }
template <> __device__ __forceinline__ double tmin(double a, double b) { return fmin(a, b); }

// collision.cl:22-31: q = (uint) clamp(((p - min) / (max - min)) * 1023, 0, 1023); NaN -> 0.
template <typename T>
__device__ __forceinline__ u32 quantize(T p, T mn, T mx) {
    T t = (p - mn) / (mx - mn);
    t = t * (T)1023;
    t = tmin(tmax(t, (T)0), (T)1023);
    return (u32)t;
}

template <typename T>
__global__ __launch_bounds__(256) void k_morton(const Vec4<T> *__restrict__ coords,
                                                 const Vec4<T> *__restrict__ range, u32 n, u32 padded,
                                                 u32 *__restrict__ codes, u32 *__restrict__ ids,
                                                 const T *__restrict__ radii, Vec4<T> *__restrict__ packed,
                                                 u32 *__restrict__ zero_word) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && zero_word) *zero_word = 0;            // the pair counter (collision.py:151-154), no extra launch
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

}  // namespace

extern "C" {

// col_morton plus two by-products for col_collide: packed (x, y, z, r) rows and a zeroed word.
int col_morton_ex(void *stream, const void *coords, const void *radii, const void *range, uint32_t n, uint32_t padded,
                  int coord_bytes, uint32_t *codes, uint32_t *ids, void *packed, uint32_t *zero_word) {
    if (padded < n) return COL_EINVAL;
    if (padded == 0) return COL_OK;
    if (packed && !radii) return COL_EINVAL;
    dim3 grid((unsigned)col_ceil_div(padded, 256)), block(256);
    if (coord_bytes == 4)
        k_morton<float><<<grid, block, 0, col_stream(stream)>>>((const Vec4<float> *)coords, (const Vec4<float> *)range, n, padded,
                                                                codes, ids, (const float *)radii, (Vec4<float> *)packed, zero_word);
    else if (coord_bytes == 8)
        k_morton<double><<<grid, block, 0, col_stream(stream)>>>((const Vec4<double> *)coords, (const Vec4<double> *)range, n,
                                                                 padded, codes, ids, (const double *)radii, (Vec4<double> *)packed,
                                                                 zero_word);
    else
        return COL_EINVAL;
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_morton(void *stream, const void *coords, const void *range, uint32_t n, uint32_t padded, int coord_bytes,
               uint32_t *codes, uint32_t *ids) {
    return col_morton_ex(stream, coords, nullptr, range, n, padded, coord_bytes, codes, ids, nullptr, nullptr);
}

}  // extern "C"

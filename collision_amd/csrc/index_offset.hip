// Indexer.gather / Indexer.scatter (collision/index.py:23-55, collision/index.cl:1-13) and
// OffsetFinder.find_offsets (collision/offset.py:37-49, collision/offset.cl:3-12).
// Not called by Collider; SURVEY.md section 8(f) "next" rows.  Element-granular gathers are
// latency/sector bound, so the only shaping is: one element per lane, 4..32-byte moves.
#include "col_common.h"

namespace {

template <int B> struct Blob;
template <> struct Blob<1> { typedef uint8_t T; };
template <> struct Blob<2> { typedef uint16_t T; };
template <> struct Blob<4> { typedef uint32_t T; };
template <> struct Blob<8> { typedef uint2 T; };
template <> struct Blob<16> { typedef uint4 T; };
struct alignas(16) B32 { uint4 a, b; };
template <> struct Blob<32> { typedef B32 T; };

template <typename V, typename I, bool GATHER>
__global__ __launch_bounds__(256) void k_index(const V *__restrict__ in, const I *__restrict__ idx, V *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (GATHER) out[i] = in[idx[i]];
    else out[idx[i]] = in[i];
}

template <typename V, bool GATHER>
int by_index(hipStream_t s, const void *in, const void *idx, void *out, uint64_t n, int index_bytes) {
    dim3 grid((unsigned)col_ceil_div(n, 256)), block(256);
    if (index_bytes == 1) k_index<V, uint8_t, GATHER><<<grid, block, 0, s>>>((const V *)in, (const uint8_t *)idx, (V *)out, n);
    else if (index_bytes == 2) k_index<V, uint16_t, GATHER><<<grid, block, 0, s>>>((const V *)in, (const uint16_t *)idx, (V *)out, n);
    else if (index_bytes == 4) k_index<V, uint32_t, GATHER><<<grid, block, 0, s>>>((const V *)in, (const uint32_t *)idx, (V *)out, n);
    else if (index_bytes == 8) k_index<V, uint64_t, GATHER><<<grid, block, 0, s>>>((const V *)in, (const uint64_t *)idx, (V *)out, n);
    else return COL_EINVAL;
    COL_LAUNCH_OK();
    return COL_OK;
}

template <bool GATHER>
int by_value(void *stream, const void *in, const void *idx, void *out, uint64_t n, int val_bytes, int index_bytes) {
    if (n == 0) return COL_OK;
    hipStream_t s = col_stream(stream);
    switch (val_bytes) {
    case 1: return by_index<Blob<1>::T, GATHER>(s, in, idx, out, n, index_bytes);
    case 2: return by_index<Blob<2>::T, GATHER>(s, in, idx, out, n, index_bytes);
    case 4: return by_index<Blob<4>::T, GATHER>(s, in, idx, out, n, index_bytes);
    case 8: return by_index<Blob<8>::T, GATHER>(s, in, idx, out, n, index_bytes);
    case 16: return by_index<Blob<16>::T, GATHER>(s, in, idx, out, n, index_bytes);
    case 32: return by_index<Blob<32>::T, GATHER>(s, in, idx, out, n, index_bytes);
    default: return COL_EINVAL;
    }
}

// offset.cl:3-12: thread g looks at the sorted pair (values[g], values[g+1]) and writes
// offsets[v] = g+1 for every v in (a, b]; threads g <= values[0] write offsets[g] = 0.
template <typename V, typename O>
__global__ __launch_bounds__(256) void k_find_offsets(const V *__restrict__ values, O *__restrict__ offsets, uint64_t n_pairs) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_pairs) return;
    const uint64_t a = values[g], b = values[g + 1];      // (in 64 bits: no wrap at the value type's maximum)
    for (uint64_t v = a + 1; v <= b && v != 0; v++) offsets[v] = (O)(g + 1);
    if (g <= (uint64_t)values[0]) offsets[g] = 0;
}

}  // namespace

extern "C" {

int col_gather(void *stream, const void *in, const void *indices, void *out, uint64_t n, int val_bytes, int index_bytes) {
    return by_value<true>(stream, in, indices, out, n, val_bytes, index_bytes);
}
int col_scatter(void *stream, const void *in, const void *indices, void *out, uint64_t n, int val_bytes, int index_bytes) {
    return by_value<false>(stream, in, indices, out, n, val_bytes, index_bytes);
}

int col_find_offsets(void *stream, const void *values, uint64_t n_values, void *offsets, uint64_t n_offsets,
                     int value_bytes, int offset_bytes) {
    hipStream_t s = col_stream(stream);
    // offset.py:41-45: fill with n_values first
    int rc;
    if (offset_bytes == 1) { uint8_t v = (uint8_t)n_values; rc = col_fill(stream, offsets, &v, 1, n_offsets); }
    else if (offset_bytes == 2) { uint16_t v = (uint16_t)n_values; rc = col_fill(stream, offsets, &v, 2, n_offsets); }
    else if (offset_bytes == 4) { uint32_t v = (uint32_t)n_values; rc = col_fill(stream, offsets, &v, 4, n_offsets); }
    else if (offset_bytes == 8) { uint64_t v = n_values; rc = col_fill(stream, offsets, &v, 8, n_offsets); }
    else return COL_EINVAL;
    if (rc) return rc;
    if (value_bytes != 1 && value_bytes != 2 && value_bytes != 4 && value_bytes != 8) return COL_EINVAL;
    if (n_values < 2) return COL_OK;
    const uint64_t np = n_values - 1;
    dim3 grid((unsigned)col_ceil_div(np, 256)), block(256);
#define COL_OFFSETS(V, O) k_find_offsets<V, O><<<grid, block, 0, s>>>((const V *)values, (O *)offsets, np)
#define COL_OFFSETS_BY_O(V)                                  \
    switch (offset_bytes) {                                  \
    case 1: COL_OFFSETS(V, uint8_t); break;                  \
    case 2: COL_OFFSETS(V, uint16_t); break;                 \
    case 4: COL_OFFSETS(V, uint32_t); break;                 \
    default: COL_OFFSETS(V, uint64_t); break;                \
    }
    switch (value_bytes) {
    case 1: COL_OFFSETS_BY_O(uint8_t); break;
    case 2: COL_OFFSETS_BY_O(uint16_t); break;
    case 4: COL_OFFSETS_BY_O(uint32_t); break;
    default: COL_OFFSETS_BY_O(uint64_t); break;
    }
#undef COL_OFFSETS_BY_O
#undef COL_OFFSETS
    COL_LAUNCH_OK();
    return COL_OK;
}

}  // extern "C"

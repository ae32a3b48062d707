// Device pieces of the multi-GPU path (nothing in the reference to mirror:
// it is single-device).  One process per GPU; the collectives (AABB all-gathers, sphere
// repartition, halo exchange) are RCCL calls made by the host code in collision_amd/multi.py; the
// kernels here prepare and consume the exchanged buffers, for f32 and f64 coordinates (coord_bytes):
//
//   col_sample_rows          a rank's contribution to the splitter sample
//   col_fold_boxes_strided   gathered [min row, max row] boxes -> the global scene range
//   col_splitters_u32        world - 1 quantiles of the gathered samples' Morton codes
//   col_bucketize_u32        owner rank of every Morton code; col_digit_counts: spheres per owner
//   col_pack_records / col_unpack_records   transport records (x, y, z, r, global id)
//   col_region_box           a conservative box around everything a rank owns
//   col_select_overlap_multi the halo lists: owned spheres whose box overlaps a peer's region
//   col_pack_slots           the lists as fixed-size slots with a length header
//   col_traverse_ghost_slots received spheres as QUERIES against the local LBVH (never inserted);
//                            emits (ghost global id, local global id)
//   col_translate_pairs      local sphere indices -> global ids for the pairs found locally
//
// Rows are 4 scalars (x, y, z, r); a transport record is those 4 scalars followed by the 32-bit
// global id: 5 words for f32, 9 for f64.
#include "col_common.h"
#include <math.h>

namespace {

constexpr u32 END = 0xFFFFFFFFu;

template <typename T> struct MT;
template <> struct MT<float> { typedef float4 V4; typedef u32 Bits; static constexpr int RW = 5; };
template <> struct MT<double> { typedef double4 V4; typedef u64 Bits; static constexpr int RW = 9; };

template <typename T> __device__ __forceinline__ void rec_store(u32 *o, const typename MT<T>::V4 &c, u32 gid) {
    if constexpr (sizeof(T) == 4) {
        o[0] = __float_as_uint(c.x); o[1] = __float_as_uint(c.y); o[2] = __float_as_uint(c.z); o[3] = __float_as_uint(c.w);
    } else {                                  // (records of f64 rows are only 4-byte aligned: word-wise copies)
        const double v[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u64 b = (u64)__double_as_longlong(v[k]);
            o[2 * k] = (u32)b; o[2 * k + 1] = (u32)(b >> 32);
        }
    }
    o[MT<T>::RW - 1] = gid;
}
template <typename T> __device__ __forceinline__ typename MT<T>::V4 rec_load(const u32 *o, u32 *gid) {
    typename MT<T>::V4 c;
    if constexpr (sizeof(T) == 4) {
        c.x = __uint_as_float(o[0]); c.y = __uint_as_float(o[1]); c.z = __uint_as_float(o[2]); c.w = __uint_as_float(o[3]);
    } else {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = __longlong_as_double((long long)(((u64)o[2 * k + 1] << 32) | o[2 * k]));
        c.x = v[0]; c.y = v[1]; c.z = v[2]; c.w = v[3];
    }
    *gid = o[MT<T>::RW - 1];
    return c;
}

template <typename T>
__global__ __launch_bounds__(256) void k_pack_records(const typename MT<T>::V4 *__restrict__ rows, const u32 *__restrict__ gids,
                                                       const u32 *__restrict__ idx, u32 n, u32 *__restrict__ rec) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 s = idx ? idx[i] : i;
    rec_store<T>(rec + (u64)MT<T>::RW * i, rows[s], gids ? gids[s] : s);
}
template <typename T>
__global__ __launch_bounds__(256) void k_unpack_records(const u32 *__restrict__ rec, u32 n, typename MT<T>::V4 *__restrict__ rows,
                                                         u32 *__restrict__ gids, T *__restrict__ radii) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 gid;
    const typename MT<T>::V4 c = rec_load<T>(rec + (u64)MT<T>::RW * i, &gid);
    rows[i] = c;
    gids[i] = gid;
    if (radii) radii[i] = c.w;
}
template <typename T>
__global__ __launch_bounds__(256) void k_unpack_radii(const typename MT<T>::V4 *__restrict__ rows, u32 n, T *__restrict__ radii) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) radii[i] = rows[i].w;
}

// Halo selection against up to 8 peer boxes in one launch: list[k] collects the indices of the
// spheres overlapping the region box of peer q[k], counts[k] their number.  The boxes stay on the
// device (they come straight out of the AABB all-gather): no host round trip.
struct PeerList { int q[8]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void k_select_multi(const typename MT<T>::V4 *__restrict__ rows, u32 n,
                                                       const typename MT<T>::V4 *__restrict__ boxes, PeerList pl, u32 stride,
                                                       u32 *__restrict__ lists, u32 *__restrict__ counts) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    typename MT<T>::V4 c;
    c.x = c.y = c.z = c.w = (T)0;
    if (i < n) c = rows[i];
    const u32 lane = lane_id();
    for (int k = 0; k < pl.n; k++) {
        const typename MT<T>::V4 lo = boxes[2 * pl.q[k]], hi = boxes[2 * pl.q[k] + 1];      // wave-uniform
        const bool hit = i < n && c.x + c.w > lo.x && c.x - c.w < hi.x && c.y + c.w > lo.y && c.y - c.w < hi.y &&
                         c.z + c.w > lo.z && c.z - c.w < hi.z;               // strict, as collision.cl:164-166
        const u64 hits = __ballot(hit);
        if (!hits) continue;
        const int leader = (int)__builtin_ctzll(hits);
        u32 base = 0;
        if ((int)lane == leader) base = atomicAdd(&counts[k], (u32)__popcll(hits));
        base = __shfl(base, leader, COL_WAVE);
        if (hit) lists[(uint64_t)k * stride + base + mbcnt(hits)] = i;
    }
}

// Pack the selection lists into fixed SLOTS of 1 + slot records (blockIdx.y = list): a header record whose
// first word is the list's full length, then min(length, slot) records.  List sizes are read from the device
// (counts), so the launch needs no host knowledge of them; a fixed-size exchange carries the counts with
// the data, and the receiver sees from the header whether the slot overflowed.
template <typename T>
__global__ __launch_bounds__(256) void k_pack_slots(const typename MT<T>::V4 *__restrict__ rows, const u32 *__restrict__ gids,
                                                     const u32 *__restrict__ lists, u32 stride,
                                                     const u32 *__restrict__ counts, u32 *__restrict__ rec, u32 slot) {
    constexpr int RW = MT<T>::RW;
    const u32 k = blockIdx.y;
    const u32 len = counts[k];
    u32 *base = rec + (u64)RW * k * (slot + 1);
    if (blockIdx.x == 0 && threadIdx.x < (u32)RW) base[threadIdx.x] = threadIdx.x == 0 ? len : 0u;
    const u32 cnt = min(len, slot);
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) {
        const u32 s = lists[(uint64_t)k * stride + i];
        rec_store<T>(base + (u64)RW * (1 + i), rows[s], gids[s]);
    }
}

// destination rank of a Morton code: number of splitters <= code (splitters sorted, <= 255 of them)
__global__ __launch_bounds__(256) void k_bucketize(const u32 *__restrict__ codes, u32 n, const u32 *__restrict__ splitters,
                                                    u32 n_split, u32 *__restrict__ dest) {
    __shared__ u32 sp[256];
    if (threadIdx.x < n_split) sp[threadIdx.x] = splitters[threadIdx.x];
    __syncthreads();
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 c = codes[i];
    u32 lo = 0, hi = n_split;                // first index with sp[idx] > c
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (sp[mid] <= c) lo = mid + 1; else hi = mid;
    }
    dest[i] = lo;
}

// Ghost spheres as queries: a lane-per-ghost walk from the root over the node records (box + skip/down
// links, bvh.hip).  No position pruning: every local leaf is a candidate for a ghost.
// The ghosts are the transport records of a slotted exchange (k_pack_slots): blockIdx.y = slot, the slot's
// header gives its length; an overflowed slot is reported in flags[0] (largest length seen), flags[1]
// accumulates the number of ghosts.
constexpr int GW = 4, GCAP = 256;
template <typename T>
__global__ __launch_bounds__(GW * 64) void k_ghost(const typename MT<T>::V4 *__restrict__ rows, u32 n,
                                                   const u32 *__restrict__ local_gids, u32 *__restrict__ pairs,
                                                   u32 *__restrict__ counter, u32 capacity, const u32 *__restrict__ rec,
                                                   u32 slot, u32 *__restrict__ flags) {
    typedef typename MT<T>::V4 V4;
    typedef typename MT<T>::Bits Bits;
    __shared__ uint2 s_buf[GW][GCAP];
    const u32 lane = lane_id(), w = threadIdx.x / 64;
    const u32 leaf_start = n - 1;
    uint2 *buf = s_buf[w];
    u32 staged = 0;
    const u32 g = blockIdx.x * (GW * 64) + threadIdx.x;
    T lx = (T)INFINITY, ly = lx, lz = lx, hx = -lx, hy = -lx, hz = -lx;
    u32 gid = 0, idx = END;
    const u32 *base = rec + (u64)MT<T>::RW * blockIdx.y * (slot + 1);
    const u32 len = base[0];
    if (blockIdx.x == 0 && threadIdx.x == 0 && len) {
        atomicMax(&flags[0], len);
        atomicAdd(&flags[1], min(len, slot));
    }
    if (g < min(len, slot)) {
        const V4 c = rec_load<T>(base + (u64)MT<T>::RW * (1 + g), &gid);
        lx = c.x - c.w; ly = c.y - c.w; lz = c.z - c.w;                // same arithmetic as leafBounds, collision.cl:139-140
        hx = c.x + c.w; hy = c.y + c.w; hz = c.z + c.w;
        idx = 0;
    }
    while (__ballot(idx != END)) {
        bool hit = false;
        u32 down = 0;
        if (idx != END) {
            const V4 a = rows[2ull * idx], b = rows[2ull * idx + 1];
            const u32 skip = (u32) * reinterpret_cast<const Bits *>(&a.w);
            down = (u32) * reinterpret_cast<const Bits *>(&b.w);
            const bool overlap = hx > a.x && lx < b.x && hy > a.y && ly < b.y && hz > a.z && lz < b.z;
            const bool leaf = idx >= leaf_start;
            hit = overlap && leaf;
            idx = (overlap && !leaf) ? down : skip;
        }
        const u64 hits = __ballot(hit);
        if (hits) {
            const u32 add = (u32)__popcll(hits);
            if (staged + add > (u32)GCAP) {      // flush this wave's staging area with one atomic
                u32 b0 = 0;
                if (lane == 0) b0 = atomicAdd(counter, staged);
                b0 = (u32)__builtin_amdgcn_readfirstlane((int)b0);
                for (u32 i = lane; i < staged; i += 64)
                    if (b0 + i < capacity) *reinterpret_cast<uint2 *>(pairs + 2ull * (b0 + i)) = buf[i];
                staged = 0;
            }
            if (hit) buf[staged + mbcnt(hits)] = make_uint2(gid, local_gids ? local_gids[down] : down);
            staged += add;
        }
    }
    if (staged) {
        u32 b0 = 0;
        if (lane == 0) b0 = atomicAdd(counter, staged);
        b0 = (u32)__builtin_amdgcn_readfirstlane((int)b0);
        for (u32 i = lane; i < staged; i += 64)
            if (b0 + i < capacity) *reinterpret_cast<uint2 *>(pairs + 2ull * (b0 + i)) = buf[i];
    }
}

// pairs[first .. min(*count, capacity)) hold local indices: replace by gids[index]
__global__ __launch_bounds__(256) void k_translate(u32 *__restrict__ pairs, const u32 *__restrict__ count, u32 first,
                                                    u32 capacity, const u32 *__restrict__ gids) {
    const u32 total = min(*count, capacity);
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = 2ull * first + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < 2ull * total; i += stride)
        pairs[i] = gids[pairs[i]];
}

// ---- small protocol steps (one launch each instead of chains of tensor-library calls) ----

// fold `count` gathered [min row, max row] boxes (`stride` scalars apart) into one
template <typename T>
__global__ __launch_bounds__(64) void k_fold_boxes(const T *__restrict__ boxes, u32 count, u32 stride, T *__restrict__ out) {
    const u32 k = threadIdx.x;
    if (k >= 8) return;
    T acc = k < 4 ? (T)INFINITY : -(T)INFINITY;
    for (u32 i = 0; i < count; i++) {
        const T v = boxes[(u64)stride * i + k];
        acc = k < 4 ? (v < acc ? v : acc) : (v > acc ? v : acc);
    }
    out[k] = acc;
}

// `samples` evenly strided ROWS (x, y, z, r) of rows[0..n): what a rank contributes to the splitter
// sample (the rows are in id-hash order: a strided sample is a random sample); an empty rank
// contributes rows at +infinity (their codes clamp to the ceiling)
template <typename T>
__global__ __launch_bounds__(256) void k_sample_rows(const typename MT<T>::V4 *__restrict__ rows, u32 n, u32 samples,
                                                      typename MT<T>::V4 *__restrict__ out) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= samples) return;
    typename MT<T>::V4 v;
    v.x = v.y = v.z = (T)INFINITY; v.w = (T)0;
    if (n) v = rows[samples > 1 ? (u32)(((u64)i * (n - 1)) / (samples - 1)) : 0u];
    out[i] = v;
}

// [min row, max row] of (x, y, z, r) rows -> a box that contains every sphere: (min centre - max r,
// max centre + max r).  Conservative (exact for equal radii), which is all a halo selection needs.
template <typename T>
__global__ __launch_bounds__(64) void k_region_box(const T *__restrict__ minmax, T *__restrict__ out) {
    const u32 k = threadIdx.x;
    if (k >= 8) return;
    const T rmax = minmax[7];
    out[k] = (k & 3) == 3 ? (T)0 : (k < 4 ? minmax[k] - rmax : minmax[k] + rmax);
}

// world - 1 splitters = quantiles of the `count` gathered samples (count <= SPL_MAX): one block sorts
// them in LDS (bitonic) and picks every (count / world)-th
constexpr u32 SPL_MAX = 16384;
__global__ __launch_bounds__(1024) void k_splitters(const u32 *__restrict__ samples, u32 count, u32 world,
                                                     u32 *__restrict__ out) {
    __shared__ u32 s[SPL_MAX];
    u32 m = 1;
    while (m < count) m <<= 1;
    for (u32 i = threadIdx.x; i < m; i += 1024) s[i] = i < count ? samples[i] : 0xFFFFFFFFu;
    __syncthreads();
    for (u32 k = 2; k <= m; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 i = threadIdx.x; i < m; i += 1024) {
                const u32 l = i ^ j;
                if (l > i) {
                    const u32 a = s[i], b = s[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    const u32 step = count / world;
    for (u32 q = threadIdx.x + 1; q < world; q += 1024) out[q - 1] = s[q * step];
}

// spheres per destination from the SCANNED digit-major histogram of the owner pass: the run of digit q
// starts at scanned[q * nb]
__global__ __launch_bounds__(256) void k_digit_counts(const u32 *__restrict__ scanned, u32 nb, u32 world, u32 n,
                                                       u32 *__restrict__ out) {
    const u32 q = threadIdx.x;
    if (q >= world) return;
    const u32 lo = scanned[(u64)q * nb], hi = q + 1 < 256 ? scanned[(u64)(q + 1) * nb] : n;
    out[q] = hi - lo;
}

inline unsigned blocks_for(uint64_t n) { return (unsigned)col_ceil_div(n, 256); }

}  // namespace

#define COL_BY_COORD(call_f32, call_f64)         \
    do {                                         \
        if (coord_bytes == 4) { call_f32; }      \
        else if (coord_bytes == 8) { call_f64; } \
        else return COL_EINVAL;                  \
    } while (0)

extern "C" {

int col_fold_boxes_strided(void *stream, const void *boxes, uint32_t count, uint32_t stride_scalars, void *out8, int coord_bytes) {
    if (stride_scalars < 8) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    COL_BY_COORD((k_fold_boxes<float><<<dim3(1), dim3(64), 0, s>>>((const float *)boxes, count, stride_scalars, (float *)out8)),
                 (k_fold_boxes<double><<<dim3(1), dim3(64), 0, s>>>((const double *)boxes, count, stride_scalars, (double *)out8)));
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_sample_rows(void *stream, const void *rows, uint32_t n, uint32_t samples, void *out_rows, int coord_bytes) {
    if (samples == 0) return COL_OK;
    hipStream_t s = col_stream(stream);
    COL_BY_COORD((k_sample_rows<float><<<dim3(blocks_for(samples)), dim3(256), 0, s>>>((const float4 *)rows, n, samples, (float4 *)out_rows)),
                 (k_sample_rows<double><<<dim3(blocks_for(samples)), dim3(256), 0, s>>>((const double4 *)rows, n, samples, (double4 *)out_rows)));
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_region_box(void *stream, const void *minmax8, void *out8, int coord_bytes) {
    hipStream_t s = col_stream(stream);
    COL_BY_COORD((k_region_box<float><<<dim3(1), dim3(64), 0, s>>>((const float *)minmax8, (float *)out8)),
                 (k_region_box<double><<<dim3(1), dim3(64), 0, s>>>((const double *)minmax8, (double *)out8)));
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_splitters_u32(void *stream, const uint32_t *samples, uint32_t count, uint32_t world, uint32_t *out) {
    if (world < 2) return COL_OK;
    if (count == 0 || count > SPL_MAX || count < world) return COL_EINVAL;
    k_splitters<<<dim3(1), dim3(1024), 0, col_stream(stream)>>>(samples, count, world, out);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_digit_counts(void *stream, const uint32_t *scanned_hist, uint32_t nblocks, uint32_t world, uint32_t n,
                     uint32_t *out) {
    if (world == 0 || world > 256) return COL_EINVAL;
    k_digit_counts<<<dim3(1), dim3(256), 0, col_stream(stream)>>>(scanned_hist, nblocks, world, n, out);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_pack_records(void *stream, const void *rows, const uint32_t *gids, const uint32_t *idx, uint32_t n, void *rec,
                     int coord_bytes) {
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    if (n == 0) return COL_OK;
    hipStream_t s = col_stream(stream);
    COL_BY_COORD((k_pack_records<float><<<dim3(blocks_for(n)), dim3(256), 0, s>>>((const float4 *)rows, gids, idx, n, (u32 *)rec)),
                 (k_pack_records<double><<<dim3(blocks_for(n)), dim3(256), 0, s>>>((const double4 *)rows, gids, idx, n, (u32 *)rec)));
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_unpack_records(void *stream, const void *rec, uint32_t n, void *rows, uint32_t *gids, void *radii, int coord_bytes) {
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    if (n == 0) return COL_OK;
    hipStream_t s = col_stream(stream);
    COL_BY_COORD((k_unpack_records<float><<<dim3(blocks_for(n)), dim3(256), 0, s>>>((const u32 *)rec, n, (float4 *)rows, gids, (float *)radii)),
                 (k_unpack_records<double><<<dim3(blocks_for(n)), dim3(256), 0, s>>>((const u32 *)rec, n, (double4 *)rows, gids, (double *)radii)));
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_unpack_radii(void *stream, const void *rows, uint32_t n, void *radii, int coord_bytes) {
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    if (n == 0) return COL_OK;
    hipStream_t s = col_stream(stream);
    COL_BY_COORD((k_unpack_radii<float><<<dim3(blocks_for(n)), dim3(256), 0, s>>>((const float4 *)rows, n, (float *)radii)),
                 (k_unpack_radii<double><<<dim3(blocks_for(n)), dim3(256), 0, s>>>((const double4 *)rows, n, (double *)radii)));
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_select_overlap_multi(void *stream, const void *rows, uint32_t n, const void *boxes, const int *peers,
                             int n_peers, uint32_t stride, uint32_t *lists, uint32_t *counts, int coord_bytes) {
    if (n_peers < 0 || n_peers > 8) return COL_EINVAL;
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    if (n == 0 || n_peers == 0) return COL_OK;
    PeerList pl;
    pl.n = n_peers;
    for (int k = 0; k < 8; k++) pl.q[k] = k < n_peers ? peers[k] : -1;
    hipStream_t s = col_stream(stream);
    COL_BY_COORD((k_select_multi<float><<<dim3(blocks_for(n)), dim3(256), 0, s>>>((const float4 *)rows, n, (const float4 *)boxes, pl, stride, lists, counts)),
                 (k_select_multi<double><<<dim3(blocks_for(n)), dim3(256), 0, s>>>((const double4 *)rows, n, (const double4 *)boxes, pl, stride, lists, counts)));
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_pack_slots(void *stream, const void *rows, const uint32_t *gids, const uint32_t *lists, uint32_t stride,
                   const uint32_t *counts, int n_lists, uint32_t max_per_list, void *rec, uint32_t rec_capacity,
                   uint32_t slot_records, int coord_bytes) {
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    if (n_lists <= 0) return COL_OK;
    if (slot_records == 0 || (uint64_t)n_lists * (slot_records + 1) > rec_capacity) return COL_EINVAL;
    if (max_per_list == 0) max_per_list = 1;           // (the slot headers are still written)
    unsigned gx = blocks_for(max_per_list);
    if (gx > 1024) gx = 1024;
    hipStream_t s = col_stream(stream);
    COL_BY_COORD((k_pack_slots<float><<<dim3(gx, (unsigned)n_lists), dim3(256), 0, s>>>((const float4 *)rows, gids, lists, stride, counts, (u32 *)rec, slot_records)),
                 (k_pack_slots<double><<<dim3(gx, (unsigned)n_lists), dim3(256), 0, s>>>((const double4 *)rows, gids, lists, stride, counts, (u32 *)rec, slot_records)));
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_bucketize_u32(void *stream, const uint32_t *codes, uint32_t n, const uint32_t *splitters, uint32_t n_split,
                      uint32_t *dest) {
    if (n_split > 255) return COL_EINVAL;
    if (n == 0) return COL_OK;
    k_bucketize<<<dim3(blocks_for(n)), dim3(256), 0, col_stream(stream)>>>(codes, n, splitters, n_split, dest);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_traverse_ghost_slots(void *stream, const void *rec, uint32_t n_slots, uint32_t slot_records, const void *bounds,
                             uint32_t n, const uint32_t *local_gids, uint32_t *pairs, uint32_t *counter, uint32_t capacity,
                             uint32_t *flags, int coord_bytes) {
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    if (n_slots == 0 || slot_records == 0 || n == 0) return COL_OK;
    if ((capacity > 0 && !pairs) || !flags) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    dim3 grid((unsigned)col_ceil_div(slot_records, GW * 64), n_slots), block(GW * 64);
    COL_BY_COORD((k_ghost<float><<<grid, block, 0, s>>>((const float4 *)bounds, n, local_gids, pairs, counter, capacity, (const u32 *)rec, slot_records, flags)),
                 (k_ghost<double><<<grid, block, 0, s>>>((const double4 *)bounds, n, local_gids, pairs, counter, capacity, (const u32 *)rec, slot_records, flags)));
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_translate_pairs(void *stream, uint32_t *pairs, const uint32_t *count, uint32_t first, uint32_t capacity,
                        const uint32_t *gids) {
    if (capacity == 0) return COL_OK;
    k_translate<<<dim3(256), dim3(256), 0, col_stream(stream)>>>(pairs, count, first, capacity, gids);
    COL_LAUNCH_OK();
    return COL_OK;
}

}  // extern "C"

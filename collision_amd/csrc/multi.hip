// Device pieces of the multi-GPU path (nothing in the reference to mirror:
// it is single-device).  One process per GPU; the collectives (AABB all-gather, sphere
// repartition, halo exchange) are RCCL calls made by the host code in collision_amd/multi.py on
// the same stream; the kernels here prepare and consume the exchanged buffers:
//
//   col_pack_spheres      rows (x, y, z, r) + global ids, optionally gathered through an index list
//   col_unpack_radii      r lane of packed rows -> radii array
//   col_select_overlap    compact the indices of spheres whose box overlaps a peer's scene AABB
//                         (the halo: what a peer needs to see of this rank's spheres)
//   col_traverse_ghost    ghost spheres from other ranks as QUERIES against the local LBVH
//                         (they are never inserted); emits (ghost global id, local global id)
//   col_translate_pairs   local sphere indices -> global ids for the pairs found locally
#include "col_common.h"
#include <math.h>

namespace {

constexpr u32 END = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void k_pack(const float4 *__restrict__ coords, const float *__restrict__ radii,
                                               const u32 *__restrict__ gids, const u32 *__restrict__ idx, u32 n,
                                               float4 *__restrict__ rows, u32 *__restrict__ out_gids) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 s = idx ? idx[i] : i;
    float4 c = coords[s];
    if (radii) c.w = radii[s];
    rows[i] = c;
    if (out_gids) out_gids[i] = gids ? gids[s] : s;
}

// Transport record for the exchanges: 5 words (x, y, z, r, global id), so that one all-to-all
// moves everything a peer needs about a sphere.
__global__ __launch_bounds__(256) void k_pack5(const float4 *__restrict__ rows, const u32 *__restrict__ gids,
                                                const u32 *__restrict__ idx, u32 n, u32 *__restrict__ rec) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 s = idx ? idx[i] : i;
    const float4 c = rows[s];
    u32 *o = rec + 5ull * i;
    o[0] = __float_as_uint(c.x); o[1] = __float_as_uint(c.y); o[2] = __float_as_uint(c.z); o[3] = __float_as_uint(c.w);
    o[4] = gids ? gids[s] : s;
}
__global__ __launch_bounds__(256) void k_unpack5(const u32 *__restrict__ rec, u32 n, float4 *__restrict__ rows,
                                                  u32 *__restrict__ gids, float *__restrict__ radii) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 *o = rec + 5ull * i;
    const float4 c = make_float4(__uint_as_float(o[0]), __uint_as_float(o[1]), __uint_as_float(o[2]), __uint_as_float(o[3]));
    rows[i] = c;
    gids[i] = o[4];
    if (radii) radii[i] = c.w;
}

// Halo selection against up to 8 peer boxes in one launch: list[k] collects the indices of the
// spheres overlapping the region box of peer q[k], counts[k] their number.  The boxes stay on the
// device (they come straight out of the AABB all-gather): no host round trip.
struct PeerList { int q[8]; int n; };
__global__ __launch_bounds__(256) void k_select_multi(const float4 *__restrict__ rows, u32 n,
                                                       const float4 *__restrict__ boxes, PeerList pl, u32 stride,
                                                       u32 *__restrict__ lists, u32 *__restrict__ counts) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) c = rows[i];
    const u32 lane = lane_id();
    for (int k = 0; k < pl.n; k++) {
        const float4 lo = boxes[2 * pl.q[k]], hi = boxes[2 * pl.q[k] + 1];      // wave-uniform
        const bool hit = i < n && c.x + c.w > lo.x && c.x - c.w < hi.x && c.y + c.w > lo.y && c.y - c.w < hi.y &&
                         c.z + c.w > lo.z && c.z - c.w < hi.z;
        const u64 hits = __ballot(hit);
        if (!hits) continue;
        const int leader = (int)__builtin_ctzll(hits);
        u32 base = 0;
        if ((int)lane == leader) base = atomicAdd(&counts[k], (u32)__popcll(hits));
        base = __shfl(base, leader, COL_WAVE);
        if (hit) lists[(uint64_t)k * stride + base + mbcnt(hits)] = i;
    }
}

// Pack the n_lists selection lists into transport records; list sizes are read from the device
// (counts), so the launch needs no host knowledge of them.  blockIdx.y = list.
//   slot == 0: lists back to back.
//   slot > 0:  list k goes to a fixed SLOT of 1 + slot records: a header record whose first word is the
//              list's full length, then min(length, slot) records -- a fixed-size exchange carries the
//              counts with the data, and the receiver sees from the header whether the slot overflowed.
__global__ __launch_bounds__(256) void k_pack5_lists(const float4 *__restrict__ rows, const u32 *__restrict__ gids,
                                                      const u32 *__restrict__ lists, u32 stride,
                                                      const u32 *__restrict__ counts, u32 *__restrict__ rec, u32 rec_capacity,
                                                      u32 slot) {
    const u32 k = blockIdx.y;
    u32 cnt = counts[k];
    u32 off = 0;
    if (slot) {
        off = k * (slot + 1);
        if (blockIdx.x == 0 && threadIdx.x < 5) rec[5ull * off + threadIdx.x] = threadIdx.x == 0 ? cnt : 0u;
        off += 1;
        cnt = min(cnt, slot);
    } else {
        for (u32 j = 0; j < k; j++) off += counts[j];
    }
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) {
        if (off + i >= rec_capacity) return;            // overflow is reported by the host from the counts
        const u32 s = lists[(uint64_t)k * stride + i];
        const float4 c = rows[s];
        u32 *o = rec + 5ull * (off + i);
        o[0] = __float_as_uint(c.x); o[1] = __float_as_uint(c.y); o[2] = __float_as_uint(c.z); o[3] = __float_as_uint(c.w);
        o[4] = gids[s];
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

__global__ __launch_bounds__(256) void k_unpack_radii(const float4 *__restrict__ rows, u32 n, float *__restrict__ radii) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) radii[i] = rows[i].w;
}

// rows carry the radius in lane w.  aabb = (lo.xyz, -, hi.xyz, -).  Appends matching indices to
// out (one atomic per wave); order is not preserved.
__global__ __launch_bounds__(256) void k_select(const float4 *__restrict__ rows, u32 n, const float4 *__restrict__ aabb,
                                                 u32 *__restrict__ out, u32 *__restrict__ count) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    if (i < n) {
        const float4 c = rows[i], lo = aabb[0], hi = aabb[1];
        hit = c.x + c.w > lo.x && c.x - c.w < hi.x && c.y + c.w > lo.y && c.y - c.w < hi.y &&
              c.z + c.w > lo.z && c.z - c.w < hi.z;               // strict, as collision.cl:164-166
    }
    const u64 hits = __ballot(hit);
    if (!hits) return;
    const u32 lane = lane_id();
    const int leader = (int)__builtin_ctzll(hits);
    u32 base = 0;
    if ((int)lane == leader) base = atomicAdd(count, (u32)__popcll(hits));
    base = __shfl(base, leader, COL_WAVE);
    if (hit) out[base + mbcnt(hits)] = i;
}

// Lane-per-ghost walk from the root over the 32-byte records (box + skip/down links, bvh.hip).
// No position pruning: every local leaf is a candidate for a ghost.
// SLOTS: the ghosts are the transport records of a slotted exchange (k_pack5_lists, slot > 0): blockIdx.y =
// slot, the slot's header gives its length; an overflowed slot is reported in flags[0] (largest length
// seen), and flags[1] accumulates the number of ghosts.
constexpr int GW = 4, GCAP = 256;
template <bool SLOTS>
__global__ __launch_bounds__(GW * 64) void k_ghost(const float4 *__restrict__ ghosts, const u32 *__restrict__ ghost_gids,
                                                   u32 n_ghost, const float4 *__restrict__ rows, u32 n,
                                                   const u32 *__restrict__ local_gids, u32 *__restrict__ pairs,
                                                   u32 *__restrict__ counter, u32 capacity, const u32 *__restrict__ rec,
                                                   u32 slot, u32 *__restrict__ flags) {
    __shared__ uint2 s_buf[GW][GCAP];
    const u32 lane = lane_id(), w = threadIdx.x / 64;
    const u32 leaf_start = n - 1;
    uint2 *buf = s_buf[w];
    u32 staged = 0;
    const u32 g = blockIdx.x * (GW * 64) + threadIdx.x;
    float lx = INFINITY, ly = lx, lz = lx, hx = -lx, hy = -lx, hz = -lx;
    u32 gid = 0, idx = END;
    if (SLOTS) {
        const u32 *base = rec + 5ull * blockIdx.y * (slot + 1);
        const u32 len = base[0];
        if (blockIdx.x == 0 && threadIdx.x == 0 && len) {
            atomicMax(&flags[0], len);
            atomicAdd(&flags[1], min(len, slot));
        }
        if (g < min(len, slot)) {
            const u32 *o = base + 5ull * (1 + g);
            const float cx = __uint_as_float(o[0]), cy = __uint_as_float(o[1]), cz = __uint_as_float(o[2]), cr = __uint_as_float(o[3]);
            lx = cx - cr; ly = cy - cr; lz = cz - cr;                // same arithmetic as leafBounds, collision.cl:139-140
            hx = cx + cr; hy = cy + cr; hz = cz + cr;
            gid = o[4];
            idx = 0;
        }
    } else if (g < n_ghost) {
        const float4 c = ghosts[g];
        lx = c.x - c.w; ly = c.y - c.w; lz = c.z - c.w;          // same arithmetic as leafBounds, collision.cl:139-140
        hx = c.x + c.w; hy = c.y + c.w; hz = c.z + c.w;
        gid = ghost_gids[g];
        idx = 0;
    }
    while (__ballot(idx != END)) {
        bool hit = false;
        u32 down = 0;
        if (idx != END) {
            const float4 a = rows[2ull * idx], b = rows[2ull * idx + 1];
            const u32 skip = __float_as_uint(a.w);
            down = __float_as_uint(b.w);
            const bool overlap = hx > a.x && lx < b.x && hy > a.y && ly < b.y && hz > a.z && lz < b.z;
            const bool leaf = idx >= leaf_start;
            hit = overlap && leaf;
            idx = (overlap && !leaf) ? down : skip;
        }
        const u64 hits = __ballot(hit);
        if (hits) {
            const u32 add = (u32)__popcll(hits);
            if (staged + add > (u32)GCAP) {      // flush this wave's staging area with one atomic
                u32 base = 0;
                if (lane == 0) base = atomicAdd(counter, staged);
                base = (u32)__builtin_amdgcn_readfirstlane((int)base);
                for (u32 i = lane; i < staged; i += 64)
                    if (base + i < capacity) *reinterpret_cast<uint2 *>(pairs + 2ull * (base + i)) = buf[i];
                staged = 0;
            }
            if (hit) buf[staged + mbcnt(hits)] = make_uint2(gid, local_gids ? local_gids[down] : down);
            staged += add;
        }
    }
    if (staged) {
        u32 base = 0;
        if (lane == 0) base = atomicAdd(counter, staged);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        for (u32 i = lane; i < staged; i += 64)
            if (base + i < capacity) *reinterpret_cast<uint2 *>(pairs + 2ull * (base + i)) = buf[i];
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

// fold `count` gathered [min row, max row] boxes (`stride` floats apart) into one
__global__ __launch_bounds__(64) void k_fold_boxes(const float *__restrict__ boxes, u32 count, u32 stride, float *__restrict__ out) {
    const u32 k = threadIdx.x;
    if (k >= 8) return;
    float acc = k < 4 ? INFINITY : -INFINITY;
    for (u32 i = 0; i < count; i++) {
        const float v = boxes[(u64)stride * i + k];
        acc = k < 4 ? (v < acc ? v : acc) : (v > acc ? v : acc);
    }
    out[k] = acc;
}

// `samples` evenly strided elements of codes[0..n) (the codes are in id-hash order: a strided sample
// is a random sample); an empty rank contributes the code ceiling
__global__ __launch_bounds__(256) void k_sample(const u32 *__restrict__ codes, u32 n, u32 samples, u32 *__restrict__ out) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= samples) return;
    out[i] = n ? codes[samples > 1 ? (u32)(((u64)i * (n - 1)) / (samples - 1)) : 0u] : (1u << 30);
}

// `samples` evenly strided ROWS (x, y, z, r) of rows[0..n): what a rank contributes to the splitter
// sample; an empty rank contributes rows at +infinity (their codes clamp to the ceiling)
__global__ __launch_bounds__(256) void k_sample_rows(const float4 *__restrict__ rows, u32 n, u32 samples, float4 *__restrict__ out) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= samples) return;
    out[i] = n ? rows[samples > 1 ? (u32)(((u64)i * (n - 1)) / (samples - 1)) : 0u] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
}

// [min row, max row] of (x, y, z, r) rows -> a box that contains every sphere: (min centre - max r,
// max centre + max r).  Conservative (exact for equal radii), which is all a halo selection needs.
__global__ __launch_bounds__(64) void k_region_box(const float *__restrict__ minmax, float *__restrict__ out) {
    const u32 k = threadIdx.x;
    if (k >= 8) return;
    const float rmax = minmax[7];
    out[k] = (k & 3) == 3 ? 0.f : (k < 4 ? minmax[k] - rmax : minmax[k] + rmax);
}

// world - 1 splitters = quantiles of the `count` gathered samples (count <= SPL_MAX): one block sorts
// them in LDS (bitonic) and picks every (count / world)-th
constexpr u32 SPL_MAX = 8192;
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

// per-rank send counts of the halo exchange: out[peer k] = counts[k], 0 elsewhere
__global__ __launch_bounds__(256) void k_expand_counts(const u32 *__restrict__ counts, PeerList pl, u32 world,
                                                        u32 *__restrict__ out) {
    const u32 q = threadIdx.x;
    if (q >= world) return;
    u32 v = 0;
    for (int k = 0; k < pl.n; k++)
        if ((u32)pl.q[k] == q) v = counts[k];
    out[q] = v;
}

}  // namespace

extern "C" {

int col_fold_boxes(void *stream, const void *boxes, uint32_t count, void *out8) {
    return col_fold_boxes_strided(stream, boxes, count, 8, out8);
}

int col_fold_boxes_strided(void *stream, const void *boxes, uint32_t count, uint32_t stride_floats, void *out8) {
    if (stride_floats < 8) return COL_EINVAL;
    k_fold_boxes<<<dim3(1), dim3(64), 0, col_stream(stream)>>>((const float *)boxes, count, stride_floats, (float *)out8);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_sample_rows(void *stream, const void *rows, uint32_t n, uint32_t samples, void *out_rows) {
    if (samples == 0) return COL_OK;
    k_sample_rows<<<dim3((unsigned)col_ceil_div(samples, 256)), dim3(256), 0, col_stream(stream)>>>((const float4 *)rows, n, samples, (float4 *)out_rows);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_region_box(void *stream, const void *minmax8, void *out8) {
    k_region_box<<<dim3(1), dim3(64), 0, col_stream(stream)>>>((const float *)minmax8, (float *)out8);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_sample_u32(void *stream, const uint32_t *codes, uint32_t n, uint32_t samples, uint32_t *out) {
    if (samples == 0) return COL_OK;
    k_sample<<<dim3((unsigned)col_ceil_div(samples, 256)), dim3(256), 0, col_stream(stream)>>>(codes, n, samples, out);
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

int col_expand_counts(void *stream, const uint32_t *counts, const int *peers, int n_peers, uint32_t world,
                      uint32_t *out) {
    if (n_peers < 0 || n_peers > 8 || world == 0 || world > 256) return COL_EINVAL;
    PeerList pl;
    pl.n = n_peers;
    for (int k = 0; k < 8; k++) pl.q[k] = k < n_peers ? peers[k] : -1;
    k_expand_counts<<<dim3(1), dim3(256), 0, col_stream(stream)>>>(counts, pl, world, out);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_pack_spheres(void *stream, const void *coords, const void *radii, const uint32_t *gids, const uint32_t *idx,
                     uint32_t n, void *rows, uint32_t *out_gids) {
    if (n == 0) return COL_OK;
    k_pack<<<dim3((unsigned)col_ceil_div(n, 256)), dim3(256), 0, col_stream(stream)>>>(
        (const float4 *)coords, (const float *)radii, gids, idx, n, (float4 *)rows, out_gids);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_pack5(void *stream, const void *rows, const uint32_t *gids, const uint32_t *idx, uint32_t n, void *rec) {
    if (n == 0) return COL_OK;
    k_pack5<<<dim3((unsigned)col_ceil_div(n, 256)), dim3(256), 0, col_stream(stream)>>>((const float4 *)rows, gids, idx, n, (u32 *)rec);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_unpack5(void *stream, const void *rec, uint32_t n, void *rows, uint32_t *gids, void *radii) {
    if (n == 0) return COL_OK;
    k_unpack5<<<dim3((unsigned)col_ceil_div(n, 256)), dim3(256), 0, col_stream(stream)>>>((const u32 *)rec, n, (float4 *)rows, gids, (float *)radii);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_select_overlap_multi(void *stream, const void *rows, uint32_t n, const void *boxes, const int *peers,
                             int n_peers, uint32_t stride, uint32_t *lists, uint32_t *counts) {
    if (n_peers < 0 || n_peers > 8) return COL_EINVAL;
    if (n == 0 || n_peers == 0) return COL_OK;
    PeerList pl;
    pl.n = n_peers;
    for (int k = 0; k < n_peers; k++) pl.q[k] = peers[k];
    k_select_multi<<<dim3((unsigned)col_ceil_div(n, 256)), dim3(256), 0, col_stream(stream)>>>(
        (const float4 *)rows, n, (const float4 *)boxes, pl, stride, lists, counts);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_pack5_lists(void *stream, const void *rows, const uint32_t *gids, const uint32_t *lists, uint32_t stride,
                    const uint32_t *counts, int n_lists, uint32_t max_per_list, void *rec, uint32_t rec_capacity) {
    return col_pack5_slots(stream, rows, gids, lists, stride, counts, n_lists, max_per_list, rec, rec_capacity, 0);
}

int col_pack5_slots(void *stream, const void *rows, const uint32_t *gids, const uint32_t *lists, uint32_t stride,
                    const uint32_t *counts, int n_lists, uint32_t max_per_list, void *rec, uint32_t rec_capacity,
                    uint32_t slot_records) {
    if (n_lists <= 0) return COL_OK;
    if (slot_records && (uint64_t)n_lists * (slot_records + 1) > rec_capacity) return COL_EINVAL;
    if (max_per_list == 0) max_per_list = 1;           // (the slot headers are still written)
    unsigned gx = (unsigned)col_ceil_div(max_per_list, 256);
    if (gx > 1024) gx = 1024;
    k_pack5_lists<<<dim3(gx, (unsigned)n_lists), dim3(256), 0, col_stream(stream)>>>(
        (const float4 *)rows, gids, lists, stride, counts, (u32 *)rec, rec_capacity, slot_records);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_bucketize_u32(void *stream, const uint32_t *codes, uint32_t n, const uint32_t *splitters, uint32_t n_split,
                      uint32_t *dest) {
    if (n_split > 255) return COL_EINVAL;
    if (n == 0) return COL_OK;
    k_bucketize<<<dim3((unsigned)col_ceil_div(n, 256)), dim3(256), 0, col_stream(stream)>>>(codes, n, splitters, n_split, dest);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_unpack_radii(void *stream, const void *rows, uint32_t n, void *radii) {
    if (n == 0) return COL_OK;
    k_unpack_radii<<<dim3((unsigned)col_ceil_div(n, 256)), dim3(256), 0, col_stream(stream)>>>((const float4 *)rows, n, (float *)radii);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_select_overlap(void *stream, const void *rows, uint32_t n, const void *aabb, uint32_t *out, uint32_t *count) {
    if (n == 0) return COL_OK;
    k_select<<<dim3((unsigned)col_ceil_div(n, 256)), dim3(256), 0, col_stream(stream)>>>((const float4 *)rows, n, (const float4 *)aabb, out, count);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_traverse_ghost(void *stream, const void *ghost_rows, const uint32_t *ghost_gids, uint32_t n_ghost,
                       const void *bounds, uint32_t n, const uint32_t *local_gids, uint32_t *pairs,
                       uint32_t *counter, uint32_t capacity) {
    if (n_ghost == 0 || n == 0) return COL_OK;
    if (capacity > 0 && !pairs) return COL_EINVAL;
    k_ghost<false><<<dim3((unsigned)col_ceil_div(n_ghost, GW * 64)), dim3(GW * 64), 0, col_stream(stream)>>>(
        (const float4 *)ghost_rows, ghost_gids, n_ghost, (const float4 *)bounds, n, local_gids, pairs, counter, capacity,
        nullptr, 0, nullptr);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_traverse_ghost_slots(void *stream, const void *rec, uint32_t n_slots, uint32_t slot_records, const void *bounds,
                             uint32_t n, const uint32_t *local_gids, uint32_t *pairs, uint32_t *counter, uint32_t capacity,
                             uint32_t *flags) {
    if (n_slots == 0 || slot_records == 0 || n == 0) return COL_OK;
    if ((capacity > 0 && !pairs) || !flags) return COL_EINVAL;
    k_ghost<true><<<dim3((unsigned)col_ceil_div(slot_records, GW * 64), n_slots), dim3(GW * 64), 0, col_stream(stream)>>>(
        nullptr, nullptr, 0, (const float4 *)bounds, n, local_gids, pairs, counter, capacity, (const u32 *)rec, slot_records, flags);
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

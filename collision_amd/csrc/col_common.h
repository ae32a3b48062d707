// Shared device/host helpers for libcollision_hip.so (gfx950 only: 64-lane waves).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/collision_hip.h"
#include "../../include/collision_hip_debug.h"

#define COL_WAVE 64

#define COL_HIP(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return (int)e_;          \
    } while (0)

// launch check: hipGetLastError after a <<<>>> launch
#define COL_LAUNCH_OK()                                \
    do {                                               \
        hipError_t e_ = hipGetLastError();             \
        if (e_ != hipSuccess) return (int)e_;          \
    } while (0)

// internal entry points shared between translation units (not part of the public header)
extern "C" int col_morton_ex(void *stream, const void *coords, const void *radii, const void *range, uint32_t n,
                             uint32_t padded, int coord_bytes, uint32_t *codes, uint32_t *ids, void *packed,
                             uint32_t *zero_word, uint32_t *publish);
extern "C" int col_lbvh_ex(void *stream, const uint32_t *codes, const uint32_t *ids, const void *coords, const void *radii,
                           const void *packed, col_node *nodes, void *bounds, void *scratch, uint32_t n, int coord_bytes,
                           uint32_t *zero8,       // zero8: words the last kernel clears (the traversal's packet counters / chunk header), or NULL
                           const uint32_t *n_dev,    // device-side count (see col_morton_tile), or NULL
                           uint32_t *walk_order,     // the traversal's cost / order arrays (WALK ORDER below), or NULL
                           int order_mode,           // 1: a dense scene -- longest walks first; 0: only deal the long walks of a small launch
                           uint32_t nzero = 8,       // how many words at zero8 (8: the packet counters; 16: the chunked walk's whole header)
                           uint32_t *report_word = nullptr,   // col_radix_bucket_report's word: the report rides in k_chunk's launch (one
                           uint32_t report_n = 0);            // extra workgroup), report_n = length of `codes` with its pads
extern "C" int col_lbvh_order_forced(void);      // col_debug_lbvh bit 12: the tests' switch for order_mode 1 at every size
extern "C" int col_radix_sort_msd_dev(void *stream, const uint32_t *keys, uint32_t *keys_out, const uint32_t *vals, uint32_t *vals_out,
                                      uint64_t n, void *scratch, uint32_t *oversize, const uint32_t *n_real_dev);   // radix.hip

// col_collide's fused front end for inputs sorted with the 1024-pair tile (fewer launches, same bits):
//  * col_minmax4_stage1: stage 1 of col_reduce(MINMAX, width 4) only; `parts` partial results of
//    [min row, max row] are left in `partials` (at most COL_MINMAX_PARTS of them);
//  * col_morton_tile: col_morton_ex that folds those partials itself (no stage-2 launch) and also
//    writes a histogram for the radix sort (digit = bits hist_shift..hist_shift+7: 0 for the LSD sort's
//    pass 0, 22 for the MSD sort's bucket digit), `tile` = 1024 or 4096 codes per block (the sort's
//    tile, col_radix_tile), digit-major hist[d * nblocks + b] like k_hist;
//  * col_radix_sort_ex(have_hist0 = 1): col_radix_sort that finds the pass-0 histogram already at the
//    start of `scratch`.
#define COL_MINMAX_PARTS 256
// col_radix_sort_msd: one workgroup finishes one of 256 top-digit buckets in LDS -- 8192 pairs per bucket for
// inputs up to COL_MSD_SMALL_N codes, 16384 up to COL_MSD_MAX_N (uniform scenes: n / 256 per bucket +- 4 sigma)
#define COL_REGION_BOXES 8      /* boxes per rank region on the multi-GPU path (multi.hip: k_region) */
#define COL_MSD_SMALL_N 1900000u
#define COL_MSD_MAX_N 4050000u
extern "C" int col_minmax4_stage1(void *stream, const void *rows, uint64_t n, int coord_bytes, void *partials, uint32_t *parts);
extern "C" int col_minmax4_stage1_dev(void *stream, const void *rows, const uint32_t *n_dev, uint32_t n_max, int coord_bytes,
                                      void *partials, uint32_t *parts);
// DEVICE-SIDE COUNT (round 4, the multi-GPU step): `n_dev` (may be NULL) points at a device word that holds the real number
// of spheres; `n` is then a host-known upper bound that sizes grids and scratch, and every kernel works on min(n, *n_dev).
extern "C" int col_morton_tile(void *stream, const void *coords, const void *radii, const void *partials, uint32_t parts,
                               uint32_t n, uint32_t padded, int coord_bytes, uint32_t *codes, uint32_t *ids, void *packed,
                               uint32_t *zero_word, uint32_t *hist0, uint32_t tile, uint32_t nblocks, int hist_shift,
                               uint32_t *publish, const uint32_t *n_dev);
extern "C" int col_radix_sort_ex(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                                 uint64_t n, int key_bytes, int val_bytes, void *scratch, int copy_back, int have_hist0);

extern "C" int col_traverse_ghost_packets(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const void *bounds,
                                          uint32_t n, int coord_bytes, const uint32_t *rec, const uint32_t *order,
                                          const uint32_t *count, uint32_t max_ghosts, const uint32_t *local_gids,
                                          const uint32_t *n_dev);   // bvh.hip (n_dev: device-side count of the local tree, or NULL)
extern "C" int col_radix_sort_low_passes(void *stream, const uint32_t *keys, uint32_t *keys_out, const uint32_t *vals,
                                         uint32_t *vals_out, uint64_t n, void *scratch, int passes);      // radix.hip
extern "C" int col_radix_bucket_report(void *stream, const uint32_t *sorted_codes, uint64_t n, uint32_t *word);   // see radix.hip
extern "C" int col_radix_tile_override_active(void);      // diagnostics: col_debug_radix_tile() is in force

static inline hipStream_t col_stream(void *s) { return (hipStream_t)s; }

static inline uint64_t col_ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
typedef unsigned long long u64;
typedef unsigned int u32;
// The largest top-digit bucket (code bits 22..29, the MSD plan's bucket digit) of `n` sorted 30-bit codes (pads 0xFFFFFFFF
// behind them) -> *word, bit 31 set (col_radix_bucket_report, radix.hip).  One 256-thread block; s_start: 257 LDS words.
// A device function because the whole path lets it ride in k_chunk's launch (lbvh.hip) instead of a launch of its own.
__device__ __forceinline__ void col_bucket_report_block(const u32 *__restrict__ sorted, u32 n, u32 *word, u32 *s_start) {
    const u32 d = threadIdx.x;
    const u32 want = d << 22;
    u32 lo = 0, hi = n;                        // first position whose code is >= want
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (sorted[mid] < want) lo = mid + 1; else hi = mid;
    }
    s_start[d] = lo;
    if (d == 0) {                              // end of bucket 255: the first pad (0xFFFFFFFF) or n
        u32 l2 = 0, h2 = n;
        while (l2 < h2) {
            const u32 mid = l2 + ((h2 - l2) >> 1);
            if (sorted[mid] < (1u << 30)) l2 = mid + 1; else h2 = mid;
        }
        s_start[256] = l2;
    }
    __syncthreads();
    u32 cnt = s_start[d + 1] - s_start[d];
#pragma unroll
    for (int o = COL_WAVE / 2; o > 0; o >>= 1) cnt = max(cnt, (u32)__shfl_xor((int)cnt, o, COL_WAVE));
    __syncthreads();
    if ((d & 63u) == 0) s_start[d >> 6] = cnt;
    __syncthreads();
    if (d == 0) {
        const u32 m = max(max(s_start[0], s_start[1]), max(s_start[2], s_start[3]));
        __hip_atomic_store(word, 0x80000000u | min(m, 0x7FFFFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ __forceinline__ u32 lane_id() { return __lane_id(); }

// the real count behind a host-known bound (see col_morton_tile): a wave-uniform (scalar) load
__device__ __forceinline__ u32 count_of(u32 n, const u32 *__restrict__ n_dev) { return n_dev ? min(n, *n_dev) : n; }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ u32 mbcnt(u64 mask) {
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// inclusive wave scan (add) via shuffles; wave = 64 lanes
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
    const u32 lane = lane_id();
#pragma unroll
    for (int o = 1; o < COL_WAVE; o <<= 1) {
        u32 t = __shfl_up(v, o, COL_WAVE);
        if (lane >= (u32)o) v += t;
    }
    return v;
}

__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
    for (int o = COL_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, COL_WAVE);
    return v;
}

// Exclusive scan of one value per thread over a block of NT threads.
// `warp_sums` is NT/64 words of LDS.  Returns the exclusive prefix; *total gets the block sum.
template <int NT>
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *warp_sums, u32 *total) {
    constexpr int NW = NT / COL_WAVE;
    const u32 lane = lane_id(), w = threadIdx.x / COL_WAVE;
    u32 incl = wave_incl_scan(v);
    if (lane == COL_WAVE - 1) warp_sums[w] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        u32 s = warp_sums[i];
        if ((u32)i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}
// ---- traversal links (lane w of every node's Bound record, bvh.hip): `skip`, and `down` = left child (internal
// node) or sphere id (leaf).  LEAF BLOCKS: lbvh.hip lets an internal node that covers at most COL_LEAF_BLOCK
// consecutive sorted leaves [lo, hi] AND is dense (its box is at most a few leaf boxes wide on every axis, so a query
// that meets the node meets a good part of its leaves) carry COL_LINK_MARK | lo << 4 | (hi - lo) in `down` instead:
// a walk that hits it tests its leaves -- consecutive 32-byte records from leaf_start + lo -- straight away instead
// of descending (two dependent record fetches per leaf).  Only written while lo fits 27 bits
// (n <= COL_LEAF_BLOCK_MAX_N); larger trees carry plain child links and walkers must not interpret bit 31.
#define COL_LEAF_BLOCK 16u
#define COL_LINK_MARK 0x80000000u
#define COL_LEAF_BLOCK_MAX_N (1u << 27)
__device__ __forceinline__ u32 block_link(u32 lo, u32 hi) { return COL_LINK_MARK | (lo << 4) | (hi - lo); }
// where a walk goes on from a hit internal node with link `down`: its left child, or (leaf block) its first leaf --
// the skip links of the leaves then lead through the whole block (nested blocks are entered the same way)
__device__ __forceinline__ u32 descend_link(u32 down, u32 leaf_start, bool marks) {
    return (marks && (down & COL_LINK_MARK)) ? leaf_start + ((down >> 4) & 0x7FFFFFFu) : down;
}
// ---- WALK ORDER (round 4): the traversal's dynamic packet order hands out an XCD's walks LONGEST FIRST, by what each packet's
// walk took in the PREVIOUS call on the same scratch (any order is correct; a scene that changes little from call to call -- a
// simulation, a benchmark loop -- gets the makespan of a longest-first list schedule instead of a tail of late long walks).
// walk_order = cost[npk] then perm[npk] (u32 each, npk = packets of the BOUND n) then eight words "perm in use on XCD x": k_traverse
// writes a packet's walk time into cost[]; the tree build's last launch (k_cross: 8 x COL_ORDER_SLICES extra workgroups) turns the
// previous costs into perm[] per XCD range, and k_traverse maps walk unit u to packet perm[u] & 0x7FFFFFFF (bit 31: a long walk,
// run at raised wave priority).  Two modes (lbvh.hip): a DENSE scene (chunked pair allocation: its walks differ by 5 x) gets eight
// classes by cost / mean, longest first; any other scene of up to COL_DEAL_MAX_N spheres (a launch of two or three rounds) only
// has its few long walks dealt over the first round's batches, everything else in natural order.  Larger sparse scenes: nothing
// (walk_order = NULL -- no times recorded, no order: both cost more there than they gain).
#define COL_DEAL_MAX_N 1310720u     // 8 XCDs x 2560 packets x 64
#define COL_TRAV_WAVES 16u      // packets per batch of the dynamic order = waves per k_traverse workgroup (bvh.hip TW)
#define COL_TRAV_FIRST_BATCHES 64      // workgroups of the traversal's grid per XCD (512 / 8): the batches that start at once
__device__ __forceinline__ void xcd_packet_range(u32 npackets, u32 x, u32 &p_lo, u32 &p_end) {
    const u32 ng = (npackets + COL_TRAV_WAVES - 1) / COL_TRAV_WAVES;
    p_lo = (u32)(((u64)ng * x) >> 3) * COL_TRAV_WAVES;
    p_end = min((u32)(((u64)ng * (x + 1)) >> 3) * COL_TRAV_WAVES, npackets);
}
// ---- 30-bit Morton codes (collision.cl:14-31); shared by morton.hip and multi.hip ----
__device__ __forceinline__ u32 expand_bits(u32 v) {   // collision.cl:14-20
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
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
__device__ __forceinline__ u32 morton30(T x, T y, T z, T mnx, T mny, T mnz, T mxx, T mxy, T mxz) {
    return (expand_bits(quantize(x, mnx, mxx)) << 2) + (expand_bits(quantize(y, mny, mxy)) << 1) +
           expand_bits(quantize(z, mnz, mxz));
}
#endif


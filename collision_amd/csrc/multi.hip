// Device pieces of the multi-GPU path (nothing in the reference to mirror:
// it is single-device).  One process per GPU; the collectives (AABB all-gathers, sphere
// repartition, halo exchange) are RCCL calls made by the host code in collision_amd/multi.py; the
// kernels here prepare and consume the exchanged buffers, for f32 and f64 coordinates (coord_bytes):
//
//   col_partition_sample     a rank's contribution to the splitter sample + its centre range (ONE launch)
//   col_partition_plan       gathered samples -> global range, splitters; owner of every sphere; the
//                            per-tile owner histogram, scanned; spheres per owner   (3 launches)
//   col_partition_group      stable grouping by owner (one radix scatter) and packing: the spheres this
//                            rank keeps go straight into its owned arrays, the others into one fixed-size
//                            slot of transport records (x, y, z, r, global id) per peer   (2 launches)
//   col_partition_unpack     received slots -> owned arrays; publishes the owned count to the host
//   col_region_boxes         a rank's region: conservative boxes around what it owns, one per scene octant
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
__global__ __launch_bounds__(256) void k_unpack_radii(const typename MT<T>::V4 *__restrict__ rows, u32 n, T *__restrict__ radii) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) radii[i] = rows[i].w;
}

// Halo selection for up to 8 peers in one launch: list[k] collects the indices of the spheres overlapping the
// REGION of peer q[k], counts[k] their number.  A region is COL_REGION_BOXES boxes (k_region: one per octant of
// the scene, empty ones inverted); they stay on the device (straight out of the AABB all-gather): no host
// round trip.
struct PeerList { int q[8]; int n; };
// One global counter per list cannot take an atomic per wave: a single address retires ~88 atomics/us on this part,
// and with most waves holding a hit or two the first version of this kernel spent 0.3-0.4 ms there at 2 M owned
// spheres (1.1-1.5 ms with the hash partition, where every sphere is selected).  So a block takes SEL_ROWS rows per
// thread, remembers the outcome of its box tests in registers (a bit per row and peer), and reserves its share of
// each list with ONE atomic per block and peer after a block scan of the hit counts (at most 512 blocks, each with
// up to 32 rows per thread: ~2 k atomics on four addresses at 2 M rows; 4 rows per thread -- 7.8 k -- still took 55 us).
constexpr int SEL_ROWS = 32;      // at most: rows per thread, one mask bit each
template <typename T>
__global__ __launch_bounds__(256) void k_select_multi(const typename MT<T>::V4 *__restrict__ rows, u32 n,
                                                       const typename MT<T>::V4 *__restrict__ boxes, PeerList pl, u32 stride,
                                                       u32 *__restrict__ lists, u32 *__restrict__ counts, u32 per_thread,
                                                       const u32 *__restrict__ n_dev) {
    typedef typename MT<T>::V4 V4;
    n = count_of(n, n_dev);                  // device-side count (col_common.h)
    __shared__ u32 s_warp[4];
    __shared__ u32 s_base;
    const u32 tid = threadIdx.x;
    const u32 base = blockIdx.x * (256u * per_thread);      // this block's rows: [base, base + 256 * per_thread)
    u32 mask[8];
#pragma unroll
    for (int k = 0; k < 8; k++) mask[k] = 0;
    // the rows are only looked at, never kept: what a thread remembers is one bit per row and peer, and what goes
    // into a list is the row's index
    for (u32 j = 0; j < per_thread; j++) {
        const u32 i = base + j * 256u + tid;
        if (i >= n) break;
        const V4 c = rows[i];
        const T lx = c.x - c.w, ly = c.y - c.w, lz = c.z - c.w, hx = c.x + c.w, hy = c.y + c.w, hz = c.z + c.w;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k >= pl.n) break;
            const V4 *region = boxes + 2 * COL_REGION_BOXES * pl.q[k];      // wave-uniform
            bool hit = false;
#pragma unroll
            for (int o = 0; o < COL_REGION_BOXES; o++) {
                const V4 lo = region[2 * o], hi = region[2 * o + 1];
                hit |= hx > lo.x && lx < hi.x && hy > lo.y && ly < hi.y && hz > lo.z && lz < hi.z;   // strict, as collision.cl:164-166
            }
            mask[k] |= (u32)hit << j;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= pl.n) break;
        u32 total;
        const u32 before = block_excl_scan<256>((u32)__popc(mask[k]), s_warp, &total);
        if (total == 0) continue;                                        // (block-uniform)
        if (tid == 0) s_base = atomicAdd(&counts[k], total);             // ONE reservation per block and peer
        __syncthreads();
        u32 pos = s_base + before;
        __syncthreads();                                                 // (s_base is reused by the next peer)
        u32 m = mask[k];
        while (m) {
            const u32 j = (u32)__builtin_ctz(m);
            m &= m - 1;
            lists[(uint64_t)k * stride + pos++] = base + j * 256u + tid;
        }
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
                                                   u32 slot, u32 *__restrict__ flags, const u32 *__restrict__ n_dev) {
    typedef typename MT<T>::V4 V4;
    typedef typename MT<T>::Bits Bits;
    n = count_of(n, n_dev);                  // device-side count (col_common.h): the size of the local tree
    __shared__ uint2 s_buf[GW][GCAP];
    const u32 lane = lane_id(), w = threadIdx.x / 64;
    const u32 leaf_start = n - 1;
    const bool marks = n <= COL_LEAF_BLOCK_MAX_N;
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
    if (g < min(len, slot) && n > 0) {       // (no local tree: the ghosts are counted, nothing is walked)
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
            idx = (overlap && !leaf) ? descend_link(down, leaf_start, marks) : skip;      // (a leaf block: on through its leaves)
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

// Ghost queries as PACKETS (the default; k_ghost above is the lane-per-query walk kept for A/B and as the fallback
// without scratch): k_ghost_codes gives every received record a coarse Morton key (the top 15 bits of its 30-bit code
// under the local tree's root box -- cells of ~1/32 of the scene's extent per axis, ~60 leaves each at 2 M spheres) and
// its record number; two 8-bit sort passes order the record numbers by key; bvh.hip's packet walk then takes 64
// neighbouring ghosts per wave from the root (wave-uniform record per step, leaf blocks) -- the walk the local
// queries use, which DESIGN 4.6 measured 1.3-1.8x faster than the lane-per-query one even before leaf blocks.
// Entries beyond a slot's length get key 0xFFFF (sorted last; real keys are below 0x8000) and are never read:
// the walk runs over flags[1] ghosts, the count this kernel accumulates.
template <typename T>
__global__ __launch_bounds__(256) void k_ghost_codes(const u32 *__restrict__ rec, u32 slot, const typename MT<T>::V4 *__restrict__ rows,
                                                      u32 *__restrict__ keys, u32 *__restrict__ vals, u32 *__restrict__ flags) {
    typedef typename MT<T>::V4 V4;
    const u32 s = blockIdx.y, g = blockIdx.x * 256 + threadIdx.x;
    const u32 *base = rec + (u64)MT<T>::RW * s * (slot + 1);
    const u32 len = base[0], cnt = min(len, slot);
    if (blockIdx.x == 0 && threadIdx.x == 0 && len) {
        atomicMax(&flags[0], len);
        atomicAdd(&flags[1], cnt);
    }
    if (g >= slot) return;
    const u64 e = (u64)s * slot + g;
    u32 key = 0xFFFFu, val = 0xFFFFFFFFu;
    if (g < cnt) {
        const V4 lo = rows[0], hi = rows[1];                  // the root's box (node 0)
        u32 gid;
        const V4 c = rec_load<T>(base + (u64)MT<T>::RW * (1 + g), &gid);
        key = morton30<T>(c.x, c.y, c.z, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) >> 15;
        val = s * (slot + 1) + 1 + g;                         // record number in `rec`
    }
    keys[e] = key;
    vals[e] = val;
}

// pairs[first .. min(*count, capacity)) hold local indices: replace by gids[index]
__global__ __launch_bounds__(256) void k_translate(u32 *__restrict__ pairs, const u32 *__restrict__ count, u32 first,
                                                    u32 capacity, const u32 *__restrict__ gids) {
    const u32 total = min(*count, capacity);
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = 2ull * first + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < 2ull * total; i += stride)
        pairs[i] = gids[pairs[i]];
}

// ---- the repartition (Morton ranges of equal population), a few launches between the collectives ----

// min / max over the wave of 8 scalars (4 minima, 4 maxima)
template <typename T> __device__ __forceinline__ void wave_fold8(T (&v)[8]) {
#pragma unroll
    for (int o = COL_WAVE / 2; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const T t = __shfl_xor(v[k], o, COL_WAVE);
            v[k] = k < 4 ? (t < v[k] ? t : v[k]) : (t > v[k] ? t : v[k]);
        }
}

// [min row, max row] of rows[0..n) in two launches (a single launch that lets the last block fold the partials
// needs agent-scope fences, and on this part those write back and invalidate the whole L2 under the blocks that
// are still streaming: measured 22 us against 8 + 3 for the pair).  k_range_part: one partial per block.
constexpr int SR_BLOCKS = 256, RU = 4;
constexpr int RANGE_NT = 1024, REGION_NT = 512;      // threads per block of the two partial kernels (occupancy: few, long blocks)
template <typename T>
__global__ __launch_bounds__(RANGE_NT) void k_range_part(const typename MT<T>::V4 *__restrict__ rows, u32 n, T *__restrict__ partials) {
    typedef typename MT<T>::V4 V4;
    __shared__ T s_part[RANGE_NT / 64][8];
    const u32 tid = threadIdx.x, lane = lane_id(), w = tid / 64;
    T v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = k < 4 ? (T)INFINITY : -(T)INFINITY;
    // RU rows in flight per thread: the loop is bound by the latency of its loads, not by their bytes
    const u32 stride = gridDim.x * RANGE_NT;
    for (u32 i0 = blockIdx.x * RANGE_NT + tid; i0 < n; i0 += RU * stride) {
        V4 c[RU];
#pragma unroll
        for (int u = 0; u < RU; u++) c[u] = rows[min(i0 + u * stride, n - 1)];      // (a repeated row changes no min / max)
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const T e[4] = {c[u].x, c[u].y, c[u].z, c[u].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = e[k] < v[k] ? e[k] : v[k];
                v[4 + k] = e[k] > v[4 + k] ? e[k] : v[4 + k];
            }
        }
    }
    wave_fold8(v);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 8; k++) s_part[w][k] = v[k];
    __syncthreads();
    if (tid < 8) {
        T a = s_part[0][tid];
        for (int i = 1; i < RANGE_NT / 64; i++) { const T t = s_part[i][tid]; a = tid < 4 ? (t < a ? t : a) : (t > a ? t : a); }
        partials[(u64)blockIdx.x * 8 + tid] = a;
    }
}
// k_range_fold: out = `samples` evenly strided rows (an empty rank: rows at +infinity), then the min row, then the
// max row (block 0 folds the `parts` partials; min / max are exact and order-independent): what a rank contributes
// to the splitter sample (the rows arrive in id-hash order, so a strided sample is a random one).  Also clears
// the `zero_count` words at `zero` (the step's flags: no fill launch).
template <typename T>
__global__ __launch_bounds__(256) void k_range_fold(const typename MT<T>::V4 *__restrict__ rows, u32 n, u32 samples,
                                                     const T *__restrict__ partials, u32 parts,
                                                     typename MT<T>::V4 *__restrict__ out, u32 *__restrict__ zero, u32 zero_count) {
    typedef typename MT<T>::V4 V4;
    __shared__ T s_fold[32][8];
    const u32 tid = threadIdx.x;
    const u32 i = blockIdx.x * 256 + tid;
    if (i < samples) {
        V4 v;
        v.x = v.y = v.z = (T)INFINITY; v.w = (T)0;
        if (n) v = rows[samples > 1 ? (u32)(((u64)i * (n - 1)) / (samples - 1)) : 0u];
        out[i] = v;
    }
    if (blockIdx.x != 0) return;
    if (tid < zero_count) zero[tid] = 0;
    {                                               // 256 threads = 32 slices of the partials x 8 columns, loads in flight together
        const u32 col = tid & 7u, part = tid >> 3;
        T a = col < 4 ? (T)INFINITY : -(T)INFINITY;
#pragma unroll
        for (int j = 0; j < SR_BLOCKS / 32; j++) {
            const u32 b = part + 32u * j;
            if (b < parts) { const T t = partials[(u64)b * 8 + col]; a = col < 4 ? (t < a ? t : a) : (t > a ? t : a); }
        }
        s_fold[part][col] = a;
    }
    __syncthreads();
    if (tid < 8) {
        T a = s_fold[0][tid];
        for (int k = 1; k < 32; k++) { const T t = s_fold[k][tid]; a = tid < 4 ? (t < a ? t : a) : (t > a ? t : a); }
        reinterpret_cast<T *>(out + samples)[tid] = a;      // the min row, then the max row
    }
}

// The REGION of a rank for the halo selection: COL_REGION_BOXES = 8 boxes, one per octant of the global scene range
// (the top three bits of a sphere's Morton code), each (min centre - max r, 0, max centre + max r, 0) over the owned
// spheres of that octant -- conservative (exact for equal radii), which is all a halo selection needs; an octant
// without spheres gets an inverted box that nothing overlaps.  A rank owns a Morton RANGE: inside one octant that
// is a compact piece, while ONE box around a range that spills over an octant boundary by a few spheres would
// cover a quarter of the scene.  range8 == NULL (no repartition): everything counts as octant 0.
// Two launches like k_range (partials: 56 scalars per block); the fold also clears zero[0..zero_count).
constexpr int RG_VALS = 7 * COL_REGION_BOXES;      // per octant: min x, y, z; max x, y, z; max r
template <typename T>
__global__ __launch_bounds__(REGION_NT) void k_region_part(const typename MT<T>::V4 *__restrict__ rows, u32 n, const T *__restrict__ range8,
                                                      T *__restrict__ partials, const u32 *__restrict__ n_dev) {
    typedef typename MT<T>::V4 V4;
    n = count_of(n, n_dev);                  // device-side count (col_common.h)
    __shared__ T s_part[REGION_NT / 64][RG_VALS];
    const u32 tid = threadIdx.x, lane = lane_id(), w = tid / 64;
    T v[COL_REGION_BOXES][7];
#pragma unroll
    for (int o = 0; o < COL_REGION_BOXES; o++)
#pragma unroll
        for (int k = 0; k < 7; k++) v[o][k] = k < 3 ? (T)INFINITY : -(T)INFINITY;
    // The octant of a row = the top bit of each axis' 10-bit coordinate, exactly as morton30 quantises it (the octant
    // must be the code's own top bits: with a geometric 'side of the middle' test the few spheres between 0.5 and
    // 512/1023 of the range straddle the owner's Morton ranges, their octant boxes become slabs across the scene, and
    // the N = 2 rehearsal sends 100 834 ghosts instead of 7 261).  quantize() is monotone in its argument, so per
    // axis there is ONE threshold: the smallest coordinate whose quantised value reaches 512.  Three threads find
    // it by bisection over the ordered bit patterns (32 / 64 evaluations); the 2 M rows then cost three compares each
    // instead of three correctly rounded divisions (this kernel took 91 us beside the local pipeline, now ~15).
    __shared__ T s_thr[3];
    if (tid < 3) {
        typedef typename MT<T>::Bits Bits;
        T thr = (T)INFINITY;
        if (range8) {
            const T mn = range8[tid], mx = range8[4 + tid];
            constexpr Bits SIGN = (Bits)1 << (8 * sizeof(T) - 1);
            auto from_ord = [](Bits o) -> T { const Bits b = (o & SIGN) ? (o ^ SIGN) : ~o; return *reinterpret_cast<const T *>(&b); };
            // ordered patterns: 0 .. SIGN-1 the negative floats (most negative first), SIGN .. the non-negative ones;
            // search [ord(-inf), ord(+inf)] for the first value x with quantize(x) >= 512
            const T ninf = -(T)INFINITY, pinf = (T)INFINITY;
            Bits lo = ~*reinterpret_cast<const Bits *>(&ninf), hi = *reinterpret_cast<const Bits *>(&pinf) ^ SIGN;
            if (quantize<T>(pinf, mn, mx) >= 512u) {
                while (lo < hi) {
                    const Bits mid = lo + ((hi - lo) >> 1);
                    if (quantize<T>(from_ord(mid), mn, mx) >= 512u) hi = mid; else lo = mid + 1;
                }
                thr = from_ord(lo);
            } else thr = (T)NAN;                    // (a degenerate range: no coordinate reaches the upper half)
        }
        s_thr[tid] = thr;
    }
    __syncthreads();
    const T tx = s_thr[0], ty = s_thr[1], tz = s_thr[2];
    const u32 stride = gridDim.x * REGION_NT;
    for (u32 i0 = blockIdx.x * REGION_NT + tid; i0 < n; i0 += RU * stride) {
        V4 cc[RU];
#pragma unroll
        for (int u = 0; u < RU; u++) cc[u] = rows[min(i0 + u * stride, n - 1)];     // (a repeated row changes no min / max)
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const V4 c = cc[u];
            const u32 oct = range8 ? ((u32)(c.x >= tx) << 2) | ((u32)(c.y >= ty) << 1) | (u32)(c.z >= tz) : 0u;
            const T e[3] = {c.x, c.y, c.z};
#pragma unroll
            for (int o = 0; o < COL_REGION_BOXES; o++)
                if (oct == (u32)o) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        v[o][k] = e[k] < v[o][k] ? e[k] : v[o][k];
                        v[o][3 + k] = e[k] > v[o][3 + k] ? e[k] : v[o][3 + k];
                    }
                    v[o][6] = c.w > v[o][6] ? c.w : v[o][6];
                }
        }
    }
#pragma unroll
    for (int o = 0; o < COL_REGION_BOXES; o++)
#pragma unroll
        for (int k = 0; k < 7; k++) {
            T a = v[o][k];
#pragma unroll
            for (int d = COL_WAVE / 2; d > 0; d >>= 1) {
                const T t = __shfl_xor(a, d, COL_WAVE);
                a = k < 3 ? (t < a ? t : a) : (t > a ? t : a);
            }
            if (lane == 0) s_part[w][o * 7 + k] = a;
        }
    __syncthreads();
    if (tid < (u32)RG_VALS) {
        const bool is_min = tid % 7 < 3;
        T a = s_part[0][tid];
        for (int i = 1; i < REGION_NT / 64; i++) { const T t = s_part[i][tid]; a = is_min ? (t < a ? t : a) : (t > a ? t : a); }
        partials[(u64)blockIdx.x * RG_VALS + tid] = a;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_region_fold(const T *__restrict__ partials, u32 parts, typename MT<T>::V4 *__restrict__ out,
                                                      u32 *__restrict__ zero, u32 zero_count) {
    typedef typename MT<T>::V4 V4;
    __shared__ T s_part[4][RG_VALS];
    __shared__ T s_all[RG_VALS];
    const u32 tid = threadIdx.x;
    if (tid < zero_count) zero[tid] = 0;
    if (tid < 4u * RG_VALS) {                      // 4 slices of the partials x 56 columns, 16 loads in flight per thread
        const u32 col = tid % RG_VALS, part = tid / RG_VALS;
        const bool is_min = col % 7 < 3;
        T a = is_min ? (T)INFINITY : -(T)INFINITY;
        for (u32 b0 = part; b0 < parts; b0 += 64) {
            T t[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const u32 b = b0 + 4u * j;
                t[j] = partials[(u64)(b < parts ? b : b0) * RG_VALS + col];
            }
#pragma unroll
            for (int j = 0; j < 16; j++) a = is_min ? (t[j] < a ? t[j] : a) : (t[j] > a ? t[j] : a);
        }
        s_part[part][col] = a;
    }
    __syncthreads();
    if (tid < (u32)RG_VALS) {
        const bool is_min = tid % 7 < 3;
        T a = s_part[0][tid];
        for (int i = 1; i < 4; i++) { const T t = s_part[i][tid]; a = is_min ? (t < a ? t : a) : (t > a ? t : a); }
        s_all[tid] = a;
    }
    __syncthreads();
    if (tid < (u32)COL_REGION_BOXES) {
        const T *a = s_all + tid * 7;
        const T r = a[6];                       // (an empty octant: +inf - (-inf) = +inf, -inf + (-inf) = -inf)
        V4 lo, hi;
        lo.x = a[0] - r; lo.y = a[1] - r; lo.z = a[2] - r; lo.w = (T)0;
        hi.x = a[3] + r; hi.y = a[4] + r; hi.z = a[5] + r; hi.w = (T)0;
        out[2 * tid] = lo; out[2 * tid + 1] = hi;
    }
}

// One block: the global scene range (fold of every rank's [min row, max row]) -> range8; the Morton codes of ALL
// gathered rows (world x (samples + 2), at most SPL_MAX) under that range stay in registers; splitter q = the code
// of rank q * (count / world) among them, found WITHOUT sorting: a radix select on the 30 code bits, 8 bits per
// level, one wave per splitter -- per level an LDS histogram of the codes that share the splitter's prefix, then
// the wave finds the bin its rank falls into.  (A bitonic sort of the codes in LDS took 16 us for one rank's 1024
// codes and 91 barrier stages for eight ranks' 8192; this is 12 barriers whatever the world size.)
constexpr u32 SPL_MAX = 16384, SPL_PER = SPL_MAX / 1024, SPL_Q = 15;
template <typename T>
__global__ __launch_bounds__(1024) void k_splitters(const typename MT<T>::V4 *__restrict__ gathered, u32 world, u32 samples,
                                                     T *__restrict__ range8, u32 *__restrict__ splitters) {
    typedef typename MT<T>::V4 V4;
    __shared__ u32 s_hist[SPL_Q][256];
    __shared__ u32 s_prefix[SPL_Q], s_rank[SPL_Q];
    __shared__ T s_range[8];
    const u32 tid = threadIdx.x, lane = lane_id(), w = tid / 64;
    const u32 per = samples + 2, count = world * per, nq = world - 1;
    if (tid < 8) {
        const u32 k = tid;
        T acc = k < 4 ? (T)INFINITY : -(T)INFINITY;
        for (u32 q = 0; q < world; q++) {
            const T v = reinterpret_cast<const T *>(gathered + (u64)q * per + samples + (k >> 2))[k & 3];
            acc = k < 4 ? (v < acc ? v : acc) : (v > acc ? v : acc);
        }
        s_range[k] = acc;
        range8[k] = acc;
    }
    if (tid < nq) { s_prefix[tid] = 0; s_rank[tid] = (tid + 1) * (count / world); }
    __syncthreads();
    u32 code[SPL_PER];
#pragma unroll
    for (int j = 0; j < (int)SPL_PER; j++) {
        const u32 i = tid + 1024u * j;
        code[j] = 0xFFFFFFFFu;                   // (not a 30-bit code: matches no prefix, counted nowhere)
        if (i < count) {
            const V4 c = gathered[i];
            code[j] = morton30<T>(c.x, c.y, c.z, s_range[0], s_range[1], s_range[2], s_range[4], s_range[5], s_range[6]);
        }
    }
    // levels: code bits [29:22], [21:14], [13:6], [5:0]
    for (int level = 0; level < 4; level++) {
        const int shift = level < 3 ? 22 - 8 * level : 0, hi = shift + (level < 3 ? 8 : 6);      // prefix = code >> hi
        const u32 mask = level < 3 ? 255u : 63u;
        const u32 nh = level == 0 ? 1u : nq;         // (level 0: every splitter has the empty prefix, one histogram)
        for (u32 i = tid; i < nh * 256; i += 1024) (&s_hist[0][0])[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (int)SPL_PER; j++) {
            const u32 c = code[j];
            if (c == 0xFFFFFFFFu) continue;
            if (level == 0) atomicAdd(&s_hist[0][c >> 22], 1u);
            else
                for (u32 t = 0; t < nq; t++)
                    if ((c >> hi) == s_prefix[t]) atomicAdd(&s_hist[t][(c >> shift) & mask], 1u);
        }
        __syncthreads();
        u32 new_prefix = 0, new_rank = 0;
        if (w < nq) {                                // wave w: the bin of splitter w = first bin whose running count exceeds its rank
            const u32 *h = s_hist[level == 0 ? 0 : w];
            const u32 c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
            const u32 incl = wave_incl_scan(c0 + c1 + c2 + c3), r = s_rank[w];
            const u64 over = __ballot(incl > r);
            const int l = (int)__builtin_ctzll(over);            // (over != 0: the rank is below the number of codes with this prefix)
            if ((int)lane == l) {
                u32 before = incl - (c0 + c1 + c2 + c3), bin = 4 * lane;
                if (before + c0 <= r) { before += c0; bin++;
                    if (before + c1 <= r) { before += c1; bin++;
                        if (before + c2 <= r) { before += c2; bin++; } } }
                new_prefix = (s_prefix[w] << (level < 3 ? 8 : 6)) | bin;
                new_rank = r - before;
            }
            new_prefix = __shfl(new_prefix, l, COL_WAVE);
            new_rank = __shfl(new_rank, l, COL_WAVE);
        }
        __syncthreads();
        if (w < nq && lane == 0) { s_prefix[w] = new_prefix; s_rank[w] = new_rank; }
        __syncthreads();
    }
    if (tid < nq) splitters[tid] = s_prefix[tid];
}

// One block per tile of the radix scatter that groups the spheres by owner: Morton code of every row under the
// global range, its owner (the number of splitters <= code), and the tile's owner histogram in the digit-major
// layout of radix.hip (hist[owner * nblocks + tile]) -- the code itself is never stored.
constexpr int OWN_NT = 1024;
template <typename T>
__global__ __launch_bounds__(OWN_NT) void k_owners(const typename MT<T>::V4 *__restrict__ rows, u32 n, const T *__restrict__ range8,
                                                 const u32 *__restrict__ splitters, u32 world, u32 tile, u32 nblocks,
                                                 u32 *__restrict__ dest, u32 *__restrict__ hist) {
    typedef typename MT<T>::V4 V4;
    __shared__ u32 sp[256];
    __shared__ u32 h[256];
    const u32 tid = threadIdx.x, lane = lane_id();
    if (tid + 1 < world) sp[tid] = splitters[tid];
    if (tid < 256) h[tid] = 0;
    const T mnx = range8[0], mny = range8[1], mnz = range8[2], mxx = range8[4], mxy = range8[5], mxz = range8[6];
    __syncthreads();
    const u64 base = (u64)blockIdx.x * tile;
    // 1024 threads per tile (a 4096-row tile at 256 threads left two waves per SIMD on a 2 M-row input: 28 us for 40 MB)
    // and four rows in flight per thread: the kernel is bound by the latency of its loads, not by their bytes
    for (u32 o0 = tid; o0 < tile; o0 += 4 * OWN_NT) {
        V4 c[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u32 o = o0 + u * OWN_NT;
            const u64 i = base + o;
            ok[u] = o < tile && i < n;
            if (ok[u]) c[u] = rows[i];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u64 i = base + o0 + u * OWN_NT;
            u32 q = 0xFFFFFFFFu;
            if (ok[u]) {
                const u32 code = morton30<T>(c[u].x, c[u].y, c[u].z, mnx, mny, mnz, mxx, mxy, mxz);
                u32 lo = 0, hi = world - 1;              // first index with sp[idx] > code
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    if (sp[mid] <= code) lo = mid + 1; else hi = mid;
                }
                q = lo;
                dest[i] = q;
            }
            if (o0 + u * OWN_NT - tid >= tile) continue;      // (block-uniform: no row of this group exists)
            if (world <= 16) {                          // few owners: one ballot per owner instead of 64 atomics on 8 words
                for (u32 d = 0; d < world; d++) {
                    const u64 m = __ballot(q == d);
                    if (m && lane == 0) atomicAdd(&h[d], (u32)__popcll(m));
                }
            } else if (ok[u]) atomicAdd(&h[q], 1u);
        }
    }
    __syncthreads();
    if (tid < world) hist[(u64)tid * nblocks + blockIdx.x] = h[tid];
}

// One block: exclusive scan, in place, of the first world * nblocks entries of the digit-major owner histogram
// (the other digits do not occur) + the number of spheres per owner.
__global__ __launch_bounds__(1024) void k_owner_offsets(u32 *__restrict__ hist, u32 nblocks, u32 world, u32 *__restrict__ counts) {
    __shared__ u32 s_warp[16];
    __shared__ u32 s_start[257];
    const u32 e = world * nblocks, chunk = (e + 1023) / 1024;
    const u32 lo = min(threadIdx.x * chunk, e), hi = min(lo + chunk, e);
    u32 sum = 0;
    for (u32 i = lo; i < hi; i++) sum += hist[i];
    u32 total;
    u32 run = block_excl_scan<1024>(sum, s_warp, &total);
    for (u32 i = lo; i < hi; i++) {
        const u32 v = hist[i];
        hist[i] = run;
        if (i % nblocks == 0) s_start[i / nblocks] = run;
        run += v;
    }
    if (threadIdx.x == 0) s_start[world] = total;
    __syncthreads();
    if (threadIdx.x < world) counts[threadIdx.x] = s_start[threadIdx.x + 1] - s_start[threadIdx.x];
}

// Stable grouping by owner and packing in ONE pass: the spheres this rank keeps go straight to the front of its owned
// arrays; the others into the slot of their owner -- a header record whose first word is the full length of the list,
// then min(length, slot) transport records.  What does NOT fit into a slot STAYS WITH THIS RANK: it is appended to the
// owned arrays behind the kept rows (owner order).  Which rank owns a sphere is a matter of load balance and halo size
// only -- the halo selection works on what a rank really owns -- so a slot that is too small costs locality for one
// step and never a sphere; the header (and flags[2], longest list seen) tells the host to grow the slots.
// A block takes the tile of rows whose owner counts k_owners wrote into column `tile` of the scanned owner histogram,
// so row i of owner q lands at position hist[q][tile] - hist[q][0] + (its rank among the tile's rows of owner q) of
// q's list.  (Rounds 1-2 sorted (owner, index) pairs with a radix scatter and gathered the rows through the
// permutation: 16 random bytes per row pull a whole line, 50 of the 120 us this step took at 2 M rows.)
// A wave owns a contiguous piece of the tile (16 waves per 4096-row tile: a block of 256 threads left the GPU two waves
// per SIMD at 2 M rows): it counts its rows per owner (one ballot per owner and 64 rows), the block adds up the waves
// before it, then the wave places every row.
template <typename T, int CH, int NW>      // NW waves per tile, CH = 64-row chunks per wave: tile = 64 * CH * NW
__global__ __launch_bounds__(NW * 64) void k_partition_scatter(const typename MT<T>::V4 *__restrict__ rows, const u32 *__restrict__ gids,
                                                            const u32 *__restrict__ dest, const u32 *__restrict__ hist,
                                                            const u32 *__restrict__ counts, u32 n, u32 world, u32 rank, u32 slot,
                                                            u32 capacity, u32 nblocks, u32 *__restrict__ send,
                                                            typename MT<T>::V4 *__restrict__ own_rows, u32 *__restrict__ own_gids,
                                                            T *__restrict__ own_radii, u32 *__restrict__ flags) {
    constexpr int RW = MT<T>::RW;
    __shared__ u32 s_wcnt[NW][16];
    __shared__ u32 s_stay[16];
    const u32 tid = threadIdx.x, lane = lane_id(), w = tid / COL_WAVE;
    const u32 b = blockIdx.x;
    const u64 wave_base = ((u64)b * NW + w) * (CH * COL_WAVE);      // a wave owns a contiguous piece of the tile
    // the owners of this wave's rows, all loads in flight together (a loop that loads as it goes is bound by latency)
    u32 d[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const u64 i = wave_base + c * COL_WAVE + lane;
        d[c] = i < n ? dest[i] : 0xFFFFFFFFu;
    }
    // where the rows that stay here go: this rank's own list first, then the overflow of the other lists in owner order
    if (tid < 16) {
        u32 before = 0;
        for (u32 q = 0; q < tid && q < world; q++) {
            const u32 c = counts[q];
            before += q == rank ? c : (c > slot ? c - slot : 0u);
        }
        s_stay[tid] = before;
    }
    if (b == 0 && tid < world && tid != rank) {
        const u32 cnt = counts[tid];
        u32 *hdr = send + (u64)RW * (tid < rank ? tid : tid - 1) * (slot + 1);
        for (int k = 0; k < RW; k++) hdr[k] = k == 0 ? cnt : 0u;
        if (cnt) atomicMax(&flags[2], cnt);
    }
    // phase 1: lane q counts this wave's rows of owner q
    u32 cnt = 0;
#pragma unroll
    for (int c = 0; c < CH; c++)
        for (u32 q = 0; q < world; q++) {
            const u64 m = __ballot(d[c] == q);
            if (lane == q) cnt += (u32)__popcll(m);
        }
    if (lane < 16) s_wcnt[w][lane] = cnt;
    __syncthreads();
    // lane q: position in q's list of this wave's first row of owner q
    u32 run = 0;
    if (lane < world) {
        run = hist[(u64)lane * nblocks + b] - hist[(u64)lane * nblocks];
        for (u32 ww = 0; ww < w; ww++) run += s_wcnt[ww][lane];
    }
    const u32 kept = counts[rank];
    // phase 2: every row's position in its owner's list (registers only), then the rows themselves
    u32 pos[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        u64 mine = 0;
        u32 add = 0;
        for (u32 q = 0; q < world; q++) {
            const u64 m = __ballot(d[c] == q);
            if (d[c] == q) mine = m;
            if (lane == q) add = (u32)__popcll(m);
        }
        const u32 base = (u32)__shfl((int)run, d[c] < world ? (int)d[c] : 0, COL_WAVE);
        run += add;
        pos[c] = base + mbcnt(mine);
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const u64 i = wave_base + c * COL_WAVE + lane;
        const u32 q = d[c];
        if (q >= world) continue;                                   // (beyond n)
        const typename MT<T>::V4 r = rows[i];
        const u32 gid = gids[i];
        if (q != rank && pos[c] < slot) {
            rec_store<T>(send + (u64)RW * ((u64)(q < rank ? q : q - 1) * (slot + 1) + 1 + pos[c]), r, gid);
            continue;
        }
        const u32 before = s_stay[q] - (q > rank ? kept : 0u);
        const u64 dst = q == rank ? (u64)pos[c] : (u64)kept + before + (pos[c] - slot);
        if (dst >= capacity) continue;
        own_rows[dst] = r;
        own_gids[dst] = gid;
        own_radii[dst] = r.w;
    }
}

// Received slots (blockIdx.y = slot of the k-th other rank, in rank order) -> the owned arrays, behind the
// spheres the rank kept (its own list and the overflow of the lists it sent, k_partition_pack); block (0, 0)
// publishes the owned count: owned[0] = min(m, capacity), owned[1] = m, and (sequence number << 32 | m) into a
// host-visible word, which is all the host waits for before it sizes the local pipeline.
template <typename T>
__global__ __launch_bounds__(256) void k_partition_unpack(const u32 *__restrict__ recv, u32 world, u32 rank, u32 slot,
                                                           const u32 *__restrict__ counts, typename MT<T>::V4 *__restrict__ own_rows,
                                                           u32 *__restrict__ own_gids, T *__restrict__ own_radii, u32 capacity,
                                                           u32 *__restrict__ owned, unsigned long long *host_word, u32 seq,
                                                           u32 *__restrict__ flags) {
    constexpr int RW = MT<T>::RW;
    __shared__ u32 s_warp[4];
    __shared__ u32 s_off[256], s_len[256];
    const u32 tid = threadIdx.x, others = world - 1;
    const u32 len = tid < others ? recv[(u64)RW * tid * (slot + 1)] : 0u;
    u32 total;
    s_off[tid] = block_excl_scan<256>(min(len, slot), s_warp, &total);
    s_len[tid] = len;
    const u32 mine = tid < world ? counts[tid] : 0u;
    u32 stayed;                                  // rows already in the owned arrays: kept + overflow of the sent lists
    block_excl_scan<256>(tid == rank ? mine : (mine > slot ? mine - slot : 0u), s_warp, &stayed);
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (tid < others && len) atomicMax(&flags[2], len);
        if (tid == 0) {
            const u64 m = (u64)stayed + total;
            owned[0] = (u32)(m < capacity ? m : capacity);
            owned[1] = (u32)(m < 0xFFFFFFFFull ? m : 0xFFFFFFFFull);
            if (host_word)
                __hip_atomic_store(host_word, ((u64)seq << 32) | owned[1], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (blockIdx.y >= others) return;
    const u32 k = blockIdx.y, cnt = min(s_len[k], slot);
    const u32 *base = recv + (u64)RW * ((u64)k * (slot + 1) + 1);
    for (u32 i = blockIdx.x * 256 + tid; i < cnt; i += gridDim.x * 256) {
        const u64 dst = (u64)stayed + s_off[k] + i;
        if (dst >= capacity) break;
        u32 gid;
        const typename MT<T>::V4 c = rec_load<T>(base + (u64)RW * i, &gid);
        own_rows[dst] = c;
        own_gids[dst] = gid;
        own_radii[dst] = c.w;
    }
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

size_t col_partition_scratch_bytes(void) { return (size_t)SR_BLOCKS * RG_VALS * sizeof(double); }

static inline unsigned range_blocks(uint32_t n, int threads) {
    const uint64_t b = col_ceil_div(n, (uint64_t)threads * RU);
    return (unsigned)(b < 1 ? 1 : b > SR_BLOCKS ? SR_BLOCKS : b);
}

// scratch: col_partition_scratch_bytes() bytes (block partials)
int col_partition_sample(void *stream, const void *rows, uint32_t n, uint32_t samples, void *payload, void *scratch,
                         uint32_t *zero, uint32_t zero_count, int coord_bytes) {
    if (samples == 0 || !scratch || zero_count > 256) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    const unsigned g = range_blocks(n, RANGE_NT), gs = blocks_for(samples);
    COL_BY_COORD((k_range_part<float><<<dim3(g), dim3(RANGE_NT), 0, s>>>((const float4 *)rows, n, (float *)scratch)),
                 (k_range_part<double><<<dim3(g), dim3(RANGE_NT), 0, s>>>((const double4 *)rows, n, (double *)scratch)));
    COL_LAUNCH_OK();
    COL_BY_COORD((k_range_fold<float><<<dim3(gs), dim3(256), 0, s>>>((const float4 *)rows, n, samples, (const float *)scratch, g, (float4 *)payload, zero, zero_count)),
                 (k_range_fold<double><<<dim3(gs), dim3(256), 0, s>>>((const double4 *)rows, n, samples, (const double *)scratch, g, (double4 *)payload, zero, zero_count)));
    COL_LAUNCH_OK();
    return COL_OK;
}

// out = the COL_REGION_BOXES region boxes of rows[0..n) (2 rows of 4 scalars each; see k_region_part); range8 = the
// global scene range of the repartition, or NULL: one box (octant 0).  Clears zero[0..zero_count).
int col_region_boxes(void *stream, const void *rows, uint32_t n, const void *range8, void *scratch, void *out, uint32_t *zero,
                     uint32_t zero_count, int coord_bytes) {
    return col_region_boxes_dev(stream, rows, n, range8, scratch, out, zero, zero_count, coord_bytes, nullptr);
}

int col_region_boxes_dev(void *stream, const void *rows, uint32_t n, const void *range8, void *scratch, void *out, uint32_t *zero,
                         uint32_t zero_count, int coord_bytes, const uint32_t *n_dev) {
    if (!scratch || zero_count > 256) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    const unsigned g = range_blocks(n, REGION_NT);
    COL_BY_COORD((k_region_part<float><<<dim3(g), dim3(REGION_NT), 0, s>>>((const float4 *)rows, n, (const float *)range8, (float *)scratch, n_dev)),
                 (k_region_part<double><<<dim3(g), dim3(REGION_NT), 0, s>>>((const double4 *)rows, n, (const double *)range8, (double *)scratch, n_dev)));
    COL_LAUNCH_OK();
    COL_BY_COORD((k_region_fold<float><<<dim3(1), dim3(256), 0, s>>>((const float *)scratch, g, (float4 *)out, zero, zero_count)),
                 (k_region_fold<double><<<dim3(1), dim3(256), 0, s>>>((const double *)scratch, g, (double4 *)out, zero, zero_count)));
    COL_LAUNCH_OK();
    return COL_OK;
}

// gathered: [world][samples + 2] rows (col_partition_sample of every rank).  Writes the global range (range8),
// the world - 1 splitters, dest[i] = owner of row i, the scanned owner histogram for col_partition_group
// (256 * ceil(n / col_radix_tile(n, 4, 4)) words) and owner_counts[0..world).
int col_partition_plan(void *stream, const void *gathered, uint32_t world, uint32_t samples, const void *rows, uint32_t n,
                       void *range8, uint32_t *splitters, uint32_t *dest, uint32_t *hist, uint32_t *owner_counts, int coord_bytes) {
    if (world == 0 || world > SPL_Q + 1 || (uint64_t)world * (samples + 2) > SPL_MAX || samples + 2 < world) return COL_EINVAL;
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    COL_BY_COORD((k_splitters<float><<<dim3(1), dim3(1024), 0, s>>>((const float4 *)gathered, world, samples, (float *)range8, splitters)),
                 (k_splitters<double><<<dim3(1), dim3(1024), 0, s>>>((const double4 *)gathered, world, samples, (double *)range8, splitters)));
    COL_LAUNCH_OK();
    if (n == 0) {
        COL_HIP(hipMemsetAsync(owner_counts, 0, sizeof(u32) * world, s));
        return COL_OK;
    }
    const u32 tile = (u32)col_radix_tile(n, 4, 4);
    const u32 nb = (u32)col_ceil_div(n, tile);
    COL_BY_COORD((k_owners<float><<<dim3(nb), dim3(OWN_NT), 0, s>>>((const float4 *)rows, n, (const float *)range8, splitters, world, tile, nb, dest, hist)),
                 (k_owners<double><<<dim3(nb), dim3(OWN_NT), 0, s>>>((const double4 *)rows, n, (const double *)range8, splitters, world, tile, nb, dest, hist)));
    COL_LAUNCH_OK();
    k_owner_offsets<<<dim3(1), dim3(1024), 0, s>>>(hist, nb, world, owner_counts);
    COL_LAUNCH_OK();
    return COL_OK;
}

// send: (world - 1) slots of (slot + 1) transport records, in rank order without this rank; own_*: `capacity` rows
// (kept rows, then what did not fit into the slots).  flags[2] = max(flags[2], longest list).  world <= 16 (as
// col_partition_plan); dest / hist / owner_counts as col_partition_plan left them.
int col_partition_group(void *stream, const void *rows, const uint32_t *gids, uint32_t n, const uint32_t *dest,
                        const uint32_t *hist, const uint32_t *owner_counts, uint32_t world, uint32_t rank,
                        uint32_t slot, void *send, void *own_rows, uint32_t *own_gids,
                        void *own_radii, uint32_t capacity, uint32_t *flags, int coord_bytes) {
    if (world == 0 || world > 16 || rank >= world || slot == 0 || !flags) return COL_EINVAL;
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    const u32 tile = n ? (u32)col_radix_tile(n, 4, 4) : 1024u;
    const u32 nb = n ? (u32)col_ceil_div(n, tile) : 1u;          // (an empty rank still writes its slot headers)
#define COL_PSCATTER(T, V, CH, NW) k_partition_scatter<T, CH, NW><<<dim3(nb), dim3(NW * 64), 0, s>>>((const V *)rows, gids, dest, hist, owner_counts, n, world, rank, slot, capacity, nb, (u32 *)send, (V *)own_rows, own_gids, (T *)own_radii, flags)
    if (tile == 1024) { COL_BY_COORD(COL_PSCATTER(float, float4, 4, 4), COL_PSCATTER(double, double4, 4, 4)); }
    else if (tile == 4096) { COL_BY_COORD(COL_PSCATTER(float, float4, 4, 16), COL_PSCATTER(double, double4, 4, 16)); }
    else if (tile == 8192) { COL_BY_COORD(COL_PSCATTER(float, float4, 8, 16), COL_PSCATTER(double, double4, 8, 16)); }
    else return COL_EINVAL;
#undef COL_PSCATTER
    COL_LAUNCH_OK();
    return COL_OK;
}

// recv: the slots received from the other ranks, laid out as `send` above.  owned (2 device words) and
// *host_word (host-visible, e.g. col_host_alloc; may be NULL) receive the owned count, see k_partition_unpack.
int col_partition_unpack(void *stream, const void *recv, uint32_t world, uint32_t rank, uint32_t slot,
                         const uint32_t *owner_counts, void *own_rows, uint32_t *own_gids, void *own_radii, uint32_t capacity,
                         uint32_t *owned, void *host_word, uint32_t seq, uint32_t *flags, int coord_bytes) {
    if (world == 0 || world > 256 || rank >= world || slot == 0 || !flags || !owned) return COL_EINVAL;
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    unsigned gx = blocks_for(slot);
    if (gx > 512) gx = 512;
    if (world == 1) gx = 1;
    dim3 grid(gx, world > 1 ? world - 1 : 1);
    COL_BY_COORD((k_partition_unpack<float><<<grid, dim3(256), 0, s>>>((const u32 *)recv, world, rank, slot, owner_counts, (float4 *)own_rows, own_gids, (float *)own_radii, capacity, owned, (unsigned long long *)host_word, seq, flags)),
                 (k_partition_unpack<double><<<grid, dim3(256), 0, s>>>((const u32 *)recv, world, rank, slot, owner_counts, (double4 *)own_rows, own_gids, (double *)own_radii, capacity, owned, (unsigned long long *)host_word, seq, flags)));
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
    return col_select_overlap_multi_dev(stream, rows, n, boxes, peers, n_peers, stride, lists, counts, coord_bytes, nullptr);
}

int col_select_overlap_multi_dev(void *stream, const void *rows, uint32_t n, const void *boxes, const int *peers,
                                 int n_peers, uint32_t stride, uint32_t *lists, uint32_t *counts, int coord_bytes,
                                 const uint32_t *n_dev) {
    if (n_peers < 0 || n_peers > 8) return COL_EINVAL;
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    if (n == 0 || n_peers == 0) return COL_OK;
    PeerList pl;
    pl.n = n_peers;
    for (int k = 0; k < 8; k++) pl.q[k] = k < n_peers ? peers[k] : -1;
    hipStream_t s = col_stream(stream);
    // about 512 blocks; a thread takes at most SEL_ROWS rows (one mask bit each), so more blocks beyond 4 Mi rows
    u32 per_thread = (u32)col_ceil_div(n, 512ull * 256);
    if (per_thread < 1) per_thread = 1;
    if (per_thread > (u32)SEL_ROWS) per_thread = SEL_ROWS;
    const unsigned sel_blocks = (unsigned)col_ceil_div(n, 256ull * per_thread);
    COL_BY_COORD((k_select_multi<float><<<dim3(sel_blocks), dim3(256), 0, s>>>((const float4 *)rows, n, (const float4 *)boxes, pl, stride, lists, counts, per_thread, n_dev)),
                 (k_select_multi<double><<<dim3(sel_blocks), dim3(256), 0, s>>>((const double4 *)rows, n, (const double4 *)boxes, pl, stride, lists, counts, per_thread, n_dev)));
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

size_t col_ghost_scratch_bytes(uint32_t n_slots, uint32_t slot_records) {
    const uint64_t e = (uint64_t)n_slots * slot_records;
    return 4 * (((size_t)e * 4 + 255) & ~(size_t)255) + col_radix_scratch_bytes(e, 4, 4) + 256;
}

// scratch: col_ghost_scratch_bytes(n_slots, slot_records) bytes, or NULL for the lane-per-query walk (k_ghost).
int col_traverse_ghost_slots(void *stream, const void *rec, uint32_t n_slots, uint32_t slot_records, const void *bounds,
                             uint32_t n, const uint32_t *local_gids, uint32_t *pairs, uint32_t *counter, uint32_t capacity,
                             uint32_t *flags, int coord_bytes, void *scratch) {
    return col_traverse_ghost_slots_dev(stream, rec, n_slots, slot_records, bounds, n, local_gids, pairs, counter, capacity, flags,
                                        coord_bytes, scratch, nullptr);
}

int col_traverse_ghost_slots_dev(void *stream, const void *rec, uint32_t n_slots, uint32_t slot_records, const void *bounds,
                                 uint32_t n, const uint32_t *local_gids, uint32_t *pairs, uint32_t *counter, uint32_t capacity,
                                 uint32_t *flags, int coord_bytes, void *scratch, const uint32_t *n_dev) {
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    if (n_slots == 0 || slot_records == 0 || n == 0) return COL_OK;
    if ((capacity > 0 && !pairs) || !flags) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    const uint64_t e = (uint64_t)n_slots * slot_records;
    if (scratch && local_gids && e < 0xFFFFFFFFull && (uint64_t)n_slots * (slot_records + 1) < 0xFFFFFFFFull) {
        const size_t seg = ((size_t)e * 4 + 255) & ~(size_t)255;
        char *p = (char *)scratch;
        u32 *keys0 = (u32 *)p, *vals0 = (u32 *)(p + seg), *keys1 = (u32 *)(p + 2 * seg), *vals1 = (u32 *)(p + 3 * seg);
        void *sort_scratch = p + 4 * seg;
        dim3 grid(blocks_for(slot_records), n_slots);
        COL_BY_COORD((k_ghost_codes<float><<<grid, dim3(256), 0, s>>>((const u32 *)rec, slot_records, (const float4 *)bounds, keys0, vals0, flags)),
                     (k_ghost_codes<double><<<grid, dim3(256), 0, s>>>((const u32 *)rec, slot_records, (const double4 *)bounds, keys0, vals0, flags)));
        COL_LAUNCH_OK();
        int rc = col_radix_sort_low_passes(stream, keys0, keys1, vals0, vals1, e, sort_scratch, 2);
        if (rc) return rc;
        return col_traverse_ghost_packets(stream, pairs, counter, capacity, bounds, n, coord_bytes, (const u32 *)rec, vals1,
                                          flags + 1, (uint32_t)e, local_gids, n_dev);
    }
    dim3 grid((unsigned)col_ceil_div(slot_records, GW * 64), n_slots), block(GW * 64);
    COL_BY_COORD((k_ghost<float><<<grid, block, 0, s>>>((const float4 *)bounds, n, local_gids, pairs, counter, capacity, (const u32 *)rec, slot_records, flags, n_dev)),
                 (k_ghost<double><<<grid, block, 0, s>>>((const double4 *)bounds, n, local_gids, pairs, counter, capacity, (const u32 *)rec, slot_records, flags, n_dev)));
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

// Karras LBVH build, AABB refit and pair-overlap traversal.
// Replaces fillInternal / generateBVH / leafBounds / internalBounds / traverse
// (collision/collision.cl:55-226, enqueued at collision/collision.py:171-196).
//
// Outputs `nodes` (16 B records) and `bounds` (2 x vec4 per node) exactly as the reference
// lays them out: internal nodes [0, n-1), leaves [n-1, 2n-1) in sorted order, root 0.
//
// MI355X shape of the traversal: the reference walks from the root with a 64-entry private
// stack and touches 5 cache lines per step (node, two child nodes, two child bounds).  Here
// the unused lane w of every node's Bound carries two links written at build time --
//     min.w = "skip": the node that follows this subtree in depth-first order,
//     max.w = "down": left child (internal) or sphere id (leaf); in trees built by lbvh.hip a DENSE internal node
//             over at most 16 leaves carries a LEAF BLOCK mark instead (col_common.h: first leaf and count) and
//             the packet walk tests its leaves at once instead of descending
// -- so a traversal step is ONE 32-byte record (two float4 loads) and needs no stack.  A
// query for sorted leaf q only has to report leaves p > q (collision.cl:199-200), and those
// are exactly the subtrees hanging to the right of q's root path, i.e. the chain
// skip(leaf q), skip(skip(leaf q)), ...; it never visits the part of the tree left of q.
// Karras numbering makes skip() local: the right child that starts at leaf k is internal
// node k when node k's range runs forward, else leaf k (see right_child_at()).
#include "col_common.h"
#include <math.h>

namespace {

constexpr u32 END = 0xFFFFFFFFu;

template <typename T> struct BTypes;
template <> struct BTypes<float> {
    typedef float4 V4;
    typedef uint32_t Bits;
    struct alignas(4) V3 { float x, y, z; };
};
template <> struct BTypes<double> {
    typedef double4 V4;
    typedef uint64_t Bits;
    struct alignas(8) V3 { double x, y, z; };
};

// bounds are addressed as rows of 4 scalars: row 2*node = min, row 2*node+1 = max
template <typename T> __device__ __forceinline__ u32 link_load(const T *bounds, uint64_t row) {
    typedef typename BTypes<T>::Bits Bits;
    return (u32) reinterpret_cast<const Bits *>(bounds)[row * 4 + 3];
}
template <typename T> __device__ __forceinline__ void link_store(T *bounds, uint64_t row, u32 v) {
    typedef typename BTypes<T>::Bits Bits;
    reinterpret_cast<Bits *>(bounds)[row * 4 + 3] = (Bits)v;
}

// collision.cl:65-77 delta(i, j): common-prefix length of codes i and j, index bits as the
// tie-break for equal codes, -1 outside [0, n).
__device__ __forceinline__ int delta(const u32 *__restrict__ codes, u32 n, u32 i, u32 ci, int64_t j) {
    if (j < 0 || j >= (int64_t)n) return -1;
    const u32 cj = codes[j];
    return ci != cj ? __clz((int)(ci ^ cj)) : 32 + __clz((int)(i ^ (u32)j));
}

// The right child whose range starts at leaf k (1 <= k <= n-1): internal node k if that
// node's range runs forward from k, otherwise the single leaf k.
__device__ __forceinline__ u32 right_child_at(const u32 *__restrict__ codes, u32 n, u32 k) {
    if (k + 1 >= n) return (n - 1) + k;
    const u32 ck = codes[k];
    const bool fwd = delta(codes, n, k, ck, (int64_t)k + 1) > delta(codes, n, k, ck, (int64_t)k - 1);
    return fwd ? k : (n - 1) + k;
}

template <typename T>
__global__ __launch_bounds__(256) void k_build(const u32 *__restrict__ codes, const u32 *__restrict__ ids,
                                                col_node *__restrict__ nodes, T *__restrict__ bounds, u32 n) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 leaf_start = n - 1;

    // collision.cl:55-63 (fillInternal fills the leaves)
    const u32 id = ids[i];
    nodes[leaf_start + i].right_edge = i;
    nodes[leaf_start + i].data[0] = id;
    if (bounds) {
        const uint64_t row = 2ull * (leaf_start + i);
        link_store(bounds, row, i + 1 < n ? right_child_at(codes, n, i + 1) : END);
        link_store(bounds, row + 1, id);
    }
    if (i >= leaf_start) return;

    // collision.cl:81-121 (Karras 2012)
    const u32 ci = codes[i];
    const int dir = delta(codes, n, i, ci, (int64_t)i + 1) > delta(codes, n, i, ci, (int64_t)i - 1) ? 1 : -1;
    const int delta_min = delta(codes, n, i, ci, (int64_t)i - dir);
    int64_t len_max = 2;
    while (delta(codes, n, i, ci, (int64_t)i + dir * len_max) > delta_min) len_max *= 2;
    int64_t len = 0;
    for (int64_t t = len_max / 2; t > 0; t /= 2)
        if (delta(codes, n, i, ci, (int64_t)i + dir * (len + t)) > delta_min) len += t;
    const u32 j = (u32)((int64_t)i + dir * len);
    const int delta_node = delta(codes, n, i, ci, (int64_t)j);
    int64_t s = 0, t = len;
    do {
        t = (t + 1) / 2;
        if (delta(codes, n, i, ci, (int64_t)i + dir * (s + t)) > delta_node) s += t;
    } while (t > 1);
    const u32 gamma = dir > 0 ? (u32)(i + s) : (u32)(i - s - 1);
    const u32 lo = min(i, j), hi = max(i, j);
    const u32 child_a = (lo == gamma) ? leaf_start + gamma : gamma;
    const u32 child_b = (hi == gamma + 1) ? leaf_start + gamma + 1 : gamma + 1;

    nodes[i].right_edge = hi;
    nodes[i].data[0] = child_a;
    nodes[i].data[1] = child_b;
    nodes[child_a].parent = i;
    nodes[child_b].parent = i;
    if (bounds) {
        link_store(bounds, 2ull * i, hi + 1 < n ? right_child_at(codes, n, hi + 1) : END);
        link_store(bounds, 2ull * i + 1, child_a);      // (no leaf-block marks here: they need the boxes, see lbvh.hip)
    }
}

// leafBounds + internalBounds (collision.cl:128-162) for ANY binary tree given by parent / children
// links (the production path uses lbvh.hip, which needs the LBVH's contiguous leaf ranges).
//
// The reference's protocol -- every leaf walks up, the second arrival at a node merges -- hands
// boxes between workgroups through a flag, which on MI355X (8 XCDs, non-coherent L2s) needs an
// agent-scope release + acquire fence pair per level per wave: 6.3 ms at 1 M leaves.  Here:
//   k_refit_leaves   leaf boxes.
//   k_refit_sweep    x R, level-synchronous: a node whose two children were finished by an EARLIER
//                    launch merges them (visibility comes from the kernel boundary: no fence, no
//                    atomic).  R = log2(n) + 16 covers LBVHs.  Round r records in the (otherwise
//                    unused) flag word of leaf r whether it finished anything; if round r-1 did
//                    not, the tree is complete and round r exits on one scalar load.
//   k_refit_finish   whatever is still unfinished (trees deeper than R): the reference's walk with
//                    fences, started from every finished node whose parent is not.
// Results are identical in every case: min/max are exact.
constexpr u32 DONE = 0x80000000u;

template <typename T>
__global__ __launch_bounds__(256) void k_refit_leaves(T *__restrict__ bounds, u32 *__restrict__ flags,
                                                       const T *__restrict__ coords, const T *__restrict__ radii,
                                                       const col_node *__restrict__ nodes, u32 n) {
    typedef typename BTypes<T>::V4 V4;
    typedef typename BTypes<T>::V3 V3;
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 leaf_start = n - 1;
    const u32 node = leaf_start + i;
    const u32 id = nodes[node].data[0];
    const V4 c = reinterpret_cast<const V4 *>(coords)[id];
    const T r = radii[id];
    V3 mn = {c.x - r, c.y - r, c.z - r};
    V3 mx = {c.x + r, c.y + r, c.z + r};
    *reinterpret_cast<V3 *>(bounds + 8ull * node) = mn;
    *reinterpret_cast<V3 *>(bounds + 8ull * node + 4) = mx;
}

template <typename T> struct Box3 { T lo[3], hi[3]; };
template <typename T> __device__ __forceinline__ Box3<T> load_box(const T *bounds, u32 node) {
    typedef typename BTypes<T>::V4 V4;
    const V4 a = reinterpret_cast<const V4 *>(bounds)[2ull * node], b = reinterpret_cast<const V4 *>(bounds)[2ull * node + 1];
    Box3<T> r;
    r.lo[0] = a.x; r.lo[1] = a.y; r.lo[2] = a.z; r.hi[0] = b.x; r.hi[1] = b.y; r.hi[2] = b.z;
    return r;
}
template <typename T> __device__ __forceinline__ void merge_store(T *bounds, u32 node, Box3<T> &a, const Box3<T> &b) {
    typedef typename BTypes<T>::V3 V3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        a.lo[k] = b.lo[k] < a.lo[k] ? b.lo[k] : a.lo[k];
        a.hi[k] = b.hi[k] > a.hi[k] ? b.hi[k] : a.hi[k];
    }
    V3 mn = {a.lo[0], a.lo[1], a.lo[2]}, mx = {a.hi[0], a.hi[1], a.hi[2]};
    *reinterpret_cast<V3 *>(bounds + 8ull * node) = mn;        // lane w (traversal links) is preserved
    *reinterpret_cast<V3 *>(bounds + 8ull * node + 4) = mx;
}

template <typename T>
__global__ __launch_bounds__(256) void k_refit_sweep(T *__restrict__ bounds, u32 *__restrict__ flags,
                                                      const col_node *__restrict__ nodes, u32 n, u32 round,
                                                      u32 *__restrict__ progress) {
    if (progress && round > 1 && progress[round - 1] == 0) return;     // wave-uniform: the tree is complete
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const u32 leaf_start = n - 1;
    bool finished = false;
    if (i < leaf_start && !(flags[i] & DONE)) {
        const u32 ca = nodes[i].data[0], cb = nodes[i].data[1];
        // a child counts only if an earlier launch finished it: its box is then visible without a fence
        const u32 fa = ca >= leaf_start ? DONE : flags[ca], fb = cb >= leaf_start ? DONE : flags[cb];
        if ((fa & DONE) && (fb & DONE) && (fa & ~DONE) < round && (fb & ~DONE) < round) {
            Box3<T> box = load_box(bounds, ca);
            merge_store(bounds, i, box, load_box(bounds, cb));
            flags[i] = DONE | round;
            finished = true;
        }
    }
    if (progress && __syncthreads_or(finished) && threadIdx.x == 0) progress[round] = 1;
}

template <typename T>
__global__ __launch_bounds__(256) void k_refit_finish(T *__restrict__ bounds, u32 *__restrict__ flags,
                                                       const col_node *__restrict__ nodes, u32 n,
                                                       const u32 *__restrict__ last_progress) {
    if (last_progress && *last_progress == 0) return;       // the last sweep had nothing left to do
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const u32 leaf_start = n - 1;
    if (i == 0 || i >= 2 * n - 1) return;
    if (i < leaf_start && !(flags[i] & DONE)) return;       // not finished itself
    if (flags[nodes[i].parent] & DONE) return;              // parent already finished by a sweep
    // i is finished, its parent is not: collision.cl:143-162 from here up
    u32 cur = i;
    Box3<T> box = load_box(bounds, cur);
    do {
        const u32 parent = nodes[cur].parent;
        __threadfence();
        if ((atomicAdd(&flags[parent], 1u) & 3u) < 1u) break;   // first arrival: the sibling finishes the node
        __threadfence();
        const u32 ca = nodes[parent].data[0], cb = nodes[parent].data[1];
        merge_store(bounds, parent, box, load_box(bounds, ca == cur ? cb : ca));
        cur = parent;
    } while (cur != 0);
}

// traverse (collision.cl:174-226): pairs (id[q], id[p]) for every sorted leaf p > q whose box
// strictly overlaps q's (collision.cl:164-166); the counter counts every hit, pairs beyond
// `capacity` are dropped (collision.cl:203-207).
//
// Packet traversal, one wave = 64 consecutive sorted leaves (spatially compact, Morton order).
// A lane-per-query walk spends its time in the texture addresser: every step is 64 different
// cache lines per wave-instruction and a wave runs as long as its slowest lane (measured: 12
// visits per query but 34 trips per wave, 4 us per trip).  Here the CANDIDATE is wave-uniform and
// the 64 queries are tested against it in parallel:
//   phase 1  pairs inside the packet: lane q meets lane (q + r) mod 64 for r = 1..32 (wave
//            shuffles), every unordered pair once.
//   phase 2  candidates beyond the wave: the skip-chain that starts after the wave's LAST leaf is
//            the same for all 64 queries, so one uniform walk serves them all: one 32-byte
//            record per step at a wave-uniform address, descend iff ANY lane's box overlaps.
// Hits are appended with one atomic per candidate (ballot + mbcnt).
// Pair output.  One global counter cannot take an atomic per hit: a single address retires
// ~88 atomics/us on MI355X, and the first version of this kernel spent 0.3 ms of 0.39 ms there
// for 32 k pairs.  Hits are staged in a wave-private LDS buffer; a wave flushes it with ONE
// global atomic when it is full, and at the end of the (persistent) block the 16 waves' leftovers
// are flushed together with one atomic for the whole block.
constexpr int TW = 16;            // waves per block
constexpr int TT = TW * 64;       // 1024 threads
constexpr int CAPW = 512;         // staged pairs per wave (2 KB)
constexpr int CHUNK_PAIRS = 8192;  // list slots a block takes from the global counter at a time (chunked allocation)
// header + one hole per block of the chunked allocation (col_traverse_chunked): done = blocks finished (the last one sums up),
// A = slots allocated, T = pairs found, overflow = A > capacity (the list cannot be closed: the exact walk runs again)
struct ChunkHdr { u32 done, A, T, overflow, pad[12]; uint2 holes[512]; };

struct PairSink {
    uint2 *buf;        // this wave's LDS staging area
    u32 count;         // wave-uniform number of staged pairs
    u32 *pairs;
    u32 *counter;
    u32 capacity;
    u32 lane;
    const u32 *gids;   // ghost queries: the second id of a staged pair is a local sphere index -> gids[index]; else NULL

    __device__ __forceinline__ void copy_out(u32 base, u32 cnt) {
        for (u32 i = lane; i < cnt; i += 64) {
            const u32 k = base + i;
            if (k < capacity) {
                uint2 pr = buf[i];
                if (gids) pr.y = gids[pr.y];
                *reinterpret_cast<uint2 *>(pairs + 2ull * k) = pr;
            }
        }
    }
    // CHUNKED allocation (dense scenes): one address retires ~88 atomics/us, and a scene with 25 M pairs flushes 50 000
    // times -- 0.28 of config 3's 0.64 ms traversal were the pair counter.  `chunk` (LDS, one word per block:
    // end << 32 | cursor) is the block's current chunk of CHUNK_PAIRS list slots, taken from the global counter with ONE
    // atomic; waves reserve their 512-pair pieces from it with LDS atomics.  The wave whose reservation crosses the
    // chunk's end fills what is left, takes the next chunk and installs it with its own remainder already reserved (no
    // slot is lost inside a chunk); waves that find the chunk exhausted wait for that.  What a block does not use of its
    // LAST chunk is a hole in the list, recorded at the end of the kernel and closed by k_pairs_compact.
    unsigned long long *chunk;     // NULL: every flush reserves from the global counter itself

    __device__ __forceinline__ void copy_part(u32 from, u32 base, u32 cnt) {      // staged [from, from + cnt) -> list [base, ...)
        for (u32 i = lane; i < cnt; i += 64) {
            const u32 k = base + i;
            if (k < capacity) {
                uint2 pr = buf[from + i];
                if (gids) pr.y = gids[pr.y];
                *reinterpret_cast<uint2 *>(pairs + 2ull * k) = pr;
            }
        }
    }
    __device__ __forceinline__ void flush() {          // wave-level, count > 0
        if (chunk) {
            u32 done = 0;
            while (done < count) {
                const u32 want = count - done;
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(chunk, (unsigned long long)want);
                const u32 cur = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
                const u32 end = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
                if (cur + want <= end) {                         // it fits
                    copy_part(done, cur, want);
                    done = count;
                } else if (cur <= end) {                         // the one reservation that crosses the end
                    const u32 r = end - cur;
                    copy_part(done, cur, r);
                    done += r;
                    const u32 rest = count - done;
                    u32 nb = 0;
                    if (lane == 0) {
                        nb = atomicAdd(counter, (u32)CHUNK_PAIRS);
                        atomicExch(chunk, ((unsigned long long)(nb + (u32)CHUNK_PAIRS) << 32) | (nb + rest));
                    }
                    nb = (u32)__builtin_amdgcn_readfirstlane((int)nb);
                    copy_part(done, nb, rest);
                    done = count;
                } else {                                         // exhausted: the crossing wave installs the next chunk
                    while ((u32)(__hip_atomic_load(chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32) == end)
                        __builtin_amdgcn_s_sleep(1);
                }
            }
            count = 0;
            return;
        }
        u32 base = 0;
        if (lane == 0) base = atomicAdd(counter, count);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        copy_out(base, count);
        count = 0;
    }
    // `hits` is wave-uniform and non-zero; lanes set in it append their own (qid, pid)
    __device__ __forceinline__ void emit(u64 hits, u32 qid, u32 pid) {
        const u32 add = (u32)__popcll(hits);
        if (count + add > (u32)CAPW) flush();
        if ((hits >> lane) & 1ull) buf[count + mbcnt(hits)] = make_uint2(qid, pid);
        count += add;
    }
};

// min (MAX = false) or max of v over the 64 lanes, as a wave-uniform value.  DPP steps: xor 1, xor 2 (quad_perm),
// row_half_mirror, row_mirror (now every lane holds its row's result), row_bcast15 into rows 1 and 3,
// row_bcast31 into rows 2 and 3: lane 63 holds the result.  A lane a step does not reach gets its own value
// back (`old` = v), which min / max ignore.  Only the conservative screening box is built from it: v_min / v_max
// semantics for NaNs are fine there.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_f(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = (int)(u32)b, hi = (int)(u32)((u64)b >> 32);
    const u32 l2 = (u32)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    const u32 h2 = (u32)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double((long long)(((u64)h2 << 32) | l2));
}
template <typename T, bool MAX> __device__ __forceinline__ T wave_min_max(T v) {
    auto op = [](T a, T b) -> T {
        if constexpr (sizeof(T) == 4) return MAX ? __builtin_fmaxf(a, b) : __builtin_fminf(a, b);
        else return MAX ? __builtin_fmax(a, b) : __builtin_fmin(a, b);
    };
    v = op(v, dpp_f<0xB1, 0xF>(v));          // quad_perm [1, 0, 3, 2]
    v = op(v, dpp_f<0x4E, 0xF>(v));          // quad_perm [2, 3, 0, 1]
    v = op(v, dpp_f<0x141, 0xF>(v));         // row_half_mirror
    v = op(v, dpp_f<0x140, 0xF>(v));         // row_mirror
    v = op(v, dpp_f<0x142, 0xA>(v));         // row_bcast15 -> rows 1, 3
    v = op(v, dpp_f<0x143, 0xC>(v));         // row_bcast31 -> rows 2, 3
    if constexpr (sizeof(T) == 4) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    else {
        const long long b = __double_as_longlong(v);
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)b, 63), hi = (u32)__builtin_amdgcn_readlane((int)(u32)((u64)b >> 32), 63);
        return __longlong_as_double((long long)(((u64)hi << 32) | lo));
    }
}

// STATS: count phase-2 steps for col_traverse_stats (diagnostics); the production instance carries
// no counters.  VEC: record loads as vector loads at a uniform address (ablation).
// WALK: 0 = the generic phase-2 loop; 1 = the asm walk with the in-loop leaf-block test (needs 32-bit record offsets);
// 2 = the asm walk without it (marked nodes are entered through their leaf chain): the A/B reference of col_debug_traverse(128)
// GHOST: the queries are not the tree's own leaves but GHOST spheres (multi-GPU halo, csrc/multi.hip): transport records
// (x, y, z, r, gid) addressed through `order` -- record numbers sorted by the ghosts' Morton codes, so that the 64
// ghosts of a packet are neighbours in space -- `*count` of them; a packet has no phase 1 and its walk starts at the
// ROOT (no position pruning: every local leaf is a candidate); pairs come out as (ghost gid, local gid).
struct GhostArgs { const u32 *rec; const u32 *order; const u32 *count; const u32 *gids; };
struct NoGhost {};
template <bool G> struct GhostSel { typedef NoGhost T; };
template <> struct GhostSel<true> { typedef GhostArgs T; };

// (amdgpu_num_sgpr(80): above 80 only ONE 16-wave block fits a CU on this platform -- see the SGPR note at the asm walk;
// what does not fit is spilled to VGPR lanes, of which there are plenty)
// PROF (diagnostics, col_debug_traverse bit 12): every packet leaves the time of its phases (100 MHz clock) in
// g_walk_prof[packet] = {phase 1, phase 2, loading its own leaf records, start tick (low word)}
#define COL_PROF_PACKETS (1u << 18)
__device__ uint4 g_walk_prof[COL_PROF_PACKETS];
template <typename T, bool STATS, bool VEC, int WALK, bool GHOST = false, bool PROF = false>
__global__ __launch_bounds__(TT) __attribute__((amdgpu_num_sgpr(80))) void k_traverse(u32 *__restrict__ pairs, u32 *__restrict__ counter, u32 capacity,
                                                  const T *__restrict__ bounds, u32 n_bound, u64 *__restrict__ stats,
                                                  int mode, typename GhostSel<GHOST>::T ghost = {}, const u32 *__restrict__ n_dev = nullptr,
                                                  u32 *__restrict__ walk_order = nullptr) {
    const u32 n = count_of(n_bound, n_dev);          // device-side count (col_common.h): the grid is sized for the bound
    if (n_dev && n < (GHOST ? 1u : 2u)) return;     // (the host returns before the launch when it knows the count)
    typedef typename BTypes<T>::V4 V4;
    typedef typename BTypes<T>::Bits Bits;
    __shared__ uint2 s_buf[TW][CAPW];
    __shared__ u32 s_cnt[TW];
    __shared__ u32 s_base;
    const u32 lane = lane_id();
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));     // scalar: packet, q0, pos stay in SGPRs
    const u32 leaf_start = n - 1;
    const bool marks = n <= COL_LEAF_BLOCK_MAX_N;      // internal nodes over <= 16 leaves carry leaf-block marks
    const V4 *rows = reinterpret_cast<const V4 *>(bounds);
    __shared__ unsigned long long s_chunk;
    PairSink sink = {s_buf[w], 0u, pairs, counter, capacity, lane, nullptr, nullptr};
    // mode bit 6: run only if the chunked walk before this launch could not close its list (ChunkHdr::overflow);
    // mode bit 5: chunked allocation, `stats` is the ChunkHdr
    if (!STATS && (mode & 64) && reinterpret_cast<const ChunkHdr *>(stats)->overflow == 0) return;
    if (!STATS && (mode & 32)) {
        if (threadIdx.x == 0) s_chunk = 0;
        __syncthreads();
        sink.chunk = &s_chunk;
    }
    // (timing ablation, mode bit 4: every wave flushes through a counter of its own -- what the walk costs without the
    // contention of 50 k atomics on ONE address; the pair list is then garbage)
    if (!STATS && (mode & 16) && stats) sink.counter = reinterpret_cast<u32 *>(stats) + 32u * ((blockIdx.x * TW + w) & 8191u);
    u32 npackets = (n + 63) / 64;
    if constexpr (GHOST) {
        sink.gids = ghost.gids;
        npackets = (*ghost.count + 63u) / 64u;
    }
    u64 trips = 0, descents = 0, leaf_tests = 0, leaf_hits = 0, win[4] = {0, 0, 0, 0};

    // XCD-aware order: blockIdx % 8 is the XCD, each with its own L2.  Neighbouring packets walk nearly
    // the same nodes, so every XCD takes one contiguous eighth of the packet groups (16 packets = one
    // block's worth) and its blocks stride through it; the records a block misses in the scalar cache
    // are then mostly in its XCD's L2 already.  (Speed only; mode bit 3 restores the plain order.)
    const u32 ngroups = (npackets + TW - 1) / TW;
    const u32 xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = (gridDim.x + 7u - xcd) >> 3;
    const u32 g_lo = (u32)(((u64)ngroups * xcd) >> 3), g_hi = (u32)(((u64)ngroups * (xcd + 1)) >> 3);
    const bool plain = (mode & 8) || gridDim.x < 8;
    u32 group = plain ? blockIdx.x : g_lo + slot;
    const u32 g_end = plain ? ngroups : g_hi, g_step = plain ? gridDim.x : slots;
    // DYNAMIC PACKET ORDER (mode bit 8; `stats` = eight zeroed words, or the ChunkHdr with them in its pad): a wave's two
    // packets of the static order take 7..38 us each on BASELINE config 2, so the kernel waits 55 us for the unluckiest
    // waves while the packets' total is 30 us per wave slot (profiling instance, tools/walk3_ab.py).  Instead the
    // workgroup draws batches of 16 consecutive packets from its XCD's counter (one global atomic per batch) and its waves
    // take packets from the batch one by one (LDS word: end << 32 | cursor; the wave that finds cursor == end installs
    // the next batch, later ones wait for it; end = ~0: the XCD's range is used up).
    __shared__ unsigned long long s_sched;
    // (the workgroup's FIRST batch is the one its place on the XCD names -- no atomic: 64 workgroups asking one counter at the
    // start of the launch are served one after the other -- and the counter hands out the batches behind those; mode bit 7
    // draws the first one too: the A/B.  s_use_perm: did this call's k_cross order this XCD's packets (WALK ORDER below)?)
    __shared__ u32 s_use_perm;
    __shared__ u32 s_walk_t0[TW], s_ncost[TW];
    __shared__ uint2 s_costbuf[TW][8];        // (packet, ticks) of a wave's last walks: stored eight at a time, see WALK ORDER below
    const bool dyn = !STATS && !GHOST && (mode & 256) != 0 && !plain;
    if (dyn) {
        if (threadIdx.x == 0) {
            unsigned long long word = 0;
            if (!(mode & 128)) {
                const u32 p_lo = g_lo * TW, p_end = min(g_hi * (u32)TW, npackets);
                const u32 p_hi = (mode & 512) ? p_end + (p_end - min(p_lo, p_end)) : p_end;
                const u32 nb = p_lo + slot * (u32)TW;
                word = nb >= p_hi ? 0xFFFFFFFF00000000ull : (((unsigned long long)min(nb + (u32)TW, p_hi) << 32) | nb);
            }
            s_sched = word;
            s_use_perm = walk_order ? walk_order[2u * ((n_bound + 63u) / 64u) + xcd] : 0u;
        }
        if (threadIdx.x < TW) s_ncost[threadIdx.x] = 0;
        __syncthreads();
    }
    for (;;) {
        u32 packet;
        if (dyn) {
            for (;;) {
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(&s_sched, 1ull);
                const u32 cur = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
                const u32 end = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
                packet = END;
                if (end == END) break;                                   // nothing left for this XCD
                if (cur < end) { packet = cur; break; }
                if (cur == end) {                                        // this wave draws the next batch
                    // (the XCD's packet range and counter are worked out here, once per batch, rather than kept in registers:
                    // the kernel has none to spare)
                    const u32 x = blockIdx.x & 7u, ng = (npackets + TW - 1) / TW;
                    const u32 p_lo = (u32)(((u64)ng * x) >> 3) * TW, p_end = min((u32)(((u64)ng * (x + 1)) >> 3) * (u32)TW, npackets);
                    const u32 p_hi = (mode & 512) ? p_end + (p_end - min(p_lo, p_end)) : p_end;      // split units: walks, then phase 1s
                    u32 *sched_ctr = ((mode & 32) ? reinterpret_cast<ChunkHdr *>(stats)->pad : reinterpret_cast<u32 *>(stats)) + x;
                    u32 nb = 0;
                    if (lane == 0) nb = atomicAdd(sched_ctr, (u32)TW);
                    nb = p_lo + ((mode & 128) ? 0u : slots * (u32)TW) + (u32)__builtin_amdgcn_readfirstlane((int)nb);
                    // (one store for both outcomes: two `if (lane == 0)` blocks with a `break` between them end in
                    // "illegal VGPR to SGPR copy" in hipcc's backend)
                    const bool none = nb >= p_hi;
                    const u32 e = min(nb + (u32)TW, p_hi);
                    const unsigned long long word = none ? 0xFFFFFFFF00000000ull : (((unsigned long long)e << 32) | (nb + 1u));
                    if (lane == 0) atomicExch(&s_sched, word);
                    packet = none ? END : nb;
                    break;
                }
                while ((u32)__builtin_amdgcn_readfirstlane((int)(u32)(__hip_atomic_load(&s_sched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32)) == end)
                    __builtin_amdgcn_s_sleep(1);
            }
            if (packet == END) break;
        } else {
            if (group >= g_end) break;
            packet = group * TW + w;
            group += g_step;
            if (packet >= npackets) continue;
        }
        // SPLIT UNITS (mode bit 9, with the dynamic order): a packet is two units of work -- its walk (phase 2), drawn first,
        // and its phase 1, drawn when every walk of the XCD has been handed out: the short phase-1 units fill the wave slots
        // that the last long walks leave idle
        bool only1 = false, only2 = false;
        if (dyn && (mode & 512)) {
            const u32 x = blockIdx.x & 7u, ng = (npackets + TW - 1) / TW;
            const u32 p_lo = (u32)(((u64)ng * x) >> 3) * TW, p_end = min((u32)(((u64)ng * (x + 1)) >> 3) * (u32)TW, npackets);
            only1 = packet >= p_end;
            only2 = !only1;
            if (only1) packet -= p_end - p_lo;
        }
        // WALK ORDER (col_common.h): walk unit u of the dynamic order is packet perm[u] -- the XCD's packets, longest walk of the
        // previous call first; the walk's time goes into cost[] for the next call
        const bool ordered = !STATS && !GHOST && dyn && walk_order != nullptr && !only1;
        if (ordered && __builtin_amdgcn_readfirstlane((int)s_use_perm)) {
            // (bit 31: one of the XCD's long walks -- the launch ends when the last of them does, so it runs at raised wave priority
            // among the 7 other walks of its SIMD; mode bit 10 ignores the bit: the A/B)
            const u32 e = (u32)__builtin_amdgcn_readfirstlane((int)walk_order[(n_bound + 63u) / 64u + packet]);
            packet = e & 0x7FFFFFFFu;
            if ((e >> 31) && !(mode & 1024)) __builtin_amdgcn_s_setprio(3);
        }
        const u32 q0 = packet * 64, q = q0 + lane;
        u64 prof_t0 = 0, prof_t1 = 0, prof_t2 = 0;
        if constexpr (PROF) prof_t0 = wall_clock64();
        T lx = (T)INFINITY, ly = lx, lz = lx, hx = -lx, hy = -lx, hz = -lx;      // empty box: overlaps nothing
        u32 qid = 0, qskip = END;
        if constexpr (GHOST) {
            if (q < *ghost.count) {
                constexpr int RW = 4 * (int)(sizeof(T) / 4) + 1;        // words per transport record (multi.hip)
                const u32 *o = ghost.rec + (u64)RW * ghost.order[q];
                T c[4];
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int k = 0; k < 4; k++) c[k] = __uint_as_float(o[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) c[k] = __longlong_as_double((long long)(((u64)o[2 * k + 1] << 32) | o[2 * k]));
                }
                lx = c[0] - c[3]; ly = c[1] - c[3]; lz = c[2] - c[3];    // same arithmetic as leafBounds, collision.cl:139-140
                hx = c[0] + c[3]; hy = c[1] + c[3]; hz = c[2] + c[3];
                qid = o[RW - 1];
            }
        } else if (q < n) {
            const V4 a = rows[2ull * (leaf_start + q)], b = rows[2ull * (leaf_start + q) + 1];
            lx = a.x; ly = a.y; lz = a.z; hx = b.x; hy = b.y; hz = b.z;
            qskip = (u32) * reinterpret_cast<const Bits *>(&a.w);
            qid = (u32) * reinterpret_cast<const Bits *>(&b.w);
        }
        const int last = GHOST ? 0 : (int)min(63u, n - 1 - q0);        // wave-uniform: last valid lane
        if constexpr (PROF) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prof_t1 = wall_clock64();
        }
        // phase 1: pairs inside the packet.  Exact float tests cost ~30 wave-instructions per pair
        // and almost all of them fail, so pairs are first screened with boxes quantised to 8 bits
        // per axis inside the packet's union box (lo rounded down, hi rounded up: a real overlap
        // always survives).  The three axes sit in 10-bit fields of one word, and "a >= b in every
        // field" is one subtraction: bit 8 of each field of (a + 0x100) - b.
        if (!GHOST && !(mode & 1) && !only2) {
            // the packet's union box: six wave reductions on the DPP network (6 steps + one v_readlane each,
            // no LDS) instead of six xor-shuffle butterflies through ds_bpermute (36 LDS round trips)
            T ul[3], uh[3];
            if constexpr (sizeof(T) == 4) {
                // (f32: the six reductions interleaved in one asm block, one v_min/v_max_f32_dpp per step -- the
                // dependent steps of one reduction are six instructions apart, which covers the two wait states a
                // DPP read needs after a VALU write; hipcc spends five instructions per step on the same thing)
                float a0 = lx, a1 = ly, a2 = lz, b0 = hx, b1 = hy, b2 = hz;
#define COL_DPP_STEP(ctrl)                                                      \
                "v_min_f32_dpp %0, %0, %0 " ctrl "\n\tv_min_f32_dpp %1, %1, %1 " ctrl "\n\tv_min_f32_dpp %2, %2, %2 " ctrl "\n\t" \
                "v_max_f32_dpp %3, %3, %3 " ctrl "\n\tv_max_f32_dpp %4, %4, %4 " ctrl "\n\tv_max_f32_dpp %5, %5, %5 " ctrl "\n\t"
                // (s_nop 1 first: a DPP read needs two wait states after the VALU write of its source, and the
                // compiler does not track that hazard across the asm boundary -- the inputs are copies it makes
                // right before the block)
                asm volatile("s_nop 1\n\t"
                             COL_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                             COL_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                             COL_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                             COL_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                             COL_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                             COL_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                             "s_nop 1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2));
#undef COL_DPP_STEP
                const float red[6] = {a0, a1, a2, b0, b1, b2};
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    ul[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(red[a]), 63));
                    uh[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(red[3 + a]), 63));
                }
            } else {
                ul[0] = wave_min_max<T, false>(lx); ul[1] = wave_min_max<T, false>(ly); ul[2] = wave_min_max<T, false>(lz);
                uh[0] = wave_min_max<T, true>(hx); uh[1] = wave_min_max<T, true>(hy); uh[2] = wave_min_max<T, true>(hz);
            }
            const T mylo[3] = {lx, ly, lz}, myhi[3] = {hx, hy, hz};
            u32 qlo = 0, qhi = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const T ext = uh[a] - ul[a];
                const T sc = ext > (T)0 ? (T)255 / ext : (T)0;
                T fl = floor((mylo[a] - ul[a]) * sc), fh = ceil((myhi[a] - ul[a]) * sc);
                fl = fl < (T)0 ? (T)0 : (fl > (T)255 ? (T)255 : fl);      // also maps the empty box (inf) to 255 / 0
                fh = fh < (T)0 ? (T)0 : (fh > (T)255 ? (T)255 : fh);
                if (!(fl == fl)) fl = (T)0;                                  // NaN: stay conservative
                if (!(fh == fh)) fh = (T)255;
                qlo |= (u32)fl << (10 * a);
                qhi |= (u32)fh << (10 * a);
            }
            constexpr u32 CARRY = 0x10040100u;
            const u32 qhi_c = qhi + CARRY;
            // Rotation schedule: lane q meets lane (q + r) mod 64 for r = 1..32, i.e. every unordered
            // pair of the packet exactly once (r = 32 pairs the two halves: only the lower half tests).
            // (Broadcasting candidates 1..63 with v_readlane instead takes 63 rounds in which half the lanes
            // are idle: 0.107 vs 0.097 ms at 1 M spheres, round 1.)  The two screening words ROTATE through
            // the wave one lane per round on the DPP network (wave_rol:1 -- lane i takes lane i + 1's value;
            // one v_mov_dpp each, no LDS, no address arithmetic); the exact boxes are fetched by ds_bpermute
            // only in rounds where the screen lets somebody through.
            u32 rlo = qlo, rhi_c = qhi_c;
            auto round = [&](int r, bool lower_half_only) {
                rlo = (u32)__builtin_amdgcn_update_dpp((int)rlo, (int)rlo, 0x134, 0xF, 0xF, false);        // wave_rol:1
                rhi_c = (u32)__builtin_amdgcn_update_dpp((int)rhi_c, (int)rhi_c, 0x134, 0xF, 0xF, false);
                const u32 both = (qhi_c - rlo) & (rhi_c - qlo) & CARRY;     // my hi >= its lo, its hi >= my lo
                bool hit = both == CARRY;
                if (lower_half_only) hit = hit && lane < 32u;
                if (!__builtin_amdgcn_ballot_w64(hit)) return;
                const int pl = (int)((lane + r) & 63u);
                const T plx = __shfl(lx, pl, COL_WAVE), phx = __shfl(hx, pl, COL_WAVE);
                const T ply = __shfl(ly, pl, COL_WAVE), phy = __shfl(hy, pl, COL_WAVE);
                const T plz = __shfl(lz, pl, COL_WAVE), phz = __shfl(hz, pl, COL_WAVE);
                const u32 pid = (u32)__shfl((int)qid, pl, COL_WAVE);
                hit = hit && hx > plx && lx < phx && hy > ply && ly < phy && hz > plz && lz < phz;
                const u64 hits = __builtin_amdgcn_ballot_w64(hit);
                const bool mine_first = (int)lane < pl;                  // the earlier sorted leaf comes first
                if (hits) sink.emit(hits, mine_first ? qid : pid, mine_first ? pid : qid);
            };
#pragma unroll 1
            for (int r = 1; r < 32; r++) round(r, false);
            round(32, true);
        }

        if constexpr (PROF) prof_t2 = wall_clock64();
        // phase 2: everything after the packet's last leaf, one wave-uniform walk of the skip chain: one
        // 32-byte record per step at a wave-uniform address (scalar load), tested against the 64 query boxes.
        //
        // What bounds it (measured, round 2): the CU's ONE scalar unit, which serves all 32 resident waves.
        // A step was ~16 scalar instructions (64-bit address arithmetic, load, waits, five s_and over six
        // compare masks, compares and branches) against 6 vector ones, and every variant that ADDED scalar work
        // per step lost in proportion, whatever it saved elsewhere:
        //   * a "pair walk" fetching both children of a hit node together (half the dependent loads per
        //     packet, 2 x 38 instead of 63 records): 0.112 instead of 0.077 ms (uniform 1 M), 1.22 / 0.86 (config 3);
        //   * "leaf blocks" (a hit subtree of <= 8..64 leaves tested leaf by leaf from ONE coalesced vector
        //     load instead of being descended into): 7 more scalar instructions per step to track the subtree
        //     size: 0.085 instead of 0.069 ms uniform, 0.785 instead of 0.803 ms on config 3.
        // So the step is written for the scalar unit: a 32-bit record offset (one shift instead of shift + add +
        // addc; arrays below 4 GB), the six compares narrow EXEC (v_cmpx: lanes that fail drop out, the survivors
        // are the hits) instead of producing six masks to AND, and runs of misses -- more than half of all steps --
        // stay inside one asm loop of 7 scalar instructions per step.  0.077 -> 0.069 ms, config 3 0.88 -> 0.80 ms.
        u32 idx = GHOST ? 0u : (u32)__builtin_amdgcn_readlane((int)qskip, last);      // ghosts: from the root
        if ((mode & 2) || only1) idx = END;
        if constexpr (PROF) if (mode & (2048 | 4096)) {      // profiling instance only (variant bits 24 / 25): the walk with the lower / upper half of the packet's queries
            const bool keep = (mode & 2048) ? lane < 32u : lane >= 32u;
            if (!keep) { lx = ly = lz = (T)INFINITY; hx = hy = hz = -(T)INFINITY; }
        }
        if (ordered && lane == 0) s_walk_t0[w] = (u32)__builtin_amdgcn_s_memtime();      // (in LDS: two scalar registers held across the walk cost it 8-11 %)
        if constexpr (sizeof(T) == 4 && WALK == 1 && !VEC) {
            const char *rows_b = reinterpret_cast<const char *>(rows);
            const u32 buf_lds = (u32)(uintptr_t)(__attribute__((address_space(3))) void *)sink.buf;      // this wave's staging area
            // (the asm walk narrows EXEC with v_cmpx and restores the mask it found on entry -- the full wave here:
            // 1024-thread blocks, wave-uniform control flow above)
            //
            // LEAF BLOCKS (col_common.h): a hit internal node whose `down` link carries the mark covers <= 16
            // consecutive leaves.  Descending would cost ~2 DEPENDENT record fetches per leaf; instead its leaf records
            // -- consecutive 32-byte records from a known offset -- are fetched by independent 64-byte scalar loads,
            // three candidates in flight (two in s[40:55] -- the node record is done with once its skip link is saved --
            // and one in s[56:63]), and tested with the same v_cmpx chain; hits are staged exactly like a hit leaf's.
            // BASELINE config 3: 670 -> 170 walk steps per packet + 420 block candidates
            // (tests/analysis/sim_packet_walk.py).  A 64-byte load that starts at a block's last candidate reads 32
            // bytes beyond it: the next leaf's record, or -- behind the very last record -- the rest of the 64-byte
            // line that holds it (records are 32-byte aligned and their number is odd), never another page.
            // SGPR budget: the kernel must stay at <= 80 SGPRs in total -- above that only ONE 16-wave block fits a CU
            // on this platform (measured: +55 % on the uniform scene from one clobbered high register).
            // State across an exit to the pair sink (staging area full): bcnt = candidates of the current block still to
            // test (0 = not inside a block), boff = byte offset of the next one; idx = where the walk goes on.
            u32 bcnt = 0, boff = 0;
            while (idx != END || bcnt != 0) {      // (an interrupted block is finished even when the chain ends behind it)
                u64 hits, exec0;             // hits, on exit: the lanes to stage after a flush (0 = the chain ended)
                u32 pid, v0, v1, bskip;      // pid: scratch inside the asm, the hit candidate's id on exit
#define COL_CMPX6(LX, LY, LZ, HX, HY, HZ)                                         \
                             "v_cmpx_lt_f32_e32 vcc, " LX ", %[hx]\n\t"           /* lo.x < my hi.x */ \
                             "v_cmpx_gt_f32_e32 vcc, " HX ", %[lx]\n\t"           /* hi.x > my lo.x */ \
                             "v_cmpx_lt_f32_e32 vcc, " LY ", %[hy]\n\t"                              \
                             "v_cmpx_gt_f32_e32 vcc, " HY ", %[ly]\n\t"                              \
                             "v_cmpx_lt_f32_e32 vcc, " LZ ", %[hz]\n\t"                              \
                             "v_cmpx_gt_f32_e32 vcc, " HZ ", %[lz]\n\t"
                // stage (my id, ID) for the lanes in EXEC; FULL: label to leave through when the staging area is full
#define COL_STAGE(ID, FULL)                                                       \
                             "s_bcnt1_i32_b64 %[t0], exec\n\t"                                      \
                             "s_add_u32 %[t0], %[cnt], %[t0]\n\t"                                   \
                             "s_cmp_gt_u32 %[t0], 512\n\t"                /* CAPW */                \
                             "s_cbranch_scc1 " FULL "\n\t"                                          \
                             "v_mbcnt_lo_u32_b32 %[v0], exec_lo, 0\n\t"                             \
                             "v_mbcnt_hi_u32_b32 %[v0], exec_hi, %[v0]\n\t"                         \
                             "v_add_u32 %[v0], %[cnt], %[v0]\n\t"                                   \
                             "v_lshl_add_u32 %[v0], %[v0], 3, %[buf]\n\t"                           \
                             "v_mov_b32 %[v1], " ID "\n\t"                                          \
                             "ds_write2_b32 %[v0], %[qid], %[v1] offset1:1\n\t"                     \
                             "s_mov_b32 %[cnt], %[t0]\n\t"
                // candidate number K (0..3) of the group of four in flight: its registers, then "was it the last one?"
#define COL_CAND(K, LX, LY, LZ, HX, HY, HZ, ID)                                   \
                             COL_CMPX6(LX, LY, LZ, HX, HY, HZ)                                        \
                             "s_cbranch_execz 1" K "f\n\t"                                          \
                             COL_STAGE(ID, "2" K "f")                                                 \
                             "1" K ":\n\t"                                                          \
                             "s_mov_b64 exec, %[exec0]\n\t"                                         \
                             "s_cmp_eq_u32 %[bcnt], " K " + 1\n\t"                                  \
                             "s_cbranch_scc1 7f\n\t"
                // the way out when candidate K found the staging area full: the block goes on behind it after the flush
#define COL_CAND_FULL(K, ID)                                                      \
                             "2" K ":\n\t"                                                          \
                             "s_mov_b32 %[t0], " ID "\n\t"                                          \
                             "s_sub_u32 %[bcnt], %[bcnt], " K " + 1\n\t"                            \
                             "s_add_u32 %[boff], %[boff], 32 * (" K " + 1)\n\t"                     \
                             "s_branch 5f\n"
                asm volatile("s_mov_b64 %[exec0], exec\n\t"
                             "s_mov_b32 s43, %[idx]\n\t"
                             "s_mov_b32 %[bskip], %[idx]\n\t"
                             "s_cmp_lg_u32 %[bcnt], 0\n\t"                // inside a leaf block (the sink flushed): go on with it
                             "s_cbranch_scc1 6f\n"
                             "1:\n\t"
                             "s_lshl_b32 %[t0], s43, 5\n\t"                // s43: the node to fetch, then its skip link
                             "s_load_dwordx8 s[40:47], %[base], %[t0]\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             COL_CMPX6("s40", "s41", "s42", "s44", "s45", "s46")
                             "s_cbranch_execnz 2f\n\t"                     // somebody overlaps
                             "s_mov_b64 exec, %[exec0]\n"
                             "8:\n\t"
                             "s_cmp_lg_u32 s43, -1\n\t"                    // nobody (or the leaf is done): follow the skip link
                             "s_cbranch_scc1 1b\n"
                             "9:\n\t"
                             "s_mov_b32 %[idx], -1\n\t"
                             "s_mov_b64 %[hits], 0\n\t"                    // (EXEC is full again on every path that gets here)
                             "s_branch 4f\n"
                             "2:\n\t"
                             "s_cmp_ge_u32 %[t0], %[leaf]\n\t"
                             "s_cbranch_scc1 3f\n\t"
                             "s_mov_b64 exec, %[exec0]\n\t"
                             "s_bitcmp1_b32 s47, 31\n\t"                  // a leaf block?
                             "s_cbranch_scc1 30f\n\t"
                             "s_mov_b32 s43, s47\n\t"                      // an internal node: descend (down link) and go on
                             "s_branch 1b\n"
                             "3:\n\t"                                       // a leaf: stage (my id, its id) for the hit lanes
                             "s_mov_b32 %[bskip], s43\n\t"
                             COL_STAGE("s47", "39f")
                             "s_mov_b64 exec, %[exec0]\n\t"
                             "s_branch 8b\n"
                             "30:\n\t"                                      // a leaf block: count, offset of its first leaf record
                             "s_mov_b32 %[bskip], s43\n\t"
                             "s_and_b32 %[bcnt], s47, 15\n\t"
                             "s_add_u32 %[bcnt], %[bcnt], 1\n\t"
                             "s_bfe_u32 %[boff], s47, 0x1b0004\n\t"        // bits 4..30: the first leaf
                             "s_lshl_b32 %[boff], %[boff], 5\n\t"
                             "s_add_u32 %[boff], %[boff], %[leaf]\n"
                             "6:\n\t"                                       // candidates 0, 1 -> s[40:55], candidate 2 -> s[56:63]
                             "s_load_dwordx16 s[40:55], %[base], %[boff]\n\t"
                             "s_cmp_lt_u32 %[bcnt], 3\n\t"
                             "s_cbranch_scc1 31f\n\t"
                             "s_add_u32 %[t0], %[boff], 64\n\t"
                             "s_load_dwordx8 s[56:63], %[base], %[t0]\n"
                             "31:\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             COL_CAND("0", "s40", "s41", "s42", "s44", "s45", "s46", "s47")
                             COL_CAND("1", "s48", "s49", "s50", "s52", "s53", "s54", "s55")
                             "s_cmp_lt_u32 %[bcnt], 4\n\t"                 // candidates 3, 4 while 2 is compared
                             "s_cbranch_scc1 32f\n\t"
                             "s_add_u32 %[t0], %[boff], 96\n\t"
                             "s_load_dwordx16 s[40:55], %[base], %[t0]\n"
                             "32:\n\t"
                             COL_CAND("2", "s56", "s57", "s58", "s60", "s61", "s62", "s63")
                             "s_sub_u32 %[bcnt], %[bcnt], 3\n\t"
                             "s_add_u32 %[boff], %[boff], 96\n\t"
                             "s_cmp_lt_u32 %[bcnt], 3\n\t"                 // (candidates 0, 1 of the next group are in flight already)
                             "s_cbranch_scc1 31b\n\t"
                             "s_add_u32 %[t0], %[boff], 64\n\t"
                             "s_load_dwordx8 s[56:63], %[base], %[t0]\n\t"
                             "s_branch 31b\n"
                             "7:\n\t"                                       // the block is done: on along the marked node's skip link
                             "s_mov_b32 %[bcnt], 0\n\t"
                             "s_mov_b32 s43, %[bskip]\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"                   // (a load for candidates beyond the count may be in flight)
                             "s_cmp_lg_u32 s43, -1\n\t"
                             "s_cbranch_scc1 1b\n\t"
                             "s_branch 9b\n"
                             COL_CAND_FULL("0", "s47") COL_CAND_FULL("1", "s55") COL_CAND_FULL("2", "s63")
                             "39:\n\t"
                             "s_mov_b32 %[t0], s47\n"
                             "5:\n\t"                                       // the staging area is full: let the sink flush it
                             "s_mov_b64 %[hits], exec\n\t"
                             "s_mov_b64 exec, %[exec0]\n\t"
                             "s_mov_b32 %[idx], %[bskip]\n\t"             // where the walk goes on (behind the leaf / the block)
                             "s_waitcnt lgkmcnt(0)\n"                      // (prefetched candidates may still be in flight)
                             "4:"
                             : [idx] "+s"(idx), [cnt] "+s"(sink.count), [bcnt] "+s"(bcnt), [boff] "+s"(boff), [hits] "=&s"(hits), [exec0] "=&s"(exec0),
                               [t0] "=&s"(pid), [bskip] "=&s"(bskip), [v0] "=&v"(v0), [v1] "=&v"(v1)
                             : [base] "s"(rows_b), [leaf] "s"(leaf_start * 32u), [buf] "s"(buf_lds),
                               [qid] "v"(qid), [hx] "v"(hx), [hy] "v"(hy), [hz] "v"(hz), [lx] "v"(lx), [ly] "v"(ly), [lz] "v"(lz)
                             : "vcc", "scc", "memory", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",
                               "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63");
#undef COL_CAND_FULL
#undef COL_CAND
#undef COL_STAGE
#undef COL_CMPX6
                if (!hits) break;                                           // the chain ended (idx == END, no block pending)
                sink.emit(hits, qid, pid);                                  // the staging area was full: flush, then stage;
            }                                                               // then on: the rest of the block (bcnt), the skip link (idx)
        } else if constexpr (sizeof(T) == 8 && WALK == 1 && !VEC) {
            // The asm walk for float64 records (round 4): a record is 64 bytes -- {min.xyz, skip}{max.xyz, down} as doubles, the
            // links in the low words of the fourth lanes -- fetched by ONE s_load_dwordx16 into s[40:55] at a 32-bit offset
            // (node << 6; arrays below 4 GB), tested with six v_cmpx_*_f64 against the lanes' boxes (SGPR pair against VGPR
            // pair), and the whole walk stays inside one asm loop as for float32: a miss follows the skip link (s46), a hit
            // internal node the down link (s54), a hit leaf stages its pairs right there.  Marked nodes (leaf blocks) are entered
            // through their first leaf and the leaves' skip links lead through the block (exact: col_common.h); a candidate
            // loop like float32's would need three 64-byte records in flight, which the 80-SGPR budget does not hold.
            const char *rows_b = reinterpret_cast<const char *>(rows);
            const u32 buf_lds = (u32)(uintptr_t)(__attribute__((address_space(3))) void *)sink.buf;
            while (idx != END) {
                u64 hits, exec0;
                u32 pid, v0, v1;
                asm volatile("s_mov_b64 %[exec0], exec\n\t"
                             "s_mov_b32 s46, %[idx]\n"
                             "1:\n\t"
                             "s_lshl_b32 %[t0], s46, 6\n\t"                // s46: the node to fetch, then its skip link
                             "s_load_dwordx16 s[40:55], %[base], %[t0]\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "v_cmpx_lt_f64_e32 vcc, s[40:41], %[hx]\n\t"   // lo.x < my hi.x
                             "v_cmpx_gt_f64_e32 vcc, s[48:49], %[lx]\n\t"   // hi.x > my lo.x
                             "v_cmpx_lt_f64_e32 vcc, s[42:43], %[hy]\n\t"
                             "v_cmpx_gt_f64_e32 vcc, s[50:51], %[ly]\n\t"
                             "v_cmpx_lt_f64_e32 vcc, s[44:45], %[hz]\n\t"
                             "v_cmpx_gt_f64_e32 vcc, s[52:53], %[lz]\n\t"
                             "s_cbranch_execnz 2f\n\t"                     // somebody overlaps
                             "s_mov_b64 exec, %[exec0]\n"
                             "8:\n\t"
                             "s_cmp_lg_u32 s46, -1\n\t"                    // nobody (or the leaf is done): follow the skip link
                             "s_cbranch_scc1 1b\n\t"
                             "s_mov_b32 %[idx], -1\n\t"
                             "s_mov_b64 %[hits], 0\n\t"
                             "s_branch 4f\n"
                             "2:\n\t"
                             "s_cmp_ge_u32 %[t0], %[leaf]\n\t"
                             "s_cbranch_scc1 3f\n\t"
                             "s_mov_b64 exec, %[exec0]\n\t"
                             "s_mov_b32 s46, s54\n\t"                      // an internal node: descend (down link) ...
                             "s_bitcmp1_b32 s54, 31\n\t"
                             "s_cbranch_scc0 1b\n\t"
                             "s_bfe_u32 s46, s54, 0x1b0004\n\t"            // ... or, a leaf block: on through its leaf chain
                             "s_add_u32 s46, s46, %[leafidx]\n\t"
                             "s_branch 1b\n"
                             "3:\n\t"                                       // a leaf: stage (my id, its id) for the hit lanes
                             "s_bcnt1_i32_b64 %[t0], exec\n\t"
                             "s_add_u32 %[t0], %[cnt], %[t0]\n\t"
                             "s_cmp_gt_u32 %[t0], 512\n\t"                 // CAPW
                             "s_cbranch_scc1 5f\n\t"
                             "v_mbcnt_lo_u32_b32 %[v0], exec_lo, 0\n\t"
                             "v_mbcnt_hi_u32_b32 %[v0], exec_hi, %[v0]\n\t"
                             "v_add_u32 %[v0], %[cnt], %[v0]\n\t"
                             "v_lshl_add_u32 %[v0], %[v0], 3, %[buf]\n\t"
                             "v_mov_b32 %[v1], s54\n\t"
                             "ds_write2_b32 %[v0], %[qid], %[v1] offset1:1\n\t"
                             "s_mov_b32 %[cnt], %[t0]\n\t"
                             "s_mov_b64 exec, %[exec0]\n\t"
                             "s_branch 8b\n"
                             "5:\n\t"                                       // the staging area is full: let the sink flush it
                             "s_mov_b64 %[hits], exec\n\t"
                             "s_mov_b64 exec, %[exec0]\n\t"
                             "s_mov_b32 %[t0], s54\n\t"                    // the hit leaf's id
                             "s_mov_b32 %[idx], s46\n"                      // where the walk goes on: behind the leaf
                             "4:"
                             : [idx] "+s"(idx), [cnt] "+s"(sink.count), [hits] "=&s"(hits), [exec0] "=&s"(exec0),
                               [t0] "=&s"(pid), [v0] "=&v"(v0), [v1] "=&v"(v1)
                             : [base] "s"(rows_b), [leaf] "s"(leaf_start * 64u), [leafidx] "s"(leaf_start), [buf] "s"(buf_lds),
                               [qid] "v"(qid), [hx] "v"(hx), [hy] "v"(hy), [hz] "v"(hz), [lx] "v"(lx), [ly] "v"(ly), [lz] "v"(lz)
                             : "vcc", "scc", "memory", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",
                               "s51", "s52", "s53", "s54", "s55");
                if (!hits) break;                                           // the chain ended
                sink.emit(hits, qid, pid);                                  // the staging area was full: flush, then stage; then on
            }
        } else if constexpr (sizeof(T) == 4 && WALK == 2 && !VEC) {
            typedef int v8i __attribute__((ext_vector_type(8)));
            const char *rows_b = reinterpret_cast<const char *>(rows);
            const u32 buf_lds = (u32)(uintptr_t)(__attribute__((address_space(3))) void *)sink.buf;
            while (idx != END) {
                u64 hits, exec0;
                u32 off, t0, v0, v1;
                v8i r;
                asm volatile("s_mov_b64 %[exec0], exec\n\t"
                             "s_mov_b32 s67, %[idx]\n"
                             "1:\n\t"
                             "s_lshl_b32 %[off], s67, 5\n\t"
                             "s_load_dwordx8 s[64:71], %[base], %[off]\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "v_cmpx_lt_f32_e32 vcc, s64, %[hx]\n\t"
                             "v_cmpx_gt_f32_e32 vcc, s68, %[lx]\n\t"
                             "v_cmpx_lt_f32_e32 vcc, s65, %[hy]\n\t"
                             "v_cmpx_gt_f32_e32 vcc, s69, %[ly]\n\t"
                             "v_cmpx_lt_f32_e32 vcc, s66, %[hz]\n\t"
                             "v_cmpx_gt_f32_e32 vcc, s70, %[lz]\n\t"
                             "s_cbranch_execnz 2f\n\t"
                             "s_mov_b64 exec, %[exec0]\n\t"
                             "s_cmp_lg_u32 s67, -1\n\t"
                             "s_cbranch_scc1 1b\n\t"
                             "s_mov_b32 %[idx], -1\n\t"
                             "s_mov_b64 %[hits], 0\n\t"
                             "s_branch 4f\n"
                             "2:\n\t"
                             "s_cmp_ge_u32 %[off], %[leaf]\n\t"
                             "s_cbranch_scc1 3f\n\t"
                             "s_mov_b64 exec, %[exec0]\n\t"
                             "s_mov_b32 s67, s71\n\t"                      // descend (down link) ...
                             "s_bitcmp1_b32 s71, 31\n\t"
                             "s_cbranch_scc0 1b\n\t"
                             "s_bfe_u32 s67, s71, 0x1b0004\n\t"            // ... or, a leaf block: on through its leaf chain
                             "s_add_u32 s67, s67, %[leafidx]\n\t"
                             "s_branch 1b\n"
                             "3:\n\t"
                             "s_bcnt1_i32_b64 %[t0], exec\n\t"
                             "s_add_u32 %[t0], %[cnt], %[t0]\n\t"
                             "s_cmp_gt_u32 %[t0], %[capw]\n\t"
                             "s_cbranch_scc1 5f\n\t"
                             "v_mbcnt_lo_u32_b32 %[v0], exec_lo, 0\n\t"
                             "v_mbcnt_hi_u32_b32 %[v0], exec_hi, %[v0]\n\t"
                             "v_add_u32 %[v0], %[cnt], %[v0]\n\t"
                             "v_lshl_add_u32 %[v0], %[v0], 3, %[buf]\n\t"
                             "v_mov_b32 %[v1], s71\n\t"
                             "ds_write2_b32 %[v0], %[qid], %[v1] offset1:1\n\t"
                             "s_mov_b32 %[cnt], %[t0]\n\t"
                             "s_mov_b64 exec, %[exec0]\n\t"
                             "s_cmp_lg_u32 s67, -1\n\t"
                             "s_cbranch_scc1 1b\n\t"
                             "s_mov_b32 %[idx], -1\n\t"
                             "s_mov_b64 %[hits], 0\n\t"
                             "s_branch 4f\n"
                             "5:\n\t"
                             "s_mov_b64 %[hits], exec\n\t"
                             "s_mov_b64 exec, %[exec0]\n"
                             "4:"
                             : [idx] "+s"(idx), [cnt] "+s"(sink.count), [hits] "=s"(hits), [off] "=&s"(off), [t0] "=&s"(t0),
                               [v0] "=&v"(v0), [v1] "=&v"(v1), [exec0] "=&s"(exec0), "=&{s[64:71]}"(r)
                             : [base] "s"(rows_b), [leaf] "s"(leaf_start * 32u), [leafidx] "s"(leaf_start), [capw] "s"((u32)CAPW),
                               [buf] "s"(buf_lds), [qid] "v"(qid), [hx] "v"(hx), [hy] "v"(hy), [hz] "v"(hz), [lx] "v"(lx),
                               [ly] "v"(ly), [lz] "v"(lz)
                             : "vcc", "scc", "memory");
                if (!hits) break;
                sink.emit(hits, qid, (u32)r[7]);
                idx = (u32)r[3];
            }
        } else {
            auto test = [&](const V4 &a, const V4 &b) -> u64 {
                return __builtin_amdgcn_ballot_w64(hx > a.x) & __builtin_amdgcn_ballot_w64(lx < b.x) &
                       __builtin_amdgcn_ballot_w64(hy > a.y) & __builtin_amdgcn_ballot_w64(ly < b.y) &
                       __builtin_amdgcn_ballot_w64(hz > a.z) & __builtin_amdgcn_ballot_w64(lz < b.z);
            };
            while (idx != END) {
                u32 li = idx;
                if (VEC) asm volatile("" : "+v"(li));               // vector load at a uniform address
                const V4 a = rows[2ull * li], b = rows[2ull * li + 1];
                const u32 skip = (u32) * reinterpret_cast<const Bits *>(&a.w);
                const u32 down = (u32) * reinterpret_cast<const Bits *>(&b.w);
                const u64 hits = test(a, b);
                const bool is_leaf = idx >= leaf_start;
                if (STATS) {
                    trips++; leaf_tests += is_leaf;
                    // how many steps stay within W sorted positions of the block's first leaf (W = 1k, 2k, 4k, 8k)
                    const u32 p2 = is_leaf ? idx - leaf_start : idx, b0 = (packet & ~(u32)(TW - 1)) * 64;
                    for (int k = 0; k < 4; k++) win[k] += p2 >= b0 && p2 < b0 + (1024u << k);
                }
                u32 next = skip;
                if (hits) {
                    if (is_leaf) { sink.emit(hits, qid, down); if (STATS) leaf_hits++; }
                    else { next = descend_link(down, leaf_start, marks); if (STATS) descents++; }      // (a leaf block: on through its leaves)
                }
                idx = (u32)__builtin_amdgcn_readfirstlane((int)next);
            }
        }
        if (ordered) {
            // (the walk's time goes to cost[] through a wave-private LDS list, eight walks per store instruction and the rest when the
            // wave leaves: a store per walk put its acknowledgement -- vmcnt counts stores too -- in front of the next packet's
            // record loads: + 7 % at 2 M spheres, + 16 % at 16 M)
            const u32 k = s_ncost[w];
            if (lane == 0) {
                s_costbuf[w][k] = make_uint2(packet, (u32)__builtin_amdgcn_s_memtime() - s_walk_t0[w]);
                s_ncost[w] = k == 7u ? 0u : k + 1u;
            }
            if (k == 7u && lane < 8) { const uint2 e = s_costbuf[w][lane]; walk_order[e.x] = e.y; }
            __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (PROF) {
            const u64 prof_t3 = wall_clock64();
            if (lane == 0 && packet < COL_PROF_PACKETS && !only1)
                g_walk_prof[packet] = make_uint4((u32)(prof_t2 - prof_t1), (u32)(prof_t3 - prof_t2), (u32)(prof_t1 - prof_t0), (u32)prof_t0);
        }
    }

    if (dyn && walk_order) {
        const u32 k = s_ncost[w];
        if (lane < k) { const uint2 e = s_costbuf[w][lane]; walk_order[e.x] = e.y; }
    }
    if (!STATS && sink.chunk) {
        // chunked allocation: the leftovers go the same way; then the block records what it leaves unused of its last
        // chunk, and the LAST block to get here sums up: T = slots allocated - holes = the number of pairs
        if (sink.count) sink.flush();
        __syncthreads();
        ChunkHdr *hdr = reinterpret_cast<ChunkHdr *>(stats);
        __shared__ u32 s_last;
        if (threadIdx.x == 0) {
            const unsigned long long v = s_chunk;
            const u32 cur = (u32)v, end = (u32)(v >> 32);
            hdr->holes[blockIdx.x] = make_uint2(cur, end - cur);
            __threadfence();
            s_last = atomicAdd(&hdr->done, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            u32 sum = 0;
            for (u32 i = threadIdx.x; i < gridDim.x; i += TT) sum += __hip_atomic_load(&hdr->holes[i].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sum = wave_sum(sum);
            if (lane == 0) s_cnt[w] = sum;
            __syncthreads();
            if (threadIdx.x == 0) {
                u32 holes = 0;
                for (int i = 0; i < TW; i++) holes += s_cnt[i];
                const u32 A = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u32 over = capacity > 0 && (A > capacity || A > (1u << 29)) ? 1u : 0u;      // (count-only: nothing is stored, nothing to close)
                hdr->A = A;
                hdr->T = A - holes;
                hdr->overflow = over;
                hdr->done = 0;
                *counter = over ? 0u : A - holes;         // the exact walk counts again from zero
            }
        }
        return;
    }
    // block-level flush of what is still staged: one atomic for the 16 waves
    if (lane == 0) s_cnt[w] = sink.count;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 total = 0;
        for (int i = 0; i < TW; i++) { const u32 c = s_cnt[i]; s_cnt[i] = total; total += c; }
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    sink.copy_out(s_base + s_cnt[w], sink.count);
    if (STATS) {                       // diagnostics: phase-2 steps, one atomic per block
        __shared__ unsigned long long s_st[8];
        if (threadIdx.x < 8) s_st[threadIdx.x] = 0;
        __syncthreads();
        if (lane == 0) {
            atomicAdd(&s_st[0], (unsigned long long)trips);
            atomicAdd(&s_st[1], (unsigned long long)descents);
            atomicAdd(&s_st[2], (unsigned long long)leaf_tests);
            atomicAdd(&s_st[3], (unsigned long long)leaf_hits);
            for (int k = 0; k < 4; k++) atomicAdd(&s_st[4 + k], (unsigned long long)win[k]);
        }
        __syncthreads();
        if (threadIdx.x < 8) atomicAdd(&stats[threadIdx.x], (u64)s_st[threadIdx.x]);
    }
}

// Lane-per-query variant: every lane walks its own skip chain (12 visits per query on the uniform
// scene).  Divergent, but it has no intra-wave phase; kept for sparse scenes, selected by
// col_traverse's `variant` heuristics and for A/B measurements.
template <typename T>
__global__ __launch_bounds__(TT) void k_traverse_lane(u32 *__restrict__ pairs, u32 *__restrict__ counter, u32 capacity,
                                                       const T *__restrict__ bounds, u32 n, int g_skip) {
    typedef typename BTypes<T>::V4 V4;
    typedef typename BTypes<T>::Bits Bits;
    __shared__ uint2 s_buf[TW][CAPW];
    __shared__ u32 s_cnt[TW];
    __shared__ u32 s_base;
    const u32 lane = lane_id(), w = threadIdx.x / 64;
    const u32 leaf_start = n - 1;
    const bool marks = n <= COL_LEAF_BLOCK_MAX_N;
    const V4 *rows = reinterpret_cast<const V4 *>(bounds);
    PairSink sink = {s_buf[w], 0u, pairs, counter, capacity, lane, nullptr, nullptr};
    const u32 npackets = (n + 63) / 64;
    for (u32 packet = blockIdx.x * TW + w; packet < npackets; packet += gridDim.x * TW) {
        const u32 q = packet * 64 + lane;
        T lx = (T)INFINITY, ly = lx, lz = lx, hx = -lx, hy = -lx, hz = -lx;
        u32 qid = 0, idx = END;
        if (q < n) {
            const V4 a = rows[2ull * (leaf_start + q)], b = rows[2ull * (leaf_start + q) + 1];
            lx = a.x; ly = a.y; lz = a.z; hx = b.x; hy = b.y; hz = b.z;
            idx = (u32) * reinterpret_cast<const Bits *>(&a.w);
            qid = (u32) * reinterpret_cast<const Bits *>(&b.w);
        }
        if (g_skip) idx = END;
        while (__ballot(idx != END)) {              // wave-uniform loop so that emit() stays convergent
            bool hit = false;
            u32 down = 0;
            if (idx != END) {
                const V4 a = rows[2ull * idx], b = rows[2ull * idx + 1];
                const u32 skip = (u32) * reinterpret_cast<const Bits *>(&a.w);
                down = (u32) * reinterpret_cast<const Bits *>(&b.w);
                const bool overlap = hx > a.x && lx < b.x && hy > a.y && ly < b.y && hz > a.z && lz < b.z;
                const bool leaf = idx >= leaf_start;
                hit = overlap && leaf;
                idx = (overlap && !leaf) ? descend_link(down, leaf_start, marks) : skip;
            }
            const u64 hits = __ballot(hit);
            if (hits) sink.emit(hits, qid, down);
        }
    }
    if (lane == 0) s_cnt[w] = sink.count;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 total = 0;
        for (int i = 0; i < TW; i++) { const u32 c = s_cnt[i]; s_cnt[i] = total; total += c; }
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    sink.copy_out(s_base + s_cnt[w], sink.count);
}

int g_traverse_variant = 0;       // diagnostics switch, see col_debug_traverse
// from this many spheres on col_collide's traversal takes its work in dynamic order (k_traverse), and up to
// COL_SPLIT_UNITS_UPTO a packet is two units (its walk, then its phase 1).  Whole path, one box, static / dynamic / dynamic
// with split units (tools/dyn_ab.py): 0.5 M 0.119 / 0.123 / 0.119 ms, 1 M 0.171-0.177 / 0.172-0.174 / 0.168-0.169,
// 2 M 0.302-0.312 / 0.292-0.294 / 0.291-0.292, 4 M 0.624-0.631 / 0.590 / 0.604-0.606, 16 M 2.44 / 2.27 / 2.39; config 3
// (chunked allocation) 0.5 M 0.311 / 0.314 / 0.297, 1 M 0.602-0.609 / 0.585-0.589 / 0.558-0.566, 2 M 1.47 / 1.39 / 1.34
#define COL_DYNAMIC_PACKETS_FROM 400000u
#define COL_SPLIT_UNITS_UPTO 3000000u

template <typename T>
int launch_traverse(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const void *bounds,
                    uint32_t n, uint64_t *stats, int mode, uint32_t *sched = nullptr, const uint32_t *n_dev = nullptr,
                    uint32_t *walk_order = nullptr) {
    const u32 npackets = (n + 63) / 64;
    u32 blocks = (u32)col_ceil_div(npackets, TW);
    if (blocks > 512) blocks = 512;               // 2 resident blocks of 16 waves per CU, grid-stride beyond
    dim3 g(blocks), t(TT);
    hipStream_t s = col_stream(stream);
    const T *bd = (const T *)bounds;
    u64 *st = (u64 *)stats;
    // Measured on MI355X, 1 M spheres (tools/trav_ab.py): packet walk vs lane-per-query 0.109 vs 0.120 ms on
    // the uniform scene, 0.78 vs 1.04 ms on a clustered one (11 M pairs).  Walking K = 2 / 4 packets per
    // wave in lock step was slower (0.134 / 0.169 ms) and is gone.  Halving the resident waves (variant
    // bit 2: 256 blocks, 4 waves/SIMD) costs +35 % (0.102 -> 0.138 ms), i.e. T ~ 66 us + 288 us /
    // waves-per-SIMD: about a third of the walk is exposed latency of its dependent record loads at the
    // hardware's 8 waves/SIMD, the rest is instruction issue.  Vector instead of scalar record loads
    // (variant bit 1): 0.112 ms.
    if (g_traverse_variant & 4) g = dim3(blocks > 256 ? 256 : blocks);
    if (g_traverse_variant & 16) mode |= 8;       // plain packet order
    if (g_traverse_variant & 256) mode |= 1;      // timing ablation: skip phase 1 (pairs inside the packets)
    if (g_traverse_variant & 512) mode |= 2;      // timing ablation: skip phase 2 (the walk)
    if (g_traverse_variant & 16777216) mode |= 2048;      // timing experiment: the walk for lanes 0..31 only
    if (g_traverse_variant & 33554432) mode |= 4096;      // ... for lanes 32..63 only
    // (diagnostics buffers: one per device, allocated on first use and kept; variant bits 10 and 13 only)
    constexpr int MAX_DEV = 16;
    static u64 *spread_dev[MAX_DEV] = {nullptr};   // timing ablation: per-wave flush counters (variant bit 10)
    static u32 *dyn_sched_dev[MAX_DEV] = {nullptr};
    int dev = 0;
    if (g_traverse_variant & (1024 | 8192)) {
        COL_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= MAX_DEV) return COL_EINVAL;
    }
    u64 *&spread = spread_dev[dev];
    if (g_traverse_variant & 1024) {
        if (!spread && hipMalloc((void **)&spread, 8192 * 32 * 4) != hipSuccess) return COL_EINVAL;
        mode |= 16;
    }
    // the asm walk needs 32-bit record offsets (the record array below 4 GB); variant bit 6 forces the generic loop
    // ... and 64-byte aligned records: its leaf-block test fetches candidates with 64-byte scalar loads, which may read up to 32
    // bytes past the last record -- inside the allocation granule of an aligned array, possibly across a mapping boundary of
    // a 32-byte aligned sub-allocation (include/collision_hip.h: alignment of `bounds`); anything else takes the generic loop
    const bool off32 = (2ull * n - 1) * 8 * sizeof(T) < (1ull << 32) && !(g_traverse_variant & 64) && ((uintptr_t)bounds & 63) == 0;
    // dynamic packet order (k_traverse, mode bit 8): `sched` = eight zeroed words.  Variant bit 13 forces it (with
    // counters of its own, cleared by a memset), bit 14 forbids it: the A/B switches of tools/walk3_ab.py
    u32 *&dyn_sched = dyn_sched_dev[dev];
    if ((g_traverse_variant & 8192) && !sched && !st && !(mode & (32 | 64))) {
        if (!dyn_sched && hipMalloc((void **)&dyn_sched, 64) != hipSuccess) return COL_EINVAL;
        COL_HIP(hipMemsetAsync(dyn_sched, 0, 64, s));
        sched = dyn_sched;
    }
    if (sched && !st && !(g_traverse_variant & (16384 | 1 | 2 | 4 | 128 | 1024))) {      // (the other walks are A/B material)
        mode |= 256;
        if (g_traverse_variant & 4194304) mode |= 128;      // A/B: the first batch drawn from the counter too
        if (g_traverse_variant & 8388608) mode |= 1024;     // A/B: long walks at the ordinary priority
        if ((n <= COL_SPLIT_UNITS_UPTO || (g_traverse_variant & 65536)) && !(g_traverse_variant & 131072)) mode |= 512;      // split units
        if (off32 && (g_traverse_variant & 4096)) k_traverse<T, false, false, 1, false, true><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, (u64 *)sched, mode, NoGhost{}, n_dev, (g_traverse_variant & 2097152) ? nullptr : walk_order);
        else if (off32) k_traverse<T, false, false, 1><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, (u64 *)sched, mode, NoGhost{}, n_dev, (g_traverse_variant & 2097152) ? nullptr : walk_order);
        else k_traverse<T, false, false, 0><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, (u64 *)sched, mode, NoGhost{}, n_dev, (g_traverse_variant & 2097152) ? nullptr : walk_order);
        COL_LAUNCH_OK();
        return COL_OK;
    }
    if ((g_traverse_variant & 255) == 1) k_traverse_lane<T><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, 0);
    else if (st) k_traverse<T, true, false, 0><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, st, mode);
    else if (g_traverse_variant & 2) k_traverse<T, false, true, 0><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, st, mode);
    else if (off32 && (g_traverse_variant & 128)) k_traverse<T, false, false, 2><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, st, mode);
    else if (off32 && (g_traverse_variant & 4096)) k_traverse<T, false, false, 1, false, true><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, st, mode, NoGhost{}, n_dev);
    else if (off32) k_traverse<T, false, false, 1><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, (g_traverse_variant & 1024) ? spread : st, mode, NoGhost{}, n_dev);
    else k_traverse<T, false, false, 0><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, st, mode, NoGhost{}, n_dev);
    COL_LAUNCH_OK();
    return COL_OK;
}

// Closes the holes a chunked walk left in the pair list (ChunkHdr): the list occupies [0, A) with one hole per block --
// the unused end of its last chunk -- and T = A - (all holes) pairs; the pairs that lie at or beyond T move into the hole
// slots below T (there are exactly as many of one as of the other), which leaves the dense list [0, T).  Every block sorts
// the <= 512 holes by position for itself (rank by counting), then thread k moves the k-th pair.
__global__ __launch_bounds__(256) void k_pairs_compact(u32 *__restrict__ pairs, const ChunkHdr *__restrict__ hdr, u32 nholes) {
    __shared__ u32 s_start[512], s_len[512];
    __shared__ u32 s_low[513], s_fill[513];       // exclusive prefixes: hole slots below T; pairs at / beyond T before hole i
    __shared__ u32 s_b[513], s_e[513];
    if (hdr->overflow) return;
    const u32 A = hdr->A, T = hdr->T, tid = threadIdx.x;
    if (A == T) return;                            // no holes
    // sort the holes by position without sorting: every chunk starts at a multiple of CHUNK_PAIRS (the counter only ever
    // grows by whole chunks) and holds at most one hole, so a bitmap over the chunk numbers ranks them (popcounts)
    __shared__ unsigned long long s_bits[1024];          // 65 536 chunks = 2^29 pairs; longer lists count as overflow
    __shared__ u32 s_wpre[1024];
    __shared__ u32 s_ws2[4];
    for (u32 i = tid; i < 1024; i += 256) s_bits[i] = 0;
    for (u32 i = tid; i < 513; i += 256) { if (i < 512) { s_start[i] = A; s_len[i] = 0; } }
    __syncthreads();
    uint2 mine[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const u32 i = tid + 256 * u;
        mine[u] = i < nholes ? hdr->holes[i] : make_uint2(0u, 0u);
        if (mine[u].y) {
            const u32 c = mine[u].x / (u32)CHUNK_PAIRS;
            atomicOr(&s_bits[c >> 6], 1ull << (c & 63u));
        }
    }
    __syncthreads();
    {
        u32 pc[4], sum = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) { pc[u] = (u32)__popcll(s_bits[4 * tid + u]); sum += pc[u]; }
        u32 total;
        u32 run = block_excl_scan<256>(sum, s_ws2, &total);
#pragma unroll
        for (int u = 0; u < 4; u++) { s_wpre[4 * tid + u] = run; run += pc[u]; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++)
        if (mine[u].y) {
            const u32 c = mine[u].x / (u32)CHUNK_PAIRS;
            const u32 rank = s_wpre[c >> 6] + (u32)__popcll(s_bits[c >> 6] & ((1ull << (c & 63u)) - 1ull));
            s_start[rank] = mine[u].x; s_len[rank] = mine[u].y;
        }
    __syncthreads();
    {   // exclusive prefixes over the 512 sorted holes, two per thread
        __shared__ u32 s_ws[4];
        u32 lowc[2], highc[2], bb[2], ee[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const u32 i = 2 * tid + u;
            const u32 st = s_start[i], ln = s_len[i];
            bb[u] = ln ? (st > T ? st : T) : A;                          // the hole's part at / beyond T: [b, e)
            ee[u] = ln ? (st + ln > T ? st + ln : T) : A;
            lowc[u] = ln ? (st >= T ? 0u : (st + ln <= T ? ln : T - st)) : 0u;      // its slots below T
            highc[u] = ee[u] - bb[u];
        }
        u32 total_low, total_high;
        const u32 plow = block_excl_scan<256>(lowc[0] + lowc[1], s_ws, &total_low);
        const u32 phigh = block_excl_scan<256>(highc[0] + highc[1], s_ws, &total_high);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const u32 i = 2 * tid + u;
            const u32 lo_before = plow + (u ? lowc[0] : 0u), hi_before = phigh + (u ? highc[0] : 0u);
            s_low[i] = lo_before;
            s_b[i] = bb[u]; s_e[i] = ee[u];
            s_fill[i] = (bb[u] - T) - hi_before;                          // pairs in [T, b)
        }
        if (tid == 0) {
            s_low[512] = total_low;
            s_fill[512] = (A - T) - total_high;                           // = the number of pairs to move = total_low
            s_b[512] = A; s_e[512] = A;
        }
    }
    __syncthreads();
    const u32 moves = s_low[512];
    // Four moves per thread and round, their searches side by side: both are nine steps over 512 sorted entries, written without
    // branches (step 256, 128, .. 1: the candidate index never passes 511), so that the 4 x 18 dependent LDS reads of a round overlap
    // and its four pairs are on their way together (one move at a time: 18.5 us for config 3's 2 M moves; R4.20).
    constexpr int MV = 4;
    const u32 stride = gridDim.x * 256;
    for (u32 k0 = blockIdx.x * 256 + tid; k0 < moves; k0 += MV * stride) {
        u32 kk[MV], lo1[MV], lo2[MV];
#pragma unroll
        for (int u = 0; u < MV; u++) { kk[u] = min(k0 + u * stride, moves - 1u); lo1[u] = 0; lo2[u] = 0; }
#pragma unroll
        for (u32 step = 256; step; step >>= 1) {
#pragma unroll
            for (int u = 0; u < MV; u++) {
                if (s_low[lo1[u] + step] <= kk[u]) lo1[u] += step;       // -> the last i with s_low[i] <= k (s_low[0] = 0)
                if (s_fill[lo2[u] + step] <= kk[u]) lo2[u] += step;      // -> the last i with s_fill[i] <= k, if s_fill[0] <= k
            }
        }
        uint2 pr[MV];
        u32 dst[MV];
#pragma unroll
        for (int u = 0; u < MV; u++) {
            // destination: the k-th hole slot below T; source: the k-th pair at / beyond T -- behind hole lo2, or before the first hole
            dst[u] = s_start[lo1[u]] + (kk[u] - s_low[lo1[u]]);
            const u32 src = kk[u] < s_fill[0] ? T + kk[u] : s_e[lo2[u]] + (kk[u] - s_fill[lo2[u]]);
            pr[u] = *reinterpret_cast<const uint2 *>(pairs + 2ull * src);
        }
#pragma unroll
        for (int u = 0; u < MV; u++)
            if (k0 + u * stride < moves) *reinterpret_cast<uint2 *>(pairs + 2ull * dst[u]) = pr[u];
    }
}

// The walk with chunked pair allocation (dense scenes), for lists that start at *counter == 0: four launches -- clear the
// header, walk, close the holes, and the exact walk again, which returns at once unless the chunks ran past `capacity`
// (the list could then not be closed and is rebuilt the exact way: still min(count, capacity) valid pairs).
template <typename T>
int launch_traverse_chunked(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const void *bounds,
                            uint32_t n, void *scratch, const uint32_t *n_dev = nullptr, uint32_t *walk_order = nullptr,
                            bool hdr_cleared = false) {
    // (dynamic packet order, its counters in the header's pad: always -- a dense scene's packets differ by 5 x in time)
    const bool split = (n <= COL_SPLIT_UNITS_UPTO || (g_traverse_variant & 65536)) && !(g_traverse_variant & 131072);
    const int dyn = (g_traverse_variant & 16384) ? 0 : (split ? 256 | 512 : 256) | ((g_traverse_variant & 4194304) ? 128 : 0) | ((g_traverse_variant & 8388608) ? 1024 : 0);
    const u32 npackets = (n + 63) / 64;
    u32 blocks = (u32)col_ceil_div(npackets, TW);
    if (blocks > 512) blocks = 512;
    dim3 g(blocks), t(TT);
    hipStream_t s = col_stream(stream);
    const T *bd = (const T *)bounds;
    ChunkHdr *hdr = (ChunkHdr *)scratch;
    if (!hdr_cleared) COL_HIP(hipMemsetAsync(hdr, 0, 64, s));        // (the whole path: the tree build's last kernel has cleared it)
    k_traverse<T, false, false, 1><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, (u64 *)hdr, 32 | dyn, NoGhost{}, n_dev,
                                                   (dyn & 256) && !(g_traverse_variant & 2097152) ? walk_order : nullptr);
    COL_LAUNCH_OK();
    if (capacity) {
        k_pairs_compact<<<dim3(512), dim3(256), 0, s>>>(pairs, hdr, blocks);
        COL_LAUNCH_OK();
        k_traverse<T, false, false, 1><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, (u64 *)hdr, 64, NoGhost{}, n_dev);
        COL_LAUNCH_OK();
    }
    return COL_OK;
}

template <typename T>
int launch_ghost(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const void *bounds, uint32_t n,
                 const GhostArgs &ga, uint32_t max_ghosts, const uint32_t *n_dev) {
    const u32 npackets = (max_ghosts + 63) / 64;
    u32 blocks = (u32)col_ceil_div(npackets, TW);
    if (blocks > 512) blocks = 512;
    if (blocks == 0) return COL_OK;
    dim3 g(blocks), t(TT);
    hipStream_t s = col_stream(stream);
    const T *bd = (const T *)bounds;
    const bool off32 = (2ull * n - 1) * 8 * sizeof(T) < (1ull << 32) && !(g_traverse_variant & 64) && ((uintptr_t)bounds & 63) == 0;
    if (off32) k_traverse<T, false, false, 1, true><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, nullptr, 0, ga, n_dev);
    else k_traverse<T, false, false, 0, true><<<g, t, 0, s>>>(pairs, counter, capacity, bd, n, nullptr, 0, ga, n_dev);
    COL_LAUNCH_OK();
    return COL_OK;
}

}  // namespace

// ghost spheres as packets of 64 Morton-sorted queries against the tree in `bounds` (see GhostArgs); internal, called
// by col_traverse_ghost_slots (multi.hip).  `count` (device) = the number of sorted ghosts, at most max_ghosts.
extern "C" int col_traverse_ghost_packets(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const void *bounds,
                                          uint32_t n, int coord_bytes, const uint32_t *rec, const uint32_t *order,
                                          const uint32_t *count, uint32_t max_ghosts, const uint32_t *local_gids,
                                          const uint32_t *n_dev) {
    if (n == 0 || max_ghosts == 0) return COL_OK;
    const GhostArgs ga = {rec, order, count, local_gids};
    if (coord_bytes == 4) return launch_ghost<float>(stream, pairs, counter, capacity, bounds, n, ga, max_ghosts, n_dev);
    if (coord_bytes == 8) return launch_ghost<double>(stream, pairs, counter, capacity, bounds, n, ga, max_ghosts, n_dev);
    return COL_EINVAL;
}

extern "C" {

void col_debug_traverse(int variant) { g_traverse_variant = variant; }
// diagnostics: the per-packet records of the profiling instances (variant bit 12), 4 words per packet: ticks of 10 ns
// {phase 1, phase 2, loading the packet's own leaf records, start tick}; npackets <= 2^18
int col_debug_walk_profile(uint32_t *out, uint32_t npackets) {
    if (npackets > COL_PROF_PACKETS) return COL_EINVAL;
    COL_HIP(hipDeviceSynchronize());
    COL_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_prof), 16ull * npackets));
    return COL_OK;
}

int col_bvh_build(void *stream, const uint32_t *codes, const uint32_t *ids, col_node *nodes, void *bounds,
                  uint32_t n, int coord_bytes) {
    if (n == 0) return COL_OK;
    if (n >= 0x80000000u) return COL_EINVAL;
    dim3 grid((unsigned)col_ceil_div(n, 256)), block(256);
    if (coord_bytes == 4) k_build<float><<<grid, block, 0, col_stream(stream)>>>(codes, ids, nodes, (float *)bounds, n);
    else if (coord_bytes == 8) k_build<double><<<grid, block, 0, col_stream(stream)>>>(codes, ids, nodes, (double *)bounds, n);
    else return COL_EINVAL;
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_bvh_refit(void *stream, void *bounds, uint32_t *flags, const void *coords, const void *radii,
                  const col_node *nodes, uint32_t n, int coord_bytes) {
    if (n == 0) return COL_OK;
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    dim3 leaves((unsigned)col_ceil_div(n, 256)), inner((unsigned)col_ceil_div(n > 1 ? n - 1 : 1, 256)),
        all((unsigned)col_ceil_div(2ull * n - 1, 256)), block(256);
    int rounds = 16;
    for (uint32_t m = n; m > 1; m >>= 1) rounds++;
    // progress words live in the flag slots of leaves 1..rounds (never used by the node protocol)
    uint32_t *progress = (n > (uint32_t)rounds + 1) ? flags + (n - 1) : nullptr;
    if (coord_bytes == 4) {
        k_refit_leaves<float><<<leaves, block, 0, s>>>((float *)bounds, flags, (const float *)coords, (const float *)radii, nodes, n);
        if (n > 1) {
            for (int r = 1; r <= rounds; r++) k_refit_sweep<float><<<inner, block, 0, s>>>((float *)bounds, flags, nodes, n, (u32)r, progress);
            k_refit_finish<float><<<all, block, 0, s>>>((float *)bounds, flags, nodes, n, progress ? progress + rounds : nullptr);
        }
    } else {
        k_refit_leaves<double><<<leaves, block, 0, s>>>((double *)bounds, flags, (const double *)coords, (const double *)radii, nodes, n);
        if (n > 1) {
            for (int r = 1; r <= rounds; r++) k_refit_sweep<double><<<inner, block, 0, s>>>((double *)bounds, flags, nodes, n, (u32)r, progress);
            k_refit_finish<double><<<all, block, 0, s>>>((double *)bounds, flags, nodes, n, progress ? progress + rounds : nullptr);
        }
    }
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_traverse(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const col_node *nodes,
                 const void *bounds, uint32_t n, int coord_bytes) {
    (void)nodes;   // the traversal runs on the 32-byte records in `bounds` alone
    if (n < 2) return COL_OK;
    if (capacity > 0 && !pairs) return COL_EINVAL;
    if (coord_bytes == 4) return launch_traverse<float>(stream, pairs, counter, capacity, bounds, n, nullptr, 0);
    if (coord_bytes == 8) return launch_traverse<double>(stream, pairs, counter, capacity, bounds, n, nullptr, 0);
    return COL_EINVAL;
}

size_t col_traverse_chunked_scratch_bytes(void) { return sizeof(ChunkHdr); }

// col_traverse for scenes with millions of pairs: the same pair list, but the workgroups take list space in chunks of
// 8192 pairs (one atomic on the counter per chunk instead of one per 512 pairs) and a small kernel closes the holes at
// the end.  *counter must be 0 on entry (the list starts here).  Record arrays below 4 GB and 64-byte aligned (what the
// asm walks need); anything else takes col_traverse.  scratch: col_traverse_chunked_scratch_bytes().
static int traverse_chunked_dev(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const col_node *nodes,
                                const void *bounds, uint32_t n, int coord_bytes, void *scratch, const uint32_t *n_dev,
                                uint32_t *walk_order = nullptr, bool hdr_cleared = false);
int col_traverse_chunked(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const col_node *nodes,
                         const void *bounds, uint32_t n, int coord_bytes, void *scratch) {
    return traverse_chunked_dev(stream, pairs, counter, capacity, nodes, bounds, n, coord_bytes, scratch, nullptr);
}
static int traverse_chunked_dev(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const col_node *nodes,
                                const void *bounds, uint32_t n, int coord_bytes, void *scratch, const uint32_t *n_dev,
                                uint32_t *walk_order, bool hdr_cleared) {
    const bool off32 = (coord_bytes == 4 || coord_bytes == 8) && (2ull * n - 1) * 8 * (unsigned)coord_bytes < (1ull << 32) &&
                       ((uintptr_t)bounds & 63) == 0;                                                // (see launch_traverse)
    if (!off32 || !scratch || (g_traverse_variant & ~(16384 | 32768 | 65536 | 131072 | 2097152 | 4194304 | 8388608)))
    {
        if (n < 2) return COL_OK;
        if (capacity > 0 && !pairs) return COL_EINVAL;
        if (coord_bytes == 4) return launch_traverse<float>(stream, pairs, counter, capacity, bounds, n, nullptr, 0, nullptr, n_dev);
        if (coord_bytes == 8) return launch_traverse<double>(stream, pairs, counter, capacity, bounds, n, nullptr, 0, nullptr, n_dev);
        return COL_EINVAL;
    }
    if (n < 2) return COL_OK;
    if (capacity > 0 && !pairs) return COL_EINVAL;
    if (coord_bytes == 8) return launch_traverse_chunked<double>(stream, pairs, counter, capacity, bounds, n, scratch, n_dev, walk_order, hdr_cleared);      // (round 4: the asm walk has a float64 form)
    return launch_traverse_chunked<float>(stream, pairs, counter, capacity, bounds, n, scratch, n_dev, walk_order, hdr_cleared);
}

// Diagnostics: same traversal, also accumulates stats[0..7] (8 x u64) = phase-2 steps, descents, leaf
// tests, leaf hits, and the steps within 1k / 2k / 4k / 8k sorted positions of the block's first leaf.  mode bit0 skips phase 1, bit1 skips phase 2 (timing ablations only).
int col_traverse_stats(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const void *bounds,
                       uint32_t n, int coord_bytes, uint64_t *stats, int mode) {
    if (n < 2) return COL_OK;
    if (coord_bytes == 4) return launch_traverse<float>(stream, pairs, counter, capacity, bounds, n, stats, mode);
    if (coord_bytes == 8) return launch_traverse<double>(stream, pairs, counter, capacity, bounds, n, stats, mode);
    return COL_EINVAL;
}

size_t col_collide_scratch_bytes(uint32_t n, uint32_t padded, int coord_bytes) {
    return 256 /* scene range */ + col_reduce_scratch_bytes(coord_bytes == 8 ? COL_F64 : COL_F32, 4) +
           col_radix_scratch_bytes(padded, 4, 4) + col_lbvh_scratch_bytes(n, coord_bytes) +
           (size_t)n * 4 * coord_bytes /* packed (x,y,z,r) rows */ + sizeof(ChunkHdr) + 256 /* chunked pair allocation */ + 1024 +
           8 * ((size_t)n / 64 + 1) + 256 /* walk order: cost[] + perm[] */;
}

int col_collide(void *stream, const void *coords, const void *radii, uint32_t n, uint32_t padded, int coord_bytes,
                uint32_t *codes0, uint32_t *codes1, uint32_t *ids0, uint32_t *ids1, col_node *nodes, void *bounds,
                uint32_t *flags, void *scratch, uint32_t *counter, uint32_t *pairs, uint32_t capacity) {
    return col_collide_plan(stream, coords, radii, n, padded, coord_bytes, codes0, codes1, ids0, ids1, nodes, bounds, flags,
                            scratch, counter, pairs, capacity, COL_SORT_LSD, nullptr);
}

int col_collide_plan(void *stream, const void *coords, const void *radii, uint32_t n, uint32_t padded, int coord_bytes,
                     uint32_t *codes0, uint32_t *codes1, uint32_t *ids0, uint32_t *ids1, col_node *nodes, void *bounds,
                     uint32_t *flags, void *scratch, uint32_t *counter, uint32_t *pairs, uint32_t capacity,
                     int sort_plan, uint32_t *oversize) {
    return col_collide_plan_partials(stream, coords, radii, n, padded, coord_bytes, codes0, codes1, ids0, ids1, nodes, bounds,
                                     flags, scratch, counter, pairs, capacity, sort_plan, oversize, nullptr, 0);
}

// col_collide_plan with the bounds partials of `coords` already computed (col_minmax4_stage1 / _dev: `parts` records of
// [min row, max row] at `partials`; NULL = compute them here).  The fused front end applies at every size of a (u32, u32) sort;
// only the unfused fallback below (a tile class it does not know) ignores the partials.
int col_collide_plan_partials(void *stream, const void *coords, const void *radii, uint32_t n, uint32_t padded, int coord_bytes,
                              uint32_t *codes0, uint32_t *codes1, uint32_t *ids0, uint32_t *ids1, col_node *nodes, void *bounds,
                              uint32_t *flags, void *scratch, uint32_t *counter, uint32_t *pairs, uint32_t capacity,
                              int sort_plan, uint32_t *oversize, const void *partials, uint32_t parts) {
    return col_collide_plan_dev(stream, coords, radii, n, padded, coord_bytes, codes0, codes1, ids0, ids1, nodes, bounds, flags, scratch,
                                counter, pairs, capacity, sort_plan, oversize, partials, parts, nullptr);
}

// The whole path with the number of spheres ON THE DEVICE (include/collision_hip.h): n is a host-known bound that sizes grids,
// scratch and the sort (rows from *n_dev on become pads), every kernel works on min(n, *n_dev).  Needs the bounds partials
// (col_minmax4_stage1_dev reads the same word).
int col_collide_plan_dev(void *stream, const void *coords, const void *radii, uint32_t n, uint32_t padded, int coord_bytes,
                         uint32_t *codes0, uint32_t *codes1, uint32_t *ids0, uint32_t *ids1, col_node *nodes, void *bounds,
                         uint32_t *flags, void *scratch, uint32_t *counter, uint32_t *pairs, uint32_t capacity,
                         int sort_plan, uint32_t *oversize, const void *partials, uint32_t parts, const uint32_t *n_dev) {
    if (coord_bytes != 4 && coord_bytes != 8) return COL_EINVAL;
    if (n_dev && !partials) return COL_EINVAL;
    if (partials && (parts == 0 || parts > COL_MINMAX_PARTS)) return COL_EINVAL;
    if (padded < n || (capacity > 0 && !pairs)) return COL_EINVAL;
    if (!scratch) return COL_ENOSCRATCH;
    // a forced radix tile class (diagnostics) would not match the histogram the fused Morton kernel writes
    if (col_radix_tile_override_active()) return COL_EINVAL;
    hipStream_t s = col_stream(stream);
    if (n == 0) {
        COL_HIP(hipMemsetAsync(counter, 0, sizeof(uint32_t), s));             // collision.py:151-154
        return COL_OK;
    }
    (void)flags;   // the arrival counters of internalBounds (collision.py:147-150) are not needed: see lbvh.hip
    char *p = (char *)scratch;
    void *range = p;           p += 256;
    void *red_scratch = p;     p += col_reduce_scratch_bytes(coord_bytes == 8 ? COL_F64 : COL_F32, 4);
    p = (char *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    void *sort_scratch = p;    p += col_radix_scratch_bytes(padded, 4, 4);
    p = (char *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    void *lbvh_scratch = p;    p += col_lbvh_scratch_bytes(n, coord_bytes);
    p = (char *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    void *packed = p;          p += (size_t)n * 4 * coord_bytes;
    p = (char *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    void *chunk_hdr = p;
    // sort_plan: bit 0 = the sort (COL_SORT_LSD / COL_SORT_MSD), bit 1 = COL_TRAVERSE_CHUNKED (dense scenes)
    const bool chunked = (sort_plan & COL_TRAVERSE_CHUNKED) != 0;
    sort_plan &= 1;
    uint32_t *publish = oversize ? oversize + 2 : nullptr;      // the previous call's pair count, for the caller's next choice
    int rc;
    uint32_t *report_word = nullptr;      // the LSD plan's bucket report rides in the tree build's first launch (lbvh.hip)
    const uint32_t tile = col_radix_tile(padded, 4, 4);
    if (tile == 1024 || tile == 4096 || tile == 8192) {
        // The front end is fused (every tile class of a (u32, u32) sort): the Morton kernel folds the bounds partials itself and
        // counts the digits of the sort's first pass (the histogram sits at the start of the sort scratch): two
        // launches less.  The MSD plan (one global pass + an LDS finish per bucket) applies up to COL_MSD_MAX_N.
        const bool msd = sort_plan == COL_SORT_MSD && padded <= COL_MSD_MAX_N;
        if (!partials) {
            if ((rc = col_minmax4_stage1(stream, coords, n, coord_bytes, red_scratch, &parts))) return rc;
            partials = red_scratch;
        }
        if ((rc = col_morton_tile(stream, coords, radii, partials, parts, n, padded, coord_bytes, codes0, ids0, packed,
                                  counter, (uint32_t *)sort_scratch, tile, (uint32_t)col_ceil_div(padded, tile), msd ? 22 : 0,
                                  publish, n_dev))) return rc;
        if (msd) rc = col_radix_sort_msd_dev(stream, codes0, codes1, ids0, ids1, padded, sort_scratch, oversize, n_dev);
        else rc = col_radix_sort_ex(stream, codes0, codes1, ids0, ids1, padded, 4, 4, sort_scratch, 0, 1);
        if (rc) return rc;
        // the LSD plan was taken where the MSD plan could apply: tell the caller how clustered the codes are (oversize[1])
        if (!msd && oversize && padded <= COL_MSD_MAX_N) report_word = oversize + 1;
    } else {
        if (n_dev) return COL_EINVAL;            // (a forced tile class: diagnostics only)
        if ((rc = col_reduce(stream, coords, n, coord_bytes == 8 ? COL_F64 : COL_F32, 4, COL_OP_MINMAX, red_scratch, range))) return rc;
        if ((rc = col_morton_ex(stream, coords, radii, range, n, padded, coord_bytes, codes0, ids0, packed, counter, publish))) return rc;
        if ((rc = col_radix_sort(stream, codes0, codes1, ids0, ids1, padded, 4, 4, sort_scratch, 0))) return rc;
    }
    // the traversal's packet counters (dynamic packet order, k_traverse): cleared by the tree build's last kernel
    // (variant bit 15: from any size, so that the small parity cases of tests/ take this way too)
    uint32_t *sched = (!chunked && (n >= COL_DYNAMIC_PACKETS_FROM || (g_traverse_variant & 32768))) ? (uint32_t *)((char *)chunk_hdr + sizeof(ChunkHdr)) : nullptr;
    // the walk order (col_common.h): cost[] and perm[] behind the chunk header, wherever the dynamic packet order runs and the tree
    // build ends with k_cross (more than one chunk)
    // WALK ORDER (col_common.h): a dense scene orders its walks longest first, a sparse one of up to COL_DEAL_MAX_N spheres deals its
    // long walks over the first round; larger sparse scenes record and order nothing
    const int order_mode = chunked || col_lbvh_order_forced() ? 1 : 0;
    uint32_t *walk_order = n > 256 && (chunked || (sched && (n <= COL_DEAL_MAX_N || order_mode))) ? (uint32_t *)((char *)chunk_hdr + sizeof(ChunkHdr) + 256) : nullptr;
    // (chunked: the tree build's last kernel clears the walk's whole 64-byte header -- its memset was a launch of 4.8 us)
    if ((rc = col_lbvh_ex(stream, codes1, ids1, coords, radii, packed, nodes, bounds, lbvh_scratch, n, coord_bytes,
                          chunked ? (uint32_t *)chunk_hdr : sched, n_dev, walk_order, order_mode, chunked ? 16u : 8u, report_word, padded))) return rc;
    if (chunked) return traverse_chunked_dev(stream, pairs, counter, capacity, nodes, bounds, n, coord_bytes, chunk_hdr, n_dev, walk_order, true);
    if (n < 2) return COL_OK;
    if (coord_bytes == 4) return launch_traverse<float>(stream, pairs, counter, capacity, bounds, n, nullptr, 0, sched, n_dev, walk_order);
    return launch_traverse<double>(stream, pairs, counter, capacity, bounds, n, nullptr, 0, sched, n_dev, walk_order);
}

}  // extern "C"

// Fused Karras build + AABB refit for the production path (col_collide).
// Replaces fillInternal + generateBVH + leafBounds + internalBounds
// (collision/collision.cl:55-162, enqueued at collision/collision.py:171-190) and produces the
// same `nodes` and `bounds` arrays as bvh.hip's k_build + k_refit (which stay as the
// reference-shaped, tree-generic kernels for the kernel-level parity tests).
//
// Why not walk the tree: internalBounds hands child boxes from one workgroup to another through
// a flag atomic (collision.cl:152-161).  On MI355X that needs an agent-scope release + acquire
// per level per wave (8 XCDs, non-coherent L2s) and cost 6.3 ms at 1 M spheres.  An LBVH node
// covers a CONTIGUOUS range [first, last] of sorted leaves, so its box is a range-min/max query
// over the leaf boxes, and min/max are exact and idempotent: any grouping gives the reference's
// bits.  So there is no inter-workgroup hand-off at all:
//
//   k_chunk   one block per 256 consecutive sorted leaves: sorted codes staged in an LDS window
//             (chunk +- 256) for Karras' dependent probes; leaf boxes -> LDS; Karras topology for
//             internal nodes [c*256, c*256+256); a node whose range stays inside the chunk waits in
//             LDS (workgroup scope) until both children are final, merges them and publishes its
//             box; every node is written as one 32-byte record (box + traversal links) per thread.
//             A node that crosses the chunk boundary stores the half of its range that lies in this
//             chunk (`partial`: the prefix / suffix unions of the chunk's leaf boxes).
//   k_group   (1-2 tiny launches) sparse tables over chunk totals, 256 chunks per group,
//             then over group totals, ...
//   k_cross   the ~1-2 % of nodes that cross chunks: partial[first] U partial[last] U
//             O(1) table look-ups for the whole chunks in between.  k_chunk lists them per chunk (CROSS_CAP words),
//             k_cross runs one thread per list slot (round 3: 76 -> 43 us at 16 M spheres; a thread per NODE left
//             one or two busy lanes per wave).
//
// partial[] is indexed by NODE index: a crossing node that runs forward from `first` owns
// partial[first]; the far end `last` of that node is the index of the backward crossing node
// that ends at `last` (its nearest left-child ancestor-or-self chain: Karras gives a left child
// the index of its last leaf), which stored the prefix box of its chunk up to `last` -- exactly
// the piece the forward node is missing.  Symmetrically for backward nodes; leaf n-1 (no
// internal node has that index) stores its prefix itself.
#include "col_common.h"
#include <math.h>

namespace {

constexpr u32 END = 0xFFFFFFFFu;
constexpr int C = 256;          // leaves per chunk == threads per block
constexpr int LV = 9;           // table levels 0..8 (2^8 = 256)

template <typename T> struct BT;
template <> struct BT<float> { typedef float4 V4; typedef uint32_t Bits; };
template <> struct BT<double> { typedef double4 V4; typedef uint64_t Bits; };

template <typename T> struct Box { T lo[3], hi[3]; };

template <typename T> __device__ __forceinline__ Box<T> box_empty() {
    Box<T> b;
#pragma unroll
    for (int a = 0; a < 3; a++) { b.lo[a] = (T)INFINITY; b.hi[a] = -(T)INFINITY; }
    return b;
}
// min / max as ONE instruction.  `o < a ? o : a` compiles to v_cmp + two wait states + v_cndmask on gfx950 (a VALU
// instruction may not read an SGPR mask a VALU instruction has just written), and box unions are a third of k_chunk's
// vector instructions.  v_min / v_max return the same bits as the compare-and-select for every pair of ordinary
// numbers; they differ only where the reference's own `min(x, y)` (collision.cl:157: "y < x ? y : x") depends on the
// ORDER of its arguments -- a tie between +0 and -0 (the instruction: -0 for min, +0 for max, whatever the order) and
// NaN coordinates (ignored unless both are NaN) -- i.e. where a range-query refit could not promise the tree-order
// result anyway; no fixture of the reference has either.
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double hw_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double hw_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <typename T> __device__ __forceinline__ void box_merge(Box<T> &a, const Box<T> &o) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        a.lo[k] = hw_min(a.lo[k], o.lo[k]);
        a.hi[k] = hw_max(a.hi[k], o.hi[k]);
    }
}

// global table / partial entry: two vec4 rows (lo.xyz, -) (hi.xyz, -)
template <typename T> __device__ __forceinline__ void box_store(T *base, uint64_t entry, const Box<T> &b) {
    typedef typename BT<T>::V4 V4;
    V4 *p = reinterpret_cast<V4 *>(base) + 2 * entry;
    V4 lo, hi;
    lo.x = b.lo[0]; lo.y = b.lo[1]; lo.z = b.lo[2]; lo.w = (T)0;
    hi.x = b.hi[0]; hi.y = b.hi[1]; hi.z = b.hi[2]; hi.w = (T)0;
    p[0] = lo; p[1] = hi;
}
template <typename T> __device__ __forceinline__ Box<T> box_load(const T *base, uint64_t entry) {
    typedef typename BT<T>::V4 V4;
    const V4 *p = reinterpret_cast<const V4 *>(base) + 2 * entry;
    const V4 lo = p[0], hi = p[1];
    Box<T> b;
    b.lo[0] = lo.x; b.lo[1] = lo.y; b.lo[2] = lo.z;
    b.hi[0] = hi.x; b.hi[1] = hi.y; b.hi[2] = hi.z;
    return b;
}

// LDS sparse table, structure of arrays: t[level][component][pos]
template <typename T> struct Table { T v[LV][6][C]; };

template <typename T> __device__ __forceinline__ void tab_put(Table<T> &t, int l, int pos, const Box<T> &b) {
#pragma unroll
    for (int k = 0; k < 3; k++) { t.v[l][k][pos] = b.lo[k]; t.v[l][3 + k][pos] = b.hi[k]; }
}
template <typename T> __device__ __forceinline__ Box<T> tab_get(const Table<T> &t, int l, int pos) {
    Box<T> b;
#pragma unroll
    for (int k = 0; k < 3; k++) { b.lo[k] = t.v[l][k][pos]; b.hi[k] = t.v[l][3 + k][pos]; }
    return b;
}
// level l from level l-1 (all 256 threads, barrier inside)
template <typename T> __device__ __forceinline__ void tab_build(Table<T> &t, int tid) {
    for (int l = 1; l < LV; l++) {
        __syncthreads();
        Box<T> b = tab_get(t, l - 1, tid);
        const int o = tid + (1 << (l - 1));
        if (o < C) box_merge(b, tab_get(t, l - 1, o));
        tab_put(t, l, tid, b);
    }
    __syncthreads();
}
// union of entries [a, b], 0 <= a <= b < 256
template <typename T> __device__ __forceinline__ Box<T> tab_query(const Table<T> &t, int a, int b) {
    const int k = 31 - __clz(b - a + 1);
    Box<T> r = tab_get(t, k, a);
    box_merge(r, tab_get(t, k, b - (1 << k) + 1));
    return r;
}

// same query against a global table laid out [group][level][256] of 2 x vec4 entries
template <typename T> __device__ __forceinline__ Box<T> gtab_query(const T *tab, uint64_t group, int a, int b) {
    const int k = 31 - __clz(b - a + 1);
    const uint64_t base = (group * LV + k) * C;
    Box<T> r = box_load(tab, base + a);
    box_merge(r, box_load(tab, base + b - (1 << k) + 1));
    return r;
}

// Sorted codes seen through an LDS window [w0, w0 + WIN) around the chunk: almost every Karras
// probe of a node stays within a few hundred leaves of it, so the dependent-load chains of the
// search run at LDS latency; probes outside the window fall back to global memory.
constexpr int HALO = 256;
constexpr int WIN = C + 2 * HALO;
// I: the type of a (possibly out-of-range) leaf position in the Karras search: int32_t below 2^30 leaves --
// i +- 2 * range cannot overflow it, and 32-bit index arithmetic is half the vector instructions -- else int64_t
typedef __attribute__((address_space(3))) const u32 LdsWord;
// FAST (32-bit positions only): delta() without per-lane branches, see below.
template <typename I, bool FAST = false> struct Codes {
    const u32 *__restrict__ g;
    LdsWord *win;              // (an LDS pointer by type: as a generic pointer the two sides of at() become ONE flat_load)
    I w0;
    u32 n;
    __device__ __forceinline__ u32 at(I j) const {          // 0 <= j < n
        if constexpr (FAST && sizeof(I) == 4) {
            // one wave-uniform branch instead of a branch per lane (see delta() below)
            const u32 o = (u32)(j - w0);
            const bool inwin = o < (u32)WIN;
            if (__builtin_amdgcn_ballot_w64(!inwin) == 0) return win[o];
            return inwin ? win[o] : g[j];
        } else {
            const I o = j - w0;
            return (o >= 0 && o < WIN) ? win[o] : g[j];
        }
    }
};
// v_ffbh_u32: leading zeros, 0xFFFFFFFF for 0
__device__ __forceinline__ u32 ffbh_raw(u32 x) { u32 r; asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x)); return r; }
// collision.cl:65-77
template <typename I, bool FAST> __device__ __forceinline__ int delta(const Codes<I, FAST> &c, u32 i, u32 ci, I j) {
    if constexpr (FAST && sizeof(I) == 4) {
        // The plain form below compiles to three nested per-lane branches (range check, window or global memory, equal
        // codes): ~40 instructions per probe, a third of them scalar EXEC bookkeeping, and a wave makes ~28 probes.  Here:
        // one wave-uniform branch (does ANY lane's probe leave the code window?  almost never), an unconditional LDS read, and
        // the tie rule as an unsigned minimum -- clz(ci ^ cj) is 0xFFFFFFFF exactly when the codes are equal, and then
        // 32 + clz(i ^ j) (j != i in every probe, so that is at most 63) is the smaller one.
        const bool inb = (u32)j < c.n;                     // 0 <= j < n
        const u32 o = (u32)(j - c.w0);
        const bool inwin = o < (u32)WIN;
#ifdef COL_EXP_NO_FAR      // timing experiment: no probe leaves the code window (wrong trees)
        if (!inwin) return -1;
#endif
        u32 cj;
        if (__builtin_amdgcn_ballot_w64(inb && !inwin) == 0) cj = c.win[inwin ? o : 0u];
        else cj = inb ? (inwin ? c.win[o] : c.g[j]) : 0u;
        const u32 d = min(ffbh_raw(ci ^ cj), 32u + ffbh_raw(i ^ (u32)j));
        return inb ? (int)d : -1;
    } else {
        if (j < 0 || j >= (I)c.n) return -1;
        const u32 cj = c.at(j);
        return ci != cj ? __clz((int)(ci ^ cj)) : 32 + __clz((int)(i ^ (u32)j));
    }
}
// the right child that starts at leaf k (see bvh.hip)
template <typename I, bool FAST> __device__ __forceinline__ u32 right_child_at(const Codes<I, FAST> &c, u32 k) {
    if (k + 1 >= c.n) return (c.n - 1) + k;
    const u32 ck = c.at((I)k);
    const bool fwd = delta(c, k, ck, (I)k + 1) > delta(c, k, ck, (I)k - 1);
    return fwd ? k : (c.n - 1) + k;
}

template <typename T>
__device__ __forceinline__ void record_store(T *bounds, uint64_t node, const Box<T> &b, u32 skip, u32 down) {
    typedef typename BT<T>::V4 V4;
    typedef typename BT<T>::Bits Bits;
    V4 lo, hi;
    lo.x = b.lo[0]; lo.y = b.lo[1]; lo.z = b.lo[2];
    hi.x = b.hi[0]; hi.y = b.hi[1]; hi.z = b.hi[2];
    const Bits s = (Bits)skip, d = (Bits)down;
    lo.w = *reinterpret_cast<const T *>(&s);
    hi.w = *reinterpret_cast<const T *>(&d);
    V4 *p = reinterpret_cast<V4 *>(bounds) + 2 * node;
    p[0] = lo; p[1] = hi;
}
template <typename T> __device__ __forceinline__ void links_store(T *bounds, uint64_t node, u32 skip, u32 down) {
    typedef typename BT<T>::Bits Bits;
    Bits *p = reinterpret_cast<Bits *>(bounds) + 8 * node;
    p[3] = (Bits)skip;
    p[7] = (Bits)down;
}

// LDS image of one chunk: the leaves, finished boxes of in-chunk internal nodes, ready flags / meeting places.
// Round 4, second half: the image is PACKED so that more workgroups than the CU's 32 wave slots admit fit its LDS (f32: 14.9 KB
// = 12 granules of 1280 B: ten workgroups; it was 19.7 KB: eight) -- the waves of a workgroup that have no crossing node leave
// right after their stores while the others search, and the wave slots they free are only of use to a NEW workgroup if the
// LDS has room for it; float64 (25.3 KB, was 31.4) gets six workgroups per CU instead of five.
//   leaf[4][C]  the leaf's sphere (x, y, z, r), not its box: the box is c -+ r wherever it is read -- the same two IEEE
//               operations on the same operands as in the leaf's own thread, so the same bits -- and a third less LDS
//   flags       the searching instances' ready[C + 1] ([C] = 1 for ever: the 'flag' of a child that is a leaf); CLIMB: meet[],
//               two 16-bit places per word (a value is a far end + 1 <= 256; the first arrival ORs it into a zero place and
//               sees zero, the second sees the first's value -- and spoils the place, which nobody reads again)
//   adj         one signed byte per delta (-1 .. 63)
//   oe, split   CLIMB: other end and split of node c0 + t as chunk-local positions in a byte each; oe[t] == t = not an
//               in-chunk node (a node's range holds two leaves or more: its other end is never itself)
template <typename T, bool CLIMB> struct ChunkLds {
    T leaf[4][C];
    T node[6][C];
    T wave_tot[2][C / COL_WAVE][6];     // per-wave totals for the prefix / suffix scans
    u32 flags[CLIMB ? C / 2 : C + 1];
    u32 ncross;                         // nodes of this chunk that cross its boundary, so far (see CROSS_CAP)
    signed char adj[C + 4];             // adj[1 + t] = delta(p, p + 1) of the chunk's position t, adj[0] = delta(c0 - 1, c0) (FAST_DELTA: see k_chunk)
    unsigned char oe[CLIMB ? C : 4], split[CLIMB ? C : 4];
};
template <typename T> __device__ __forceinline__ Box<T> leaf_get(const T (&a)[4][C], int pos) {
    const T x = a[0][pos], y = a[1][pos], z = a[2][pos], r = a[3][pos];
    Box<T> b;
    b.lo[0] = x - r; b.lo[1] = y - r; b.lo[2] = z - r;
    b.hi[0] = x + r; b.hi[1] = y + r; b.hi[2] = z + r;
    return b;
}
// The 1-2 % of nodes that cross a chunk boundary are listed per chunk for k_cross: CROSS_CAP words per chunk, END = free;
// a chunk with more than CROSS_CAP - 1 of them (deep trees: duplicate codes) sets the last word to CROSS_DENSE and k_cross
// goes through all its nodes instead.  (32 since the end of round 4.  tests/analysis/crossing_nodes_per_chunk.py: a chunk has 8.3
// crossing nodes on average, at most 14 on config 2 -- and up to 21 on config 3, where 13 of the 3907 chunks had more than 15: a
// handful of dense chunks, sixteen dependent rounds each, were enough to hold k_cross for 15 us instead of 5.7; whole step 0.515 ->
// 0.505 ms; 64 gains nothing more and costs the 16 M launch 1 %.)
constexpr u32 CROSS_CAP = 32, CROSS_DENSE = 0xFFFFFFFEu;
template <typename T> __device__ __forceinline__ Box<T> soa_get(const T (&a)[6][C], int pos) {
    Box<T> b;
#pragma unroll
    for (int k = 0; k < 3; k++) { b.lo[k] = a[k][pos]; b.hi[k] = a[3 + k][pos]; }
    return b;
}
template <typename T> __device__ __forceinline__ void soa_put(T (&a)[6][C], int pos, const Box<T> &b) {
#pragma unroll
    for (int k = 0; k < 3; k++) { a[k][pos] = b.lo[k]; a[3 + k][pos] = b.hi[k]; }
}
template <typename T> __device__ __forceinline__ Box<T> box_shfl_up(const Box<T> &b, int o) {
    Box<T> r;
#pragma unroll
    for (int k = 0; k < 3; k++) { r.lo[k] = __shfl_up(b.lo[k], o, COL_WAVE); r.hi[k] = __shfl_up(b.hi[k], o, COL_WAVE); }
    return r;
}
template <typename T> __device__ __forceinline__ Box<T> box_shfl_down(const Box<T> &b, int o) {
    Box<T> r;
#pragma unroll
    for (int k = 0; k < 3; k++) { r.lo[k] = __shfl_down(b.lo[k], o, COL_WAVE); r.hi[k] = __shfl_down(b.hi[k], o, COL_WAVE); }
    return r;
}

// Inclusive prefix (lanes 0..l) and suffix (lanes l..63) unions of a wave's f32 boxes on the DPP network: per value one
// v_min / v_max_f32_dpp per step (row_shr 1, 2, 4, 8 inside the rows of 16 -- a lane the shift does not reach keeps its
// value --, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3), and for the suffix row_shl 1, 2, 4, 8
// and three rounds "first lane of row 3 / 2 / 1 -> the rows below it" (v_readlane + a v_min / v_max under a narrowed EXEC:
// there is no broadcast towards lower rows).  96 vector instructions and no LDS for the twelve scans instead of
// 144 ds_bpermute + ~240 VALU.  The six values are interleaved, so dependent DPP steps are six instructions apart (two
// wait states are needed after the VALU write of a DPP source, and after a VALU write of an SGPR that a VALU reads);
// s_nop 1 at both ends because the compiler does not track these hazards across the asm boundary.  Needs a full wave.
#define COL_SCAN6(ctrl)                                                                                                       \
    "v_min_f32_dpp %0, %0, %0 " ctrl "\n\tv_min_f32_dpp %1, %1, %1 " ctrl "\n\tv_min_f32_dpp %2, %2, %2 " ctrl "\n\t"        \
    "v_max_f32_dpp %3, %3, %3 " ctrl "\n\tv_max_f32_dpp %4, %4, %4 " ctrl "\n\tv_max_f32_dpp %5, %5, %5 " ctrl "\n\t"
#define COL_DOWN6(src_lane)                                                                                                   \
    "v_readlane_b32 %6, %0, " src_lane "\n\tv_readlane_b32 %7, %1, " src_lane "\n\tv_readlane_b32 %8, %2, " src_lane "\n\t"   \
    "v_readlane_b32 %9, %3, " src_lane "\n\tv_readlane_b32 %10, %4, " src_lane "\n\tv_readlane_b32 %11, %5, " src_lane "\n\t"
#define COL_APPLY6                                                                                                            \
    "v_min_f32 %0, %6, %0\n\tv_min_f32 %1, %7, %1\n\tv_min_f32 %2, %8, %2\n\t"                                               \
    "v_max_f32 %3, %9, %3\n\tv_max_f32 %4, %10, %4\n\tv_max_f32 %5, %11, %5\n\t"
__device__ __forceinline__ void wave_prefix_union(Box<float> &b) {
    asm volatile("s_nop 1\n\t"
                 COL_SCAN6("row_shr:1 row_mask:0xf bank_mask:0xf")
                 COL_SCAN6("row_shr:2 row_mask:0xf bank_mask:0xf")
                 COL_SCAN6("row_shr:4 row_mask:0xf bank_mask:0xf")
                 COL_SCAN6("row_shr:8 row_mask:0xf bank_mask:0xf")
                 COL_SCAN6("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 COL_SCAN6("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(b.lo[0]), "+v"(b.lo[1]), "+v"(b.lo[2]), "+v"(b.hi[0]), "+v"(b.hi[1]), "+v"(b.hi[2]));
}
__device__ __forceinline__ void wave_suffix_union(Box<float> &b) {
    u32 t0, t1, t2, t3, t4, t5;
    u64 save;
    asm volatile("s_nop 1\n\t"
                 COL_SCAN6("row_shl:1 row_mask:0xf bank_mask:0xf")
                 COL_SCAN6("row_shl:2 row_mask:0xf bank_mask:0xf")
                 COL_SCAN6("row_shl:4 row_mask:0xf bank_mask:0xf")
                 COL_SCAN6("row_shl:8 row_mask:0xf bank_mask:0xf")
                 "s_mov_b64 %12, exec\n\t"
                 "s_nop 1\n\t"
                 COL_DOWN6("48")                                   // row 3's total -> rows 0..2
                 "s_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, 0xffff\n\t"
                 COL_APPLY6
                 COL_DOWN6("32")                                   // rows 2..3 -> rows 0..1
                 "s_mov_b32 exec_hi, 0\n\t"
                 COL_APPLY6
                 COL_DOWN6("16")                                   // rows 1..3 -> row 0
                 "s_mov_b32 exec_lo, 0xffff\n\t"
                 COL_APPLY6
                 "s_mov_b64 exec, %12\n\t"
                 "s_nop 1"
                 : "+v"(b.lo[0]), "+v"(b.lo[1]), "+v"(b.lo[2]), "+v"(b.hi[0]), "+v"(b.hi[1]), "+v"(b.hi[2]),
                   "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(save));
}
#undef COL_SCAN6
#undef COL_DOWN6
#undef COL_APPLY6
// The same scans for float64 boxes (round 4: the f64 kernel shuffled its twelve doubles through 144 x 2 ds_bpermute and took
// 76.6 us at 1 M spheres against 51 for f32).  There is no v_min_f64 with a DPP operand, so a step moves the two words of a
// value with v_mov_b32_dpp (a lane the shift does not reach keeps its own value: `old` = the source) and merges with
// v_min / v_max_f64; hipcc places the wait states of the builtins itself.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = (int)(u32)b, hi = (int)(u32)((u64)b >> 32);
    const u32 l2 = (u32)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    const u32 h2 = (u32)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double((long long)(((u64)h2 << 32) | l2));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ void box_dpp_merge(Box<double> &b) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        b.lo[k] = hw_min(b.lo[k], dpp_d<CTRL, ROW_MASK>(b.lo[k]));
        b.hi[k] = hw_max(b.hi[k], dpp_d<CTRL, ROW_MASK>(b.hi[k]));
    }
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)b, l), hi = (u32)__builtin_amdgcn_readlane((int)(u32)((u64)b >> 32), l);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ void wave_prefix_union(Box<double> &b) {
    box_dpp_merge<0x111, 0xF>(b);       // row_shr:1, 2, 4, 8 inside the rows of 16
    box_dpp_merge<0x112, 0xF>(b);
    box_dpp_merge<0x114, 0xF>(b);
    box_dpp_merge<0x118, 0xF>(b);
    box_dpp_merge<0x142, 0xA>(b);       // row_bcast:15 -> rows 1, 3
    box_dpp_merge<0x143, 0xC>(b);       // row_bcast:31 -> rows 2, 3
}
__device__ __forceinline__ void wave_suffix_union(Box<double> &b) {
    box_dpp_merge<0x101, 0xF>(b);       // row_shl:1, 2, 4, 8
    box_dpp_merge<0x102, 0xF>(b);
    box_dpp_merge<0x104, 0xF>(b);
    box_dpp_merge<0x108, 0xF>(b);
    const u32 lane = lane_id();
#pragma unroll
    for (int first = 48; first >= 16; first -= 16) {      // row 3's total -> rows 0..2, rows 2..3 -> rows 0..1, rows 1..3 -> row 0
        Box<double> t;
#pragma unroll
        for (int k = 0; k < 3; k++) { t.lo[k] = readlane_d(b.lo[k], first); t.hi[k] = readlane_d(b.hi[k], first); }
        if (lane < (u32)first) box_merge(b, t);
    }
}

// node records without the `parent` word, which the parent's thread writes (collision.cl:119-120)
struct __attribute__((packed, aligned(4))) LeafTail { u32 right_edge, id; };
struct __attribute__((packed, aligned(4))) InnerTail { u32 right_edge, child_a, child_b; };

// Diagnostics (col_debug_lbvh: timing ablations of k_chunk) live in a separate instance, DIAG = true, with its
// own kernel argument; the production instance takes no mode argument and carries none of the branches.
struct ChunkDiagOff {};
struct ChunkDiagOn { int mode; };
template <bool DIAG> struct ChunkDiag { typedef ChunkDiagOff T; };
template <> struct ChunkDiag<true> { typedef ChunkDiagOn T; };
__device__ __forceinline__ constexpr int chunk_mode(ChunkDiagOff) { return 0; }
__device__ __forceinline__ int chunk_mode(ChunkDiagOn d) { return d.mode; }

// WALK ORDER (col_common.h), workgroup `ob` of the 8 x COL_ORDER_SLICES that order the packets of the traversal that follows: part of
// k_chunk's launch (the first workgroups of its grid: done long before the launch is) or, with the searching instances of k_chunk, of
// k_cross's (where they were first, and made that launch 11 us instead of 5 at 1 M spheres).  n: this call's number of leaves.
constexpr u32 COL_ORDER_SLICES = 8;     // workgroups per XCD that order its packets
constexpr u32 COL_DEAL_UPTO = COL_DEAL_MAX_N / 512;      // packets per XCD up to which a sparse scene's long walks are dealt
__device__ __forceinline__ void order_packets(u32 ob, u32 n, u32 n_bound, u32 *__restrict__ walk_order, int order_mode) {
    // WALK ORDER (col_common.h): COL_ORDER_SLICES workgroups per XCD look at the costs the previous call's walks left for the
    // XCD's packets and decide this call's order.  The statistics come from a sample of at most 1024 packets (every
    // workgroup of the XCD takes the same one: 16 M spheres are 31 250 packets per XCD, and reading them all in one
    // workgroup made this launch 105 us instead of 42).
    __shared__ unsigned long long s_sum;
    __shared__ u32 s_cnt[8], s_cur[8];
    const u32 x = ob & 7u, g = ob >> 3, npk_bound = (n_bound + 63u) / 64u;
    const u32 *cost = walk_order;
    u32 *perm = walk_order + npk_bound;
    u32 p_lo, p_end;
    xcd_packet_range((n + 63u) / 64u, x, p_lo, p_end);
    if (threadIdx.x == 0) s_sum = 0;
    if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const u32 cnt = p_end > p_lo ? p_end - p_lo : 0u;
    const u32 stride = (cnt + 1023u) / 1024u, nsamp = stride ? (cnt + stride - 1u) / stride : 0u;
    u32 sample[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {                  // (the four loads of a thread go out together: each is a round trip to memory)
        const u32 e = threadIdx.x + 256u * k;
        sample[k] = e < nsamp ? min(cost[p_lo + e * stride], 1u << 24) : 0u;        // (a first call's garbage, capped)
    }
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) mine += sample[k];
    atomicAdd(&s_sum, mine);
    __syncthreads();
    const unsigned long long mean = nsamp ? max(s_sum / nsamp, 1ull) : 1ull;
    u32 *use_perm = walk_order + 2u * npk_bound + x;       // read by k_traverse's workgroups of this XCD
    if (!cnt) { if (threadIdx.x == 0 && g == 0) *use_perm = 0; return; }
    // Which order: decided by the caller (col_common.h, WALK ORDER).  The walks' times themselves do not tell a dense scene from a
    // large sparse one -- a walk of a 16 M-sphere uniform scene takes 25 us among its 8191 neighbours, and a longest-first order
    // there (it scatters a batch's 16 neighbouring packets, which walk nearly the same nodes, over the XCD's range) cost the
    // traversal 15 %.
    if (order_mode != 1) {
        if (g) return;
        // Moderate scenes (config 2: walks of 7..38 us around a mean of 13, the same packets long in every call): the kernel
        // ends when the last long walk does, so the long walks (>= 1.5 x the mean: one packet in twelve) must START in the
        // first round -- but SPREAD over it: a CU's scalar unit is what its 32 walks share, and long walks in a row (a
        // longest-first list puts sixteen into one workgroup) only slow each other down.  So they are dealt over the leading
        // slots of the XCD's first COL_TRAV_FIRST_BATCHES batches (one per workgroup of the traversal's grid), the longest
        // class (>= 2 x) first; every other packet keeps its natural order in the slots that remain.
        // ... and only where the launch is a few rounds long (up to COL_DEAL_UPTO packets on the XCD: 1.3 M spheres): the order costs
        // every packet a dependent load of its number (1 M: - 1.6 us net of that; 2 M: + 6 us, `gpurun_out/order_ab_6.txt`)
        const u32 nb1 = min(cnt / (u32)COL_TRAV_WAVES, (u32)COL_TRAV_FIRST_BATCHES);
        if (nb1 == 0 || cnt > COL_DEAL_UPTO) {
            if (threadIdx.x == 0) *use_perm = 0;          // natural order: k_traverse does not read perm[]
            return;
        }
        const u32 thr2 = (u32)min(2ull * mean, 0xFFFFFFFFull), thr1 = (u32)min(mean + mean / 2, 0xFFFFFFFFull);
        if (threadIdx.x < 2) s_cur[threadIdx.x] = 0;
        constexpr int PER = (int)(COL_DEAL_UPTO + 255u) / 256;       // a thread's packets: PER consecutive ones from p_lo + PER * threadIdx.x
        u32 mycost[PER];
        const u32 q0 = p_lo + (u32)PER * threadIdx.x;
#pragma unroll
        for (int k = 0; k < PER; k++) mycost[k] = q0 + k < p_end ? min(cost[q0 + k], 1u << 24) : 0u;
        u32 my_shorts = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (q0 + k >= p_end) continue;
            if (mycost[k] >= thr1) atomicAdd(&s_cnt[mycost[k] >= thr2 ? 0 : 1], 1u);
            else my_shorts++;
        }
        __syncthreads();
        const u32 l2 = s_cnt[0], l = l2 + s_cnt[1];
        if (l == 0 || l > 8u * nb1) {
            if (threadIdx.x == 0) *use_perm = 0;
            return;
        }
        if (threadIdx.x == 0) *use_perm = 1;
        const u32 per = l / nb1, rem = l % nb1;                      // batch j starts with per + (j < rem) long walks
        const u32 cap_a = (u32)COL_TRAV_WAVES - per - 1u, cap_b = (u32)COL_TRAV_WAVES - per, in_a = rem * cap_a, in_ab = in_a + (nb1 - rem) * cap_b;
        __shared__ u32 s_ws[4];
        u32 tot;
        u32 r = block_excl_scan<256>(my_shorts, s_ws, &tot);       // (one scan: the threads' runs of packets are in natural order)
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const u32 q = q0 + k, c = mycost[k];
            if (q >= p_end) continue;
            u32 pos;
            if (c < thr1) {
                if (r < in_a) pos = (r / cap_a) * (u32)COL_TRAV_WAVES + per + 1u + r % cap_a;
                else if (r < in_ab) pos = (rem + (r - in_a) / cap_b) * (u32)COL_TRAV_WAVES + per + (r - in_a) % cap_b;
                else pos = nb1 * (u32)COL_TRAV_WAVES + (r - in_ab);
                r++;
            } else {
                const u32 kk = c >= thr2 ? atomicAdd(&s_cur[0], 1u) : l2 + atomicAdd(&s_cur[1], 1u);
                pos = (kk % nb1) * (u32)COL_TRAV_WAVES + kk / nb1;
            }
            perm[p_lo + pos] = c < thr1 ? q : q | 0x80000000u;        // (bit 31: walked at raised priority)
        }
        return;
    }
    // Longest first: eight classes by cost * 4 / mean (capped), the longest class first, any order inside a class (an LDS
    // cursor).  From 4096 packets the XCD's range is cut into COL_ORDER_SLICES slices, a workgroup each, and the slices' orders
    // are interleaved (element k of slice g goes to place k * slices + g): longest first up to the slices' differences.
    const u32 slices = cnt >= 4096u ? (u32)COL_ORDER_SLICES : 1u;
    if (g >= slices) return;
    if (threadIdx.x == 0 && g == 0) *use_perm = 1;
    const u32 len = cnt / slices, longer = cnt % slices;                       // slice g: len + (g < longer) packets
    const u32 s_lo = p_lo + g * len + min(g, longer), s_end = s_lo + len + (g < longer ? 1u : 0u);
    // (four loads of a thread go out together in both passes: each is a round trip to memory)
    auto cls_of = [&](u32 c) -> u32 { return (u32)min(7ull, (unsigned long long)c * 4ull / mean); };
    for (u32 base = s_lo; base < s_end; base += 1024) {
        u32 c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const u32 q = base + 256u * k + threadIdx.x; c[k] = q < s_end ? min(cost[q], 1u << 24) : 0u; }
#pragma unroll
        for (int k = 0; k < 4; k++) if (base + 256u * k + threadIdx.x < s_end) atomicAdd(&s_cnt[cls_of(c[k])], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) { u32 acc = 0; for (int k = 7; k >= 0; k--) { s_cur[k] = acc; acc += s_cnt[k]; } }
    __syncthreads();
    for (u32 base = s_lo; base < s_end; base += 1024) {
        u32 c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const u32 q = base + 256u * k + threadIdx.x; c[k] = q < s_end ? min(cost[q], 1u << 24) : 0u; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 q = base + 256u * k + threadIdx.x;
            if (q >= s_end) continue;
            const u32 cls = cls_of(c[k]);
            const u32 kk = atomicAdd(&s_cur[cls], 1u);
            perm[p_lo + (kk < len ? kk * slices + g : len * slices + g)] = cls >= 6u ? q | 0x80000000u : q;
        }
    }
    return;
}

template <typename T, bool DIAG, typename I, bool DPP_SCAN = false, bool FAST_DELTA = false, bool CLIMB = false>
__global__ __launch_bounds__(C) void k_chunk(const u32 *__restrict__ gcodes, const u32 *__restrict__ ids,
                                             const T *__restrict__ coords, const T *__restrict__ radii,
                                             const T *__restrict__ packed,
                                             col_node *__restrict__ nodes, T *__restrict__ bounds,
                                             u32 *__restrict__ other_end, T *__restrict__ partial, u32 *__restrict__ cross,
                                             T *__restrict__ tab1, u32 n_bound, T block_k, const u32 *__restrict__ n_dev,
                                             typename ChunkDiag<DIAG>::T diag, u32 *__restrict__ walk_order = nullptr, int order_mode = 0,
                                             u32 *report_word = nullptr, u32 report_n = 0) {
    const int dbg = chunk_mode(diag);        // the constant 0 in the production instance
    const u32 n = count_of(n_bound, n_dev);  // (device-side count, col_common.h: the grid is sized for the bound)
    typedef typename BT<T>::V4 V4;
    __shared__ ChunkLds<T, CLIMB> lds;
    __shared__ u32 s_codes[WIN];
    const int tid = threadIdx.x, lane = tid & (COL_WAVE - 1), w = tid / COL_WAVE;
    // XCD-aware order: blockIdx % 8 is the XCD (each with its own L2).  Every XCD builds one contiguous run of chunks -- the
    // same eighth of the sorted leaves whose packets it walks in k_traverse, and neighbouring chunks read overlapping code
    // windows.  Same bits; whole path, interleaved A/B of two builds on one box: 1 M uniform 0.1706-0.1736 against
    // 0.1744-0.1757 ms, 2 M 0.293-0.296 against 0.298-0.301, 16 M and config 3 unchanged; a clustered scene at 2 M loses
    // 2.5 % (the deep chunks of a cluster all go to one XCD; strips of 16 chunks going round the XCDs avoid that and
    // gain 0.4 % instead of 1.5 %: not taken).  col_debug_lbvh bit 5 restores chunk = blockIdx.
    // (with a device-side count the grid is sized for the bound: the first `nb` blocks -- dealt round-robin over the XCDs like
    // all blocks -- share the real chunks, the others leave; with the bound's chunk count in the formula two of eight XCDs sat idle
    // at a bound of 1.3 n: 70 instead of 51 us)
    // (CLIMB: with a walk order the first 8 x COL_ORDER_SLICES workgroups of the grid make it -- see order_packets)
    const u32 nord = CLIMB && walk_order ? 8u * COL_ORDER_SLICES : 0u;
    if (blockIdx.x < nord) {
        order_packets(blockIdx.x, n, n_bound, walk_order, order_mode);
        return;
    }
    // (CLIMB: the LSD plan's bucket report -- how clustered are the codes, for the caller's next choice of plan -- is the LAST
    // workgroup of the grid: twenty dependent loads that used to be a launch of their own, 8.8 us on config 3)
    const u32 nrep = CLIMB && report_word ? 1u : 0u;
    if (nrep && blockIdx.x == gridDim.x - 1u) {
        col_bucket_report_block(gcodes, report_n, report_word, s_codes);
        return;
    }
    const u32 bid = blockIdx.x - nord;
    const u32 nb = n_dev ? (n + (u32)C - 1) / (u32)C : gridDim.x - nord - nrep;
    if (bid >= nb) return;            // (n == 0 included)
    u32 chunk = bid;
    if (!(dbg & 32)) {
        const u32 q = nb / 8, r = nb % 8, x = bid & 7u;
        chunk = x * q + (x < r ? x : r) + (bid >> 3);
    }
    const u32 c0 = chunk * C;
    const u32 p = c0 + tid;
    Codes<I, FAST_DELTA> codes = {gcodes, (LdsWord *)s_codes, (I)c0 - HALO, n};
    const bool valid = p < n;
    // Every independent load of the block first -- the three words of the code window and the leaf's id, at clamped
    // addresses so that none sits behind a branch -- and ONE wait.  As a loop with the LDS store inside, hipcc waited for
    // each window load before it issued the next: with the id and the row gather five dependent round trips per block,
    // now two.
    u32 wcode[WIN / C];
#pragma unroll
    for (int k = 0; k < WIN / C; k++) {
        const I j = codes.w0 + tid + k * C;
        const bool in = j >= 0 && j < (I)n;
        const u32 v = gcodes[in ? j : (I)0];
        wcode[k] = in ? v : 0u;
    }
    u32 id = ids[valid ? p : 0u];
    // Adjacent deltas (FAST_DELTA): delta(p - 1, p) and delta(p, p + 1) from two more coalesced loads issued with the
    // others.  Karras' direction and delta_min of node p are these two numbers (delta_min is the smaller one: three probes
    // less), and "is the right child that starts at leaf k an internal node" is adj[k] > adj[k - 1] (right_child_at: a
    // code read and two probes less, twice per thread) wherever k - 1 and k lie in this chunk.
    int d_prev = -1, d_next = -1;
    if constexpr (FAST_DELTA && sizeof(I) == 4) {
        const bool has_prev = valid && p >= 1u, has_next = p + 1u < n;
        const u32 cprev = gcodes[has_prev ? p - 1u : 0u], cnext = gcodes[has_next ? p + 1u : 0u];
        const u32 cme = wcode[HALO / C];
        if (has_prev) d_prev = (int)min(ffbh_raw(cme ^ cprev), 32u + ffbh_raw(p ^ (p - 1u)));
        if (has_next) d_next = (int)min(ffbh_raw(cme ^ cnext), 32u + ffbh_raw(p ^ (p + 1u)));
        lds.adj[1 + tid] = (signed char)d_next;
        if (tid == 0) lds.adj[0] = (signed char)d_prev;
    }
    if constexpr (CLIMB) { if (tid < C / 2) lds.flags[tid] = 0; lds.oe[tid] = (unsigned char)tid; }
#pragma unroll
    for (int k = 0; k < WIN / C; k++) s_codes[tid + k * C] = wcode[k];
    if constexpr (!CLIMB) lds.flags[tid] = 0;
    if (tid < (int)CROSS_CAP) cross[(uint64_t)chunk * CROSS_CAP + tid] = END;       // (complete before the barrier below: the fence of __syncthreads)
    if (tid == 0) { lds.ncross = 0; if constexpr (!CLIMB) lds.flags[C] = 1u; }
    const u32 leaf_start = n - 1;

    // leaves: collision.cl:55-63 (fillInternal) + collision.cl:128-141 (leafBounds)
    Box<T> leaf = box_empty<T>();
    if (!valid) id = 0;
    if (valid) {
        const u32 gid = (dbg & 1) ? p : id;
        V4 c;
        T r;
        if (packed) {                 // (x, y, z, r) rows written by col_morton_ex: one gather per leaf
            if (dbg & 16) {               // (timing ablation: the gather as a non-temporal load)
                typedef T vec4 __attribute__((ext_vector_type(4)));
                const vec4 v = __builtin_nontemporal_load(reinterpret_cast<const vec4 *>(packed) + gid);
                c.x = v.x; c.y = v.y; c.z = v.z; c.w = v.w;
            } else c = reinterpret_cast<const V4 *>(packed)[gid];
            r = c.w;
        } else {
            c = reinterpret_cast<const V4 *>(coords)[gid];
            r = radii[gid];
        }
        leaf.lo[0] = c.x - r; leaf.lo[1] = c.y - r; leaf.lo[2] = c.z - r;
        leaf.hi[0] = c.x + r; leaf.hi[1] = c.y + r; leaf.hi[2] = c.z + r;
        lds.leaf[0][tid] = c.x; lds.leaf[1][tid] = c.y; lds.leaf[2][tid] = c.z; lds.leaf[3][tid] = r;      // (a place beyond n is never read)
    }

    // inclusive prefix / suffix unions over the chunk (wave shuffles, then the 4 wave totals)
    Box<T> pre = leaf, suf = leaf;
    if constexpr (DPP_SCAN) {          // (every lane of the block is still here: full waves)
        wave_prefix_union(pre);
        wave_suffix_union(suf);
    } else {
#pragma unroll
        for (int o = 1; o < COL_WAVE; o <<= 1) {
            const Box<T> up = box_shfl_up(pre, o), dn = box_shfl_down(suf, o);
            if (lane >= o) box_merge(pre, up);
            if (lane + o < COL_WAVE) box_merge(suf, dn);
        }
    }
    if (lane == COL_WAVE - 1) { for (int k = 0; k < 3; k++) { lds.wave_tot[0][w][k] = pre.lo[k]; lds.wave_tot[0][w][3 + k] = pre.hi[k]; } }
    if (lane == 0) { for (int k = 0; k < 3; k++) { lds.wave_tot[1][w][k] = suf.lo[k]; lds.wave_tot[1][w][3 + k] = suf.hi[k]; } }
    __syncthreads();       // codes window, leaf boxes, ready flags and wave totals are in LDS
    for (int ww = 0; ww < C / COL_WAVE; ww++) {
        Box<T> t;
        for (int k = 0; k < 3; k++) { t.lo[k] = lds.wave_tot[ww < w ? 0 : 1][ww][k]; t.hi[k] = lds.wave_tot[ww < w ? 0 : 1][ww][3 + k]; }
        if (ww < w) box_merge(pre, t);
        if (ww > w) box_merge(suf, t);
    }

    if (valid) {
        LeafTail tail = {p, id};
        *reinterpret_cast<LeafTail *>(&nodes[leaf_start + p].right_edge) = tail;
        u32 skip_leaf = END;
        if (p + 1 < n) {
            if constexpr (FAST_DELTA && sizeof(I) == 4) {
                const u32 k = p + 1;
                if (k + 1 >= n) skip_leaf = (n - 1) + k;
                else if (tid + 1 < C) skip_leaf = lds.adj[tid + 2] > d_next ? k : (n - 1) + k;
                else skip_leaf = right_child_at(codes, k);            // (the chunk's last leaf: adj[k] belongs to the next chunk)
            } else skip_leaf = right_child_at(codes, p + 1);
        }
        record_store(bounds, (uint64_t)leaf_start + p, leaf, skip_leaf, id);
    }
    if (tid == 0)          // chunk total -> level 0 of the first group table
        box_store(tab1, ((uint64_t)(chunk / C) * LV + 0) * C + (chunk % C), suf);
    if (p == n - 1 && n > (u32)C) box_store(partial, p, pre);      // prefix of the last leaf
    // CLIMB (round 4): the in-chunk nodes bottom-up instead of Karras' searches.  For sorted keys delta(i, j) is the minimum of the
    // adjacent deltas between i and j (equal codes included: their deltas 32 + clz(i ^ j) are the common prefix of the keys
    // (code, position)), so the tree is the Cartesian tree of adj[]: a finished range [l, r] is the LEFT child of its parent --
    // Karras' node r -- if delta(r, r + 1) > delta(l - 1, l), else the right child -- node l --, and the two children of the node
    // that splits behind position g meet at meet[g]: the first leaves its far end there and stops, the second takes over the
    // union and goes on (Apetrei 2014, inside one chunk and in LDS).  One round costs what ONE of the search's probes did, a
    // thread climbs as far as it arrives second, and the boxes are merged on the way: no search (22 probes per wave), no waiting
    // for children.  A range whose sibling lies outside the chunk stops; its parent crosses the boundary and is found by the
    // search below, like every node that the climb has not named (oe[t] == t).
    if constexpr (CLIMB) {
        if (valid) {
            int l = tid, r = tid, my_split = 0;
            bool is_leaf = true;
            Box<T> box = leaf;
            for (;;) {
                const int dl = lds.adj[l], dr = lds.adj[r + 1];
                const bool right = dr > dl;
                if (!is_leaf) {
                    const int node = right ? r : l;
                    soa_put(lds.node, node, box);
                    lds.oe[node] = (unsigned char)(right ? l : r);
                    lds.split[node] = (unsigned char)my_split;
                }
                const int g = right ? r : l - 1;
                if ((dl & dr) < 0 || g < 0 || g >= C - 1) break;       // the root; or the sibling lies outside the chunk
                // (LDS instructions of one wave are carried out in order: the box is in lds.node before the exchange, and the reads
                // below come after it -- the compiler only has to keep that order)
                asm volatile("" ::: "memory");
                const u32 place = 16u * ((u32)g & 1u);
                const u32 old = (__hip_atomic_fetch_or(&lds.flags[g >> 1], ((u32)(right ? l : r) + 1u) << place, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> place) & 0xFFFFu;
                asm volatile("" ::: "memory");
                if (old == 0) break;                                    // first: the sibling will take it from here
                const int far = (int)old - 1;
                Box<T> sib;
                if (right) { sib = far == g + 1 ? leaf_get(lds.leaf, g + 1) : soa_get(lds.node, g + 1); r = far; }
                else { sib = far == g ? leaf_get(lds.leaf, g) : soa_get(lds.node, g); l = far; }
                box_merge(box, sib);
                my_split = g;
                is_leaf = false;
            }
        }
        __syncthreads();
    }
    // Karras' search (collision.cl:81-121) for node c0 + t: its other end and its split
    auto search = [&](u32 t, u32 &j, u32 &gamma) {
        const u32 i = c0 + t, ci = codes.win[HALO + t];       // (= codes.at(i): the chunk's own codes are always in the window)
        int dir, delta_min;
        if constexpr (FAST_DELTA && sizeof(I) == 4) {
            const int dn = lds.adj[t + 1], dp = lds.adj[t];   // delta(i, i + 1), delta(i - 1, i)
            dir = dn > dp ? 1 : -1;
            delta_min = min(dn, dp);                           // = delta(i, i - dir)
        } else {
            dir = delta(codes, i, ci, (I)i + 1) > delta(codes, i, ci, (I)i - 1) ? 1 : -1;
            delta_min = delta(codes, i, ci, (I)i - dir);
        }
        I len_max = 2;
        while (delta(codes, i, ci, (I)i + dir * len_max) > delta_min) len_max *= 2;
        I len = 0;
        for (I t2 = len_max / 2; t2 > 0; t2 /= 2)
            if (delta(codes, i, ci, (I)i + dir * (len + t2)) > delta_min) len += t2;
        j = (u32)((I)i + dir * len);
        const int delta_node = delta(codes, i, ci, (I)j);
        I s = 0, t2 = len;
        do {
            t2 = (t2 + 1) / 2;
            if (delta(codes, i, ci, (I)i + dir * (s + t2)) > delta_node) s += t2;
        } while (t2 > 1);
        gamma = dir > 0 ? (u32)(i + s) : (u32)(i - s - 1);
    };
    // node c0 + t with its other end j and its split: the node record's tail, the children's parent words, the links; returns
    // the skip link and the left child (collision.cl:104-120)
    auto node_out = [&](u32 t, u32 j, u32 gamma, u32 &skip, u32 &child_a) {
        const u32 i = c0 + t;
        const u32 lo = min(i, j), hi = max(i, j);
        const bool a_leaf = lo == gamma, b_leaf = hi == gamma + 1;
        child_a = a_leaf ? leaf_start + gamma : gamma;
        const u32 child_b = b_leaf ? leaf_start + gamma + 1 : gamma + 1;
        InnerTail tail = {hi, child_a, child_b};
        *reinterpret_cast<InnerTail *>(&nodes[i].right_edge) = tail;
        if (!(dbg & 2)) { nodes[child_a].parent = i; nodes[child_b].parent = i; }
        other_end[i] = j;
        skip = END;
        if (hi + 1 < n) {
            if constexpr (FAST_DELTA && sizeof(I) == 4) {
                const u32 k = hi + 1;
                if (k + 1 >= n) skip = (n - 1) + k;
                else if (hi >= c0 && k < c0 + (u32)C) skip = lds.adj[k - c0 + 1] > lds.adj[hi - c0 + 1] ? k : (n - 1) + k;
                else {
                    // (outside the chunk: the three codes in ONE round trip -- right_child_at's code read and two probes are three)
                    const u32 ck = gcodes[k], cm = gcodes[k - 1u], cp = gcodes[k + 1u];
                    const u32 d_fwd = min(ffbh_raw(ck ^ cp), 32u + ffbh_raw(k ^ (k + 1u))), d_bwd = min(ffbh_raw(ck ^ cm), 32u + ffbh_raw(k ^ (k - 1u)));
                    skip = d_fwd > d_bwd ? k : (n - 1) + k;
                }
            } else skip = right_child_at(codes, hi + 1);
        }
    };
    // a node whose range lies in this chunk: its record with the box, marked as a leaf block if it is small and dense
    auto record_out = [&](u32 i, u32 j, u32 skip, u32 child_a, const Box<T> &box) {
        // leaf block (col_common.h): a small DENSE node -- on every axis at most block_k times as wide as its first
        // leaf -- is marked for the packet walk.  Sparse scenes (BASELINE config 2: leaf boxes a fifth of the
        // spacing) get no marks and walk as before; where spheres overlap in heaps (config 3) nearly every node
        // of <= 16 leaves qualifies.
        const u32 lo = min(i, j), hi = max(i, j);
        u32 down = child_a;
        if (hi - lo < COL_LEAF_BLOCK && n <= COL_LEAF_BLOCK_MAX_N && block_k > (T)0) {
            const Box<T> first = leaf_get(lds.leaf, (int)(lo - c0));
            bool dense = true;
#pragma unroll
            for (int k = 0; k < 3; k++) dense = dense && (box.hi[k] - box.lo[k]) <= block_k * (first.hi[k] - first.lo[k]);
            if (dense) down = block_link(lo, hi);
        }
        record_store(bounds, (uint64_t)i, box, skip, down);
    };
    // a node that crosses the chunk's boundary leaves this chunk's half of its range -- the suffix from i (forward) or the prefix up
    // to i (backward) -- and its number for k_cross
    auto crossing_out = [&](u32 i, int dir) {
        box_store(partial, i, dir > 0 ? suf : pre);
        const u32 slot = atomicAdd(&lds.ncross, 1u);
        cross[(uint64_t)chunk * CROSS_CAP + (slot < CROSS_CAP - 1 ? slot : CROSS_CAP - 1)] = slot < CROSS_CAP - 1 ? i : CROSS_DENSE;
    };
    if constexpr (CLIMB) {
        // A node the climb has named: everything is known.  The others -- three in a hundred: their ranges cross the chunk's boundary --
        // leave their half box themselves (the direction is in adj[]); what they still need is Karras' search, and what that costs
        // is the ROUND TRIPS of the probes that leave the code window: a workgroup held its place on the CU while one lane asked
        // global memory 10..30 times in a row (with those probes answered "no" the launch was 12 us shorter at 1 M spheres).  So
        // a wave shares its lanes among its unnamed nodes -- 64, 32, 16 or 8 lanes per node -- and every step of the three searches
        // (each looks for the end of a run of "yes": delta(i, i + dir * x) > threshold is monotone in x) asks that many places at
        // once: the powers of two, then equal parts of what is left.  A fifth of the round trips, ~35 instructions per step for
        // the whole wave, no barrier below the climb's.
        // (Measured on the way: the searches compacted into the lanes of wave 0 behind two barriers: - 1 %; the same without
        // barriers, after wave 0's own output: + 15 %; 7 places per step in every lane of wave 0: + 9 % -- 300 instructions per step.)
        const bool inner = p < leaf_start;
        const bool named = inner && lds.oe[tid] != (unsigned char)tid;
        u32 skip, child_a;
        if (named) {
            const u32 j = c0 + lds.oe[tid];
            node_out((u32)tid, j, c0 + lds.split[tid], skip, child_a);
            record_out(p, j, skip, child_a, soa_get(lds.node, tid));
        } else if (inner) crossing_out(p, d_next > d_prev ? 1 : -1);
        const u64 todo = __builtin_amdgcn_ballot_w64(inner && !named);
        if (todo == 0) return;
        const u32 cnt = (u32)__builtin_popcountll(todo);
        const u32 lpn_log = cnt <= 1 ? 6u : cnt <= 2 ? 5u : cnt <= 4 ? 4u : 3u, lpn = 1u << lpn_log, groups = 64u >> lpn_log;
        const u32 g = (u32)lane >> lpn_log, sub = (u32)lane & (lpn - 1u);
        auto group_count = [&](bool yes) -> I {             // how many lanes of my group say yes
            u64 m = __builtin_amdgcn_ballot_w64(yes) >> (g << lpn_log);
            if (lpn < 64u) m &= (1ull << lpn) - 1ull;
            return (I)__builtin_popcountll(m);
        };
        for (u32 base = 0; base < cnt; base += groups) {
            const bool on = base + g < cnt;                  // my group has a node this round
            u64 m = todo;
            for (u32 b = 0; b < base + g && on; b++) m &= m - 1ull;
            const u32 t = on ? (u32)w * COL_WAVE + (u32)__builtin_ctzll(m) : (u32)tid;
            const u32 i = c0 + t, ci = codes.win[HALO + t];
            const int dn = lds.adj[t + 1], dp = lds.adj[t];   // delta(i, i + 1), delta(i - 1, i)
            const int dir = dn > dp ? 1 : -1, delta_min = min(dn, dp);
            // is place x (0 = none) in the run?  one load per lane; LDS if every lane's place is in the window
            auto yes_at = [&](I x, int thr) -> bool {
                const I jk = (I)i + dir * x;
                const bool inb = on && x > 0 && (u32)jk < n;
                const u32 o = (u32)(jk - codes.w0);
                const bool inwin = o < (u32)WIN;
                u32 cj;
                if (__builtin_amdgcn_ballot_w64(inb && !inwin) == 0) cj = codes.win[inwin ? o : 0u];
                else cj = inb ? (inwin ? codes.win[o] : gcodes[jk]) : 0u;
                const u32 d = min(ffbh_raw(ci ^ cj), 32u + ffbh_raw(i ^ (u32)jk));
                return inb && (int)d > thr;
            };
            // the first power of two that is not in the range any more
            I hi = 2;
            {
                bool more = on;
                int e0 = 1;
                while (__builtin_amdgcn_ballot_w64(more)) {
                    const int e = e0 + (int)sub;
                    const I x = (more && e < 31 && (1u << e) <= n) ? (I)(1u << e) : (I)0;
                    const I yes = group_count(yes_at(x, delta_min));
                    if (more) {
                        hi = (I)(1u << (e0 + yes));
                        more = yes == (I)lpn && e0 + (int)lpn < 31;
                        e0 += (int)lpn;
                    }
                }
            }
            // the end of a run of yes between lo (in) and hi (out)
            auto run_end = [&](I lo, I hi, int thr) -> I {
                while (__builtin_amdgcn_ballot_w64(on && hi - lo > 1)) {
                    const bool act = on && hi - lo > 1;
                    const I step = (hi - lo + (I)lpn) / ((I)lpn + 1);
                    const I x = lo + ((I)sub + 1) * step;
                    const I yes = group_count(yes_at(act && x < hi ? x : (I)0, thr));
                    if (act) { lo += yes * step; hi = min(hi, lo + step); }
                }
                return lo;
            };
            const I len = run_end(hi / 2, hi, delta_min);
            const u32 j = (u32)((I)i + dir * len);
            const int delta_node = delta(codes, i, ci, (I)j);
            const I s = run_end((I)0, len, delta_node);
            const u32 gamma = dir > 0 ? (u32)(i + s) : (u32)(i - s - 1);
            if (on && sub == 0) {
                node_out(t, j, gamma, skip, child_a);
                links_store(bounds, (uint64_t)i, skip, child_a);
            }
        }
        return;
    }
    if (p >= leaf_start || (dbg & 4)) return;                      // no barrier below this line
    u32 j = 0, gamma = 0, skip, child_a;
    search((u32)tid, j, gamma);
    const u32 i = p;
    const u32 lo = min(i, j), hi = max(i, j);
    node_out((u32)tid, j, gamma, skip, child_a);
    if (dbg & 8) return;
    if (lo >= c0 && hi < c0 + C) {
        // Both children live in this chunk.  Wait (in LDS, workgroup scope) until their boxes are
        // final, merge, publish.  The tree is at most 64 levels deep, so this ends after <= 64
        // rounds; finished lanes idle at the loop exit while the others retry.
        const bool a_leaf = lo == gamma, b_leaf = hi == gamma + 1;
        const int la = (int)(gamma - c0), lb = la + 1;
        Box<T> box;
        const int fa = a_leaf ? C : la, fb = b_leaf ? C : lb;      // both flags are read every round, unconditionally
        for (bool done = false; !done;) {
            // (relaxed loads + ONE acquire fence on success: two acquire loads are two serial LDS round trips per round)
            const u32 ra = __hip_atomic_load(&lds.flags[fa], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const u32 rb = __hip_atomic_load(&lds.flags[fb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ra & rb) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                box = a_leaf ? leaf_get(lds.leaf, la) : soa_get(lds.node, la);
                box_merge(box, b_leaf ? leaf_get(lds.leaf, lb) : soa_get(lds.node, lb));
                soa_put(lds.node, tid, box);
                __hip_atomic_store(&lds.flags[tid], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                done = true;
            }
        }
        record_out(i, j, skip, child_a, box);
    } else {
        links_store(bounds, (uint64_t)i, skip, child_a);
        crossing_out(i, j > i ? 1 : -1);
    }
}

// Sparse table over `count` level-0 entries, 256 per group; group totals go to level 0 of `next`.
template <typename T>
__global__ __launch_bounds__(C) void k_group(T *__restrict__ tab, u32 count, T *__restrict__ next, u32 n_bound, const u32 *__restrict__ n_dev,
                                             int level) {
    if (n_dev) {                            // device-side count: this level's real number of entries (chunks, groups, ...)
        u32 c = count_of(n_bound, n_dev);
        for (int k = 0; k <= level; k++) c = (c + C - 1) / C;
        count = min(count, c);
    }
    __shared__ Table<T> t;
    const int tid = threadIdx.x;
    const u32 g = blockIdx.x;
    const u32 e = g * C + tid;
    const uint64_t base = (uint64_t)g * LV * C;
    Box<T> b = e < count ? box_load(tab, base + tid) : box_empty<T>();
    if (e >= count) box_store(tab, base + tid, b);
    tab_put(t, 0, tid, b);
    tab_build(t, tid);
    for (int l = 1; l < LV; l++) box_store(tab, base + (uint64_t)l * C + tid, tab_get(t, l, tid));
    if (tid == 0 && next) box_store(next, ((uint64_t)(g / C) * LV + 0) * C + (g % C), tab_get(t, LV - 1, 0));
}

struct Tabs { void *t[3]; };

// `lin`: the table level whose sparse table was not built because it has at most LIN entries (one
// launch less on inputs up to 2 M leaves); a range on that level is a short scan of its level-0 row.
constexpr u32 LIN = 32;

template <typename T>
__device__ __forceinline__ void cross_node(T *__restrict__ bounds, const u32 *__restrict__ other_end, const T *__restrict__ partial,
                                           const Tabs &tabs, u32 i, int lin) {
    const u32 j = other_end[i];
    const u32 first = min(i, j), last = max(i, j);
    int64_t a = first / C, b = last / C;
    if (a == b) return;
    Box<T> box = box_load(partial, first);
    box_merge(box, box_load(partial, last));
    a += 1; b -= 1;                      // whole chunks strictly between the two ends
    for (int h = 0; h < 3 && a <= b; h++) {
        const T *tab = (const T *)tabs.t[h];
        if (h == lin) {                  // single group, level 0 only: entry e sits at index e
            for (int64_t e = a; e <= b; e++) box_merge(box, box_load(tab, (uint64_t)e));
            break;
        }
        const int64_t ga = a / C, gb = b / C;
        if (ga == gb) {
            box_merge(box, gtab_query(tab, (uint64_t)ga, (int)(a % C), (int)(b % C)));
            break;
        }
        box_merge(box, gtab_query(tab, (uint64_t)ga, (int)(a % C), C - 1));
        box_merge(box, gtab_query(tab, (uint64_t)gb, 0, (int)(b % C)));
        a = ga + 1; b = gb - 1;
    }
    // links (lane w) were written by k_chunk: store xyz only
    T *row = bounds + 8ull * i;
    row[0] = box.lo[0]; row[1] = box.lo[1]; row[2] = box.lo[2];
    row[4] = box.hi[0]; row[5] = box.hi[1]; row[6] = box.hi[2];
}

// One thread per (chunk, list slot): CROSS_CAP threads per chunk instead of one per node -- a wave of the per-node version
// had one or two crossing nodes among its 64 and ran their table look-ups at that utilisation (76 us at 16 M spheres).
template <typename T>
__global__ __launch_bounds__(256) void k_cross(T *__restrict__ bounds, const u32 *__restrict__ other_end,
                                               const T *__restrict__ partial, const u32 *__restrict__ cross, Tabs tabs, u32 n,
                                               u32 nchunks, int lin, u32 *__restrict__ zero8, const u32 *__restrict__ n_dev,
                                               u32 *__restrict__ walk_order, u32 cross_blocks, int order_mode, u32 nzero) {
    const u32 n_bound = n;
    if (n_dev) { n = count_of(n, n_dev); nchunks = min(nchunks, (n + (u32)C - 1) / (u32)C); }
    if (blockIdx.x >= cross_blocks) {
        order_packets(blockIdx.x - cross_blocks, n, n_bound, walk_order, order_mode);
        return;
    }
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (zero8 && t < nzero) zero8[t] = 0;      // the packet counters (or the whole chunk header) of the traversal that follows (bvh.hip), no launch of their own
    const u32 chunk = t / CROSS_CAP, slot = t % CROSS_CAP;
    if (chunk >= nchunks) return;
    if (cross[(uint64_t)chunk * CROSS_CAP + CROSS_CAP - 1] == CROSS_DENSE) {
        for (u32 k = slot; k < (u32)C; k += CROSS_CAP) {
            const u32 i = chunk * C + k;
            if (i + 1 < n) cross_node(bounds, other_end, partial, tabs, i, lin);
        }
        return;
    }
    const u32 i = cross[(uint64_t)chunk * CROSS_CAP + slot];
    if (i != END) cross_node(bounds, other_end, partial, tabs, i, lin);
}

int g_dbg = 0;         // process-wide diagnostics switch (col_debug_lbvh); see include/collision_hip.h
float g_block_k = 3.0f;   // leaf-block density criterion (col_debug_leaf_blocks): node width <= k x leaf width; 0 = no marks
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    size_t other_end, partial, cross, tab[3], total;
    u32 count[3];    // level-0 entries of each table: chunks, groups, groups of groups
};

Layout layout(uint32_t n, int coord_bytes) {
    Layout L;
    const size_t entry = 8 * (size_t)coord_bytes;       // two vec4 rows
    size_t off = 0;
    L.other_end = off; off += align256((size_t)n * 4);
    L.partial = off;   off += align256((size_t)n * entry);
    u32 cnt = (u32)col_ceil_div(n, C);
    L.cross = off;     off += align256((size_t)cnt * CROSS_CAP * 4);
    for (int h = 0; h < 3; h++) {
        L.count[h] = cnt;
        const size_t groups = col_ceil_div(cnt, C);
        L.tab[h] = off; off += align256(groups * LV * C * entry);
        cnt = (u32)groups;
    }
    L.total = off;
    return L;
}

template <typename T>
int run(hipStream_t s, const u32 *codes, const u32 *ids, const T *coords, const T *radii, const T *packed,
        col_node *nodes, T *bounds, char *scratch, u32 n, u32 *zero8, const u32 *n_dev, u32 *walk_order, int order_mode,
        u32 nzero, u32 *report_word, u32 report_n) {
    const Layout L = layout(n, sizeof(T));
    u32 *other_end = (u32 *)(scratch + L.other_end);
    T *partial = (T *)(scratch + L.partial);
    u32 *cross = (u32 *)(scratch + L.cross);
    Tabs tabs;
    for (int h = 0; h < 3; h++) tabs.t[h] = scratch + L.tab[h];
    const u32 nchunks = L.count[0];
    bool ordered_by_chunk = false;       // the production k_chunk orders the traversal's packets itself; otherwise k_cross does
    // mode bit 10 (1024) alone is not a diagnostics mode: it selects the round-3 production instance -- shuffle scans, branchy
    // delta() -- for A/Bs (tools/lbvh_scan_ab.py)
    // (bit 11 (2048), likewise: float64 keeps the shuffle scans -- the A/B of the float64 DPP scans)
    // (bit 12 (4096): the traversal's walk order is used whatever the previous costs look like -- tests)
    if (g_dbg & ~(1024 | 2048 | 4096 | 8192 | 16384))          // (bit 14: 24 KB of unused LDS per workgroup -- half the resident workgroups: a timing experiment)
        k_chunk<T, true, int64_t><<<dim3(nchunks), dim3(C), 0, s>>>(codes, ids, coords, radii, packed, nodes, bounds, other_end, partial, cross,
                                                                    (T *)tabs.t[0], n, (T)g_block_k, n_dev, ChunkDiagOn{g_dbg & ~(1024 | 2048 | 4096 | 8192 | 16384)});
    else if (n < (1u << 30) && (g_dbg & 8192))         // (bit 13: Karras' searches for every node -- the A/B of the climb)
        k_chunk<T, false, int32_t, true, true><<<dim3(nchunks), dim3(C), 0, s>>>(codes, ids, coords, radii, packed, nodes, bounds, other_end, partial, cross,
                                                                                 (T *)tabs.t[0], n, (T)g_block_k, n_dev, ChunkDiagOff{});
    else if (n < (1u << 30) && !(g_dbg & 1024) && !((g_dbg & 2048) && sizeof(T) == 8))       // production
    {
        k_chunk<T, false, int32_t, true, true, true><<<dim3(nchunks + (walk_order ? 8u * COL_ORDER_SLICES : 0u) + (report_word ? 1u : 0u)), dim3(C), (g_dbg & 16384) ? 24576 : 0, s>>>(
            codes, ids, coords, radii, packed, nodes, bounds, other_end, partial, cross, (T *)tabs.t[0], n, (T)g_block_k, n_dev, ChunkDiagOff{}, walk_order, order_mode,
            report_word, report_n);
        ordered_by_chunk = true;
        report_word = nullptr;          // done in that launch
    }
    else if (n < (1u << 30) && !(g_dbg & 1024))
        k_chunk<T, false, int32_t, false, true><<<dim3(nchunks), dim3(C), 0, s>>>(codes, ids, coords, radii, packed, nodes, bounds, other_end, partial, cross,
                                                                                  (T *)tabs.t[0], n, (T)g_block_k, n_dev, ChunkDiagOff{});
    else if (n < (1u << 30))
        k_chunk<T, false, int32_t><<<dim3(nchunks), dim3(C), 0, s>>>(codes, ids, coords, radii, packed, nodes, bounds, other_end, partial, cross,
                                                                     (T *)tabs.t[0], n, (T)g_block_k, n_dev, ChunkDiagOff{});
    else
        k_chunk<T, false, int64_t><<<dim3(nchunks), dim3(C), 0, s>>>(codes, ids, coords, radii, packed, nodes, bounds, other_end, partial, cross,
                                                                     (T *)tabs.t[0], n, (T)g_block_k, n_dev, ChunkDiagOff{});
    COL_LAUNCH_OK();
    if (report_word) {                  // (an instance of k_chunk without the report's workgroup: the report's own launch)
        const int rc = col_radix_bucket_report((void *)s, codes, report_n, report_word);
        if (rc) return rc;
    }
    if (nchunks < 2) {                  // every node lives inside the single chunk: no k_cross, so clear the packet counters here
        if (zero8) COL_HIP(hipMemsetAsync(zero8, 0, nzero * sizeof(u32), s));
        return COL_OK;
    }
    int lin = -1;
    for (int h = 0; h < 3; h++) {
        if (h > 0 && L.count[h] <= LIN) { lin = h; break; }      // k_cross scans this level's few entries itself
        const u32 groups = (u32)col_ceil_div(L.count[h], C);
        k_group<T><<<dim3(groups), dim3(C), 0, s>>>((T *)tabs.t[h], L.count[h], h + 1 < 3 ? (T *)tabs.t[h + 1] : nullptr, n, n_dev, h);
        COL_LAUNCH_OK();
        if (groups < 2) break;
    }
    const unsigned cross_blocks = (unsigned)col_ceil_div((uint64_t)nchunks * CROSS_CAP, 256);
    k_cross<T><<<dim3(cross_blocks + (walk_order && !ordered_by_chunk ? 8u * COL_ORDER_SLICES : 0u)), dim3(256), 0, s>>>(bounds, other_end, partial, cross, tabs, n, nchunks, lin, zero8,
                                                                                                          n_dev, walk_order, cross_blocks, order_mode, nzero);
    COL_LAUNCH_OK();
    return COL_OK;
}

}  // namespace

extern "C" {

void col_debug_lbvh(int mode) { g_dbg = mode; }
int col_lbvh_order_forced(void) { return (g_dbg & 4096) ? 1 : 0; }
void col_debug_leaf_blocks(float k) { g_block_k = k; }

size_t col_lbvh_scratch_bytes(uint32_t n, int coord_bytes) { return layout(n, coord_bytes).total + 256; }

// `packed`: optional (x, y, z, r) rows indexed like coords (see col_morton_ex); coords/radii are then unused.
int col_lbvh_ex(void *stream, const uint32_t *codes, const uint32_t *ids, const void *coords, const void *radii,
                const void *packed, col_node *nodes, void *bounds, void *scratch, uint32_t n, int coord_bytes, uint32_t *zero8,
                const uint32_t *n_dev, uint32_t *walk_order, int order_mode, uint32_t nzero, uint32_t *report_word, uint32_t report_n) {
    if (n == 0) return COL_OK;
    if (n >= 0x80000000u || nzero > 256u) return COL_EINVAL;
    if (!scratch) return COL_ENOSCRATCH;
    if (coord_bytes == 4)
        return run<float>(col_stream(stream), codes, ids, (const float *)coords, (const float *)radii, (const float *)packed,
                          nodes, (float *)bounds, (char *)scratch, n, zero8, n_dev, walk_order, order_mode, nzero, report_word, report_n);
    if (coord_bytes == 8)
        return run<double>(col_stream(stream), codes, ids, (const double *)coords, (const double *)radii,
                           (const double *)packed, nodes, (double *)bounds, (char *)scratch, n, zero8, n_dev, walk_order, order_mode, nzero, report_word, report_n);
    return COL_EINVAL;
}

int col_lbvh(void *stream, const uint32_t *codes, const uint32_t *ids, const void *coords, const void *radii,
             col_node *nodes, void *bounds, void *scratch, uint32_t n, int coord_bytes) {
    return col_lbvh_ex(stream, codes, ids, coords, radii, nullptr, nodes, bounds, scratch, n, coord_bytes, nullptr, nullptr, nullptr, 0);
}

}  // extern "C"

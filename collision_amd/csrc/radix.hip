// Stable LSD radix sort of (key, value) pairs -- the hot loop of the path.
// Replaces RadixSorter.sort (collision/radix.py:118-170) and its kernels block_sort / scatter
// (collision/radix.cl:48-139).  Same result (a stable sort over ALL key bits), different shape:
//
//   reference, per 4-bit pass: block_sort (4 Blelloch LDS scans per block, in-place write
//     back) -> copy -> 3..5 scan launches -> scatter -> 2 copy-backs      = 48 B/pair, 8 passes
//   here, per 8-bit pass: k_hist (read keys) -> scan -> k_scatter (read pairs, rank in
//     registers with wave ballots, stage through LDS, write runs)          = 20 B/pair, 4 passes
//
// Histogram layout is the reference's: digit-major, hist[d * nblocks + b], so one flat
// exclusive scan yields every block's global offset per digit (radix.cl:99-100,127-136).
//
// col_radix_sort_msd (col_collide's second plan for small inputs of 30-bit Morton codes): one such
// global pass on the TOP 8 code bits, then k_bucket_sort finishes each of the 256 buckets in one
// workgroup's LDS -- see the comment above k_bucket_sort.
//
// k_scatter ranking (wave64): each wave owns a contiguous 64*IT slice of the tile and reads
// it lane-striped, so (wave, item, lane) order == memory order and stability is positional.
// For an item, the lanes holding the same digit are found with 8 __ballot()s ("match-any"),
// a lane's rank inside that group is mbcnt(peers), and the group's lowest lane bumps a
// wave-private LDS counter; no atomics, no cross-wave traffic until one block-wide scan of
// the 256 digit totals.  Keys (and 4/8-byte values) are then written into LDS at their
// tile-sorted position and streamed out so that consecutive lanes write consecutive
// addresses inside each digit run.
#include "col_common.h"
#include <algorithm>

namespace {

constexpr int RT = 256;             // threads per block
constexpr int RDIG = 256;           // 8-bit digits
constexpr int BS_SHIFT_REPORT = 22; // == BS_SHIFT below: the MSD plan's bucket digit is bits 22..29 of a code
// Tile classes (threads x items per thread), template parameters of the kernels:
//   SMALL 256 x 4  = 1024 pairs  below SMALL_N pairs: a pass is bound by the latency of one block, and
//                                more, shorter blocks finish sooner (the 1 M-sphere path);
//   MID   256 x 16 = 4096 pairs  up to BIG_N (8 Mi) pairs, and for 8-byte keys (LDS);
//   BIG   512 x 16 = 8192 pairs  above: the digit runs of a tile are ~128 bytes, a full L2 line, so the
//                                stores no longer depend on the runs of neighbouring tiles meeting in L2.
//   HUGE  512 x 32 = 16384 keys  from HUGE_N keys, u32 keys WITHOUT values (k_scatter_huge): ~256-byte runs -- one
//                                whole line and two partial ones instead of two partial ones per run.
constexpr int IT_BIG = 16, IT_SMALL = 4, IT_HUGE = 32;
constexpr int NT_BIG = 512, NT_MID = 256, NT_SMALL = 256, NT_HUGE = 512;
// COL_SCATTER_KEYS_FIRST = 1: k_scatter waits for its keys only and ranks them while the value loads are still in flight.
// Measured in round 4 (two builds alternating on one box, 3 x 100 launches each, EXPERIMENTS.md): 0.2556 ms per 64 Mi-pair
// pass against 0.2328 -- like the pass with its ranking taken away (round 3), a workgroup that reaches its store phase
// sooner makes the pass SLOWER.  Off.
#ifndef COL_SCATTER_KEYS_FIRST
#define COL_SCATTER_KEYS_FIRST 0
#endif
// tools/radix_tile_sweep.py.  BIG_N: whole (u32, u32) sorts with the 4096 / 8192 tile at 8 M pairs 0.1772 / 0.1736 ms, 12 M
// 0.2485 / 0.2367, 16 M 0.3352 / 0.3005 (round 3; it was 16 Mi, taken from a sweep over powers of two)
constexpr uint64_t SMALL_N = 1u << 20, BIG_N_DEFAULT = 8u << 20, HUGE_N = 32u << 20;
constexpr int HG = 16;              // max tiles per histogram block (64-byte rows of hist)

template <int B> struct Val;
template <> struct Val<4> { typedef uint32_t T; };
template <> struct Val<8> { typedef uint2 T; };
template <> struct Val<16> { typedef uint4 T; };
struct alignas(16) V32 { uint4 a, b; };
template <> struct Val<32> { typedef V32 T; };

template <typename K> __device__ __forceinline__ u32 digit_of(K key, int shift) { return (u32)(key >> shift) & (RDIG - 1); }

// Diagnostics live in their own template instances (DIAG = true), selected by col_debug_radix(mode != 0):
// the production instances take no mode argument and carry none of the branches below.
// (mode 32: cycles per phase of k_scatter, summed over blocks)
__device__ unsigned long long g_stamp[8];
struct DiagOff {};
struct DiagOn { int mode; };
template <bool DIAG> struct DiagArg { typedef DiagOff T; };
template <> struct DiagArg<true> { typedef DiagOn T; };
__device__ __forceinline__ constexpr int diag_mode(DiagOff) { return 0; }
__device__ __forceinline__ int diag_mode(DiagOn d) { return d.mode; }
#define STAMP(slot)                                                              \
    if (DIAG && (dbg & 32) && threadIdx.x == 0) {                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();              \
        atomicAdd(&g_stamp[slot], t_ - t_prev);                                  \
        t_prev = t_;                                                             \
    }

// lanes of this wave whose 8-bit digit equals mine ("match-any").  For every bit the ballot of
// that bit is XORed with my own bit replicated over the word: the result marks the lanes that
// DIFFER from me in that bit; the eight results are ORed (v_or3) and complemented.  12 VALU per
// 32-lane half instead of 16 for the and-chain formulation.
__device__ __forceinline__ u64 match8(u32 d) {
    // Three batches (masks, ballots, folds) instead of bit by bit: a v_cmp result needs two wait states
    // before a VALU instruction may read it as an SGPR operand, and seven other compares hide them.
    u32 m[8];
    u64 bal[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        m[b] = (u32)__builtin_amdgcn_sbfe(d, b, 1);                         // my bit, replicated
        // (opaque to the optimiser: it otherwise derives the comparison from d again -- shift + sign test --
        // instead of comparing the mask it already has: 5 instead of 4 VALU per bit)
        asm("" : "+v"(m[b]));
    }
#pragma unroll
    for (int b = 0; b < 8; b++) bal[b] = __builtin_amdgcn_ballot_w64(m[b] != 0);
    u32 dlo = 0, dhi = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        // acc | (ballot ^ mine) in one v_bitop3 per half (truth table 0xF6 for (acc, ballot, mine))
        dlo = __builtin_amdgcn_bitop3_b32(dlo, (u32)bal[b], m[b], 0xF6);
        dhi = __builtin_amdgcn_bitop3_b32(dhi, (u32)(bal[b] >> 32), m[b], 0xF6);
    }
    return ~(((u64)dhi << 32) | dlo);
}

// ---- histogram: blocks handle `g` consecutive tiles and write g-entry rows per digit ----
// Round 4.  The first version counted with LDS atomics on a plain 256-bin array: 64 lanes with ~57 distinct random digits
// hit 32 banks, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 70 % (profiles/r03_radix64M_pmc.json), and the kernel read at
// 4.7 TB/s where an in-order sweep reaches 6 (MI355X_MICROARCH.md).  Now every bin has HC = 16 COLUMNS, lane l counts in
// column l % 16 of its digit: the word address is d * 16 + (l % 16), i.e. bank 16 * (d % 2) + l % 16 -- the two lanes of a
// 32-lane group that share a column collide only when their digits have the same parity, which an LDS atomic (4 cycles of
// address + data transfer per wave-instruction) hides.  After a tile thread d folds its 16 columns (four 16-byte reads,
// rotated by d / 4 so that the 16 lanes of a read group cover all 64 banks) and clears them for the next tile.
// A wave-uniform digit (sorted inputs) puts 4 lanes on each of 16 addresses instead of 64 on one: no special case.
// (amdgpu_num_sgpr: above 80 SGPRs a wave's allocation -- with the 16 the trap handler reserves on this platform -- no
// longer lets 8 waves share a SIMD)
constexpr int HC = 16;              // columns per bin
template <typename K, int TILE>
__global__ __launch_bounds__(RT) __attribute__((amdgpu_num_sgpr(80))) void k_hist(const K *__restrict__ keys, uint64_t n, u32 nblocks, u32 g,
                                             int shift, u32 *__restrict__ hist) {
    constexpr int IT = TILE / RT;
    constexpr int VEC = 16 / sizeof(K);          // keys per 16-byte load
    constexpr int NV = IT / VEC;                 // 16-byte loads per thread and tile
    __shared__ __attribute__((aligned(16))) u32 col[RDIG * HC];      // 16 KB
    __shared__ u32 h[HG * RDIG];                                     // 16 KB: the block's g rows
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const u32 tid = threadIdx.x, c = tid & (HC - 1);
#pragma unroll
    for (int j = 0; j < HC / 4; j++) *reinterpret_cast<v4u *>(col + tid * HC + 4 * j) = v4u{0, 0, 0, 0};
    __syncthreads();
    const u32 b0 = blockIdx.x * g;
    const u32 cnt = min(g, nblocks - b0);
    // The next tile's keys are on their way while this tile is counted and folded (round 4: a workgroup used to issue a tile's
    // loads only after the previous tile's fold -- one exposed round trip per tile with four workgroups on a CU).  The loads
    // are asm statements so that hipcc leaves them where they are (it sinks plain loads below the LDS atomics), and every
    // destination passes through an explicit wait before it is copied or used, with NO control flow between an asm load and
    // its wait (cdna_hip_programming.md 5.7, form ii; k_scatter's lesson: hipcc does not know the registers are in flight and
    // places its copies where it likes): the steady-state loop body is one basic block, the last full tile is peeled.
    // Only the input's last tile can be ragged: it takes the guarded element-wise path.
    const u32 m = cnt && (uint64_t)(b0 + cnt) * TILE > n ? cnt - 1 : cnt;        // this block's full tiles: [0, m)
    auto issue = [&](v4u (&dst)[NV], uint64_t base) {
#pragma unroll
        for (int k = 0; k < NV; k++)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[k]) : "v"(keys + base + ((uint64_t)k * RT + tid) * VEC) : "memory");
    };
    auto arrived = [&](v4u (&dst)[NV]) {
#pragma unroll
        for (int k = 0; k < NV; k++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(dst[k])::"memory");
    };
    auto fold = [&](u32 t) {         // between two barriers: digit `tid` folds its columns into row t and clears them
        __syncthreads();
        u32 sum = 0;
#pragma unroll
        for (int j = 0; j < HC / 4; j++) {
            v4u *p = reinterpret_cast<v4u *>(col + tid * HC + 4 * ((j + (tid >> 2)) & (HC / 4 - 1)));
            const v4u v = *p;
            sum += v.x + v.y + v.z + v.w;
            *p = v4u{0, 0, 0, 0};
        }
        h[t * RDIG + tid] = sum;
        __syncthreads();
    };
    auto count = [&](const v4u (&src)[NV]) {
#pragma unroll
        for (int k = 0; k < NV; k++) {
            K kk[VEC];
            *reinterpret_cast<v4u *>(kk) = src[k];
#pragma unroll
            for (int j = 0; j < VEC; j++) atomicAdd(&col[digit_of(kk[j], shift) * HC + c], 1u);
        }
    };
    if (m) {
        v4u q[NV], qn[NV];
        issue(q, (uint64_t)b0 * TILE);
        arrived(q);
        for (u32 t = 0; t + 1 < m; t++) {
            if constexpr (NV <= 8) {
                issue(qn, (uint64_t)(b0 + t + 1) * TILE);
                count(q);
                fold(t);
                arrived(qn);
#pragma unroll
                for (int k = 0; k < NV; k++) q[k] = qn[k];
            } else {                 // (the 16384-key tile: two tiles of keys are 128 registers -- one tile at a time, as before)
                count(q);
                fold(t);
                issue(q, (uint64_t)(b0 + t + 1) * TILE);
                arrived(q);
            }
        }
        count(q);
        fold(m - 1);
    }
    if (m < cnt) {
        const uint64_t base = (uint64_t)(b0 + m) * TILE;
        for (int k = 0; k < IT; k++) {
            const uint64_t i = base + (uint64_t)k * RT + tid;
            if (i < n) atomicAdd(&col[digit_of(keys[i], shift) * HC + c], 1u);
        }
        fold(m);
    }
    // thread d writes its row of up to g entries (contiguous: full-sector writes)
    u32 *row = hist + (uint64_t)tid * nblocks + b0;
    for (u32 t = 0; t < cnt; t++) row[t] = h[t * RDIG + tid];
}

// ablation helper (col_debug_radix modes 128 / 256)
template <int SCOPE, typename X> __device__ __forceinline__ void st_scope(X *p, X v) {
    if constexpr (sizeof(X) == 4) __hip_atomic_store(reinterpret_cast<u32 *>(p), *reinterpret_cast<u32 *>(&v), __ATOMIC_RELAXED, SCOPE);
    else if constexpr (sizeof(X) == 8) __hip_atomic_store(reinterpret_cast<u64 *>(p), *reinterpret_cast<u64 *>(&v), __ATOMIC_RELAXED, SCOPE);
    else *p = v;
}

// ---- scatter ----
template <bool NARROW> struct CntType { typedef u32 T; };
template <> struct CntType<true> { typedef uint16_t T; };
template <typename K, int VB, int IT, int NT, bool DIAG>
__global__ __launch_bounds__(NT) void k_scatter(const K *__restrict__ keys_in, K *__restrict__ keys_out,
                                                const void *__restrict__ vals_in_, void *__restrict__ vals_out_,
                                                uint64_t n, u32 nblocks, int shift,
                                                const u32 *__restrict__ offsets, typename DiagArg<DIAG>::T diag) {
    const int dbg = diag_mode(diag);        // the constant 0 in the production instance
    constexpr int TILE = NT * IT;
    constexpr int NW = NT / COL_WAVE;
    constexpr bool HAS_V = VB > 0;
    constexpr bool V_LDS = VB == 4 || VB == 8;       // small values are staged through LDS
    typedef typename Val<(VB > 0 ? VB : 4)>::T V;
    __shared__ __attribute__((aligned(16))) K s_keys[TILE];
    __shared__ __attribute__((aligned(16))) V s_vals[V_LDS ? TILE : 1];
    // (wave, digit) counters: at most 64 * IT items per wave and TILE per block, so 16 bits do when a block has 16 waves --
    // with 32-bit counters the 1024-thread instance would not fit two blocks into a CU's LDS
    typedef typename CntType<(NT > 512)>::T CNT;
    __shared__ CNT s_cnt[NW][RDIG];
    __shared__ u32 s_goff[RDIG];
    __shared__ u32 s_ws[NW];
    __shared__ u32 s_dstart[DIAG ? RDIG : 1];        // (diagnostics, mode 8192)

    const V *vals_in = reinterpret_cast<const V *>(vals_in_);
    V *vals_out = reinterpret_cast<V *>(vals_out_);
    const u32 tid = threadIdx.x, lane = tid & (COL_WAVE - 1), w = tid / COL_WAVE;
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an
    // XCD), each with its own L2.  Consecutive tiles write ADJACENT runs of every digit, so give each
    // XCD a contiguous range of tiles: the ~64-byte runs of neighbouring tiles then meet in one L2 and
    // leave it as full lines instead of partial-line writes from 8 different L2s.  (Speed only.)
    u32 b = blockIdx.x;
    if (DIAG && (dbg & (1 << 22)) && nblocks % (8u << ((dbg >> 24) & 7)) == 0) {
        // (diagnostics, mode 1 << 22: STRIPS of G = 1 << ((mode >> 24) & 7) consecutive tiles go round the XCDs instead of one
        // contiguous range per XCD: the resident workgroups then cover ONE window of consecutive tiles, i.e. one contiguous
        // piece of every digit's output region, and only every G-th pair of neighbouring tiles meets in different L2s)
        const u32 G = 1u << ((dbg >> 24) & 7), x = b % 8, j = b / 8;
        b = ((j / G) * 8 + x) * G + (j % G);
    } else if (!(dbg & 4)) {
        const u32 q = nblocks / 8, r = nblocks % 8, xcd = b % 8;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    const uint64_t tile_base = (uint64_t)b * TILE;
    const u32 valid = (u32)min((uint64_t)TILE, n - tile_base);

    unsigned long long t_prev = (DIAG && (dbg & 32)) ? __builtin_amdgcn_s_memtime() : 0ull;
    (void)t_prev;
    // (diagnostics, wave priorities by phase -- modes 1 << 23: the store phase raised; 1 << 27: the load phase raised;
    // 1 << 28: the ranking raised; 1 << 29: an s_sleep between the store phase's steps.  EXPERIMENTS.md R4.14)
    if (DIAG && (dbg & (1 << 27))) __builtin_amdgcn_s_setprio(3);
    // this tile's global offset of digit `tid`: one scattered 4-byte load per thread, issued now so that
    // its latency hides under the loads and the ranking instead of sitting between two barriers
    const u32 my_offset = tid < RDIG ? offsets[(uint64_t)tid * nblocks + b] : 0u;
    for (u32 i = tid; i < NW * RDIG; i += NT) (&s_cnt[0][0])[i] = (CNT)0;

    const u32 wbase = w * (COL_WAVE * IT) + lane;
    K key[IT];
    V val[V_LDS ? IT : 1];
    constexpr int KV = 16 / sizeof(K);                    // keys per 16-byte load
    constexpr int VV = 16 / sizeof(V);
    constexpr int NVQ = V_LDS ? IT / VV : 0;              // 16-byte value loads per thread
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    v4u vq[V_LDS ? IT / VV : 1];
    V *vstage = s_vals + w * (COL_WAVE * IT);
    const bool full = valid == (u32)TILE && !(dbg & 8);
    if (full) {
        // Full tile: 16-byte global loads (lane l takes KV consecutive keys), transposed to the
        // lane-striped order the ranking needs through this wave's own slice of the LDS staging
        // area.  LDS operations of one wave execute in order, so no barrier is needed.
        // All global loads of the tile (keys AND values) are issued before the first wait: one memory
        // round trip per tile instead of two.  Round 4: the wave then waits for its KEYS only -- vmcnt counts in
        // issue order, the NVQ value loads were issued after them -- and ranks them while the values are still on
        // their way; the values are waited for, transposed and taken into registers after the ranking.
        K *stage = s_keys + w * (COL_WAVE * IT);
        const K *src = keys_in + tile_base + w * (COL_WAVE * IT);
        const V *vsrc = vals_in + tile_base + w * (COL_WAVE * IT);
        // The loads are asm statements: hipcc sinks a plain second group of loads below the first
        // group's LDS writes (to save registers), which serialises two round trips.  hipcc does not
        // count asm loads, so every destination passes through an explicit vmcnt statement before
        // its first use (cdna_hip_programming.md 5.7, form ii).
        v4u kq[IT / KV];
        if (dbg & 64) {                          // timing ablation: non-temporal (streaming) loads
#pragma unroll
            for (int j = 0; j < IT / KV; j++)
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(kq[j]) : "v"(src + j * (COL_WAVE * KV) + lane * KV) : "memory");
            if (V_LDS) {
#pragma unroll
                for (int j = 0; j < IT / VV; j++)
                    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(vq[j]) : "v"(vsrc + j * (COL_WAVE * VV) + lane * VV) : "memory");
            }
        } else {
#pragma unroll
            for (int j = 0; j < IT / KV; j++)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kq[j]) : "v"(src + j * (COL_WAVE * KV) + lane * KV) : "memory");
            if (V_LDS) {
#pragma unroll
                for (int j = 0; j < IT / VV; j++)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vq[j]) : "v"(vsrc + j * (COL_WAVE * VV) + lane * VV) : "memory");
            }
        }
        // (NO control flow between an asm load and its wait: with a run-time choice between two waits here hipcc placed
        // register copies of the load destinations in front of the waits -- it does not know they are in flight -- and the
        // kernel ranked garbage.  COL_SCATTER_KEYS_FIRST = 0 builds the round-3 behaviour for A/Bs: tools/ab_radix.sh.)
#pragma unroll
        for (int j = 0; j < IT / KV; j++) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(kq[j]) : "n"(COL_SCATTER_KEYS_FIRST ? NVQ : 0) : "memory");
#pragma unroll
        for (int j = 0; j < IT / KV; j++)
            *reinterpret_cast<v4u *>(stage + j * (COL_WAVE * KV) + lane * KV) = kq[j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < IT; k++) key[k] = stage[k * COL_WAVE + lane];
    } else {
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const u32 li = wbase + k * COL_WAVE;
            key[k] = li < valid ? keys_in[tile_base + li] : (K)~(K)0;   // padding sorts last, never stored
        }
        if (V_LDS) {
#pragma unroll
            for (int k = 0; k < IT; k++) {
                const u32 li = wbase + k * COL_WAVE;
                if (li < valid) val[k] = vals_in[tile_base + li];
            }
        }
        // These loads must be waited for HERE, inside the branch.  Left pending, they reach the join with the full-tile
        // path as "registers that may still be loading", and hipcc then puts an s_waitcnt vmcnt(0) in front of the
        // full-tile path's first write to any of them -- which (read in the ISA, round 4) made every full tile wait for
        // the round trip of `my_offset` BEFORE issuing its loads -- and another one behind the join, which would undo the
        // overlap of the ranking with the value loads.  An empty asm that "uses" each register makes hipcc wait here.
#pragma unroll
        for (int k = 0; k < IT; k++) asm volatile("" : "+v"(key[k]));
        if (V_LDS) {
#pragma unroll
            for (int k = 0; k < IT; k++) {
                u32 *words = reinterpret_cast<u32 *>(&val[k]);
#pragma unroll
                for (int c = 0; c < (int)(sizeof(V) / 4); c++) asm volatile("" : "+v"(words[c]));
            }
        }
    }
    if (DIAG && (dbg & (1 << 27))) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    STAMP(0)      // load + transpose
    if (DIAG && (dbg & (1 << 28))) __builtin_amdgcn_s_setprio(3);

    // (diagnostics, mode 65536: the input IS the pass's tile-sorted image (mode 2 wrote it) -- an element's slot is its
    // own index, the digit counts come from the offsets table: no match-any, no counters.  Everything else -- loads,
    // LDS staging, the stores and their addresses -- is the production code: the kernel's ceiling without its ranking.)
    const bool norank = DIAG && (dbg & 65536);
    // rank inside (wave, digit): wave-private counters, program order keeps them consistent
    u32 pos[IT];
    // (diagnostics, mode 1 << 21: the rank from ONE returning LDS atomic per item instead of match-any + counter.  Stable
    // only if the LDS resolves lanes that hit the same address in lane order -- not an architectural promise, so an
    // experiment: tools/radix_atomrank_ab.py compares its output with the production pass.)
    const bool atomrank = DIAG && sizeof(CNT) == 4 && (dbg & (1 << 21));
    if (atomrank) {
        if constexpr (sizeof(CNT) == 4) {
#pragma unroll
            for (int k = 0; k < IT; k++) pos[k] = atomicAdd(reinterpret_cast<u32 *>(&s_cnt[w][digit_of(key[k], shift)]), 1u);
        }
    } else if (!norank) {
#pragma unroll
    for (int k = 0; k < IT; k++) {
        const u32 d = digit_of(key[k], shift);
        const u64 peers = match8(d);
        const u32 below = mbcnt(peers);
        // every lane of the group reads the counter (same address: a broadcast), then the group's
        // lowest lane bumps it; LDS operations of one wave execute in order, so no lane sees the bump
        const u32 prev = s_cnt[w][d];
        if (below == 0) s_cnt[w][d] = (CNT)(prev + (u32)__popcll(peers));
        pos[k] = prev + below;
    }
    }
    if (V_LDS && full) {
        // the values: in flight since the top of the kernel, through this wave's slice of the value image
#pragma unroll
        for (int j = 0; j < IT / VV; j++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(vq[j])::"memory");
#pragma unroll
        for (int j = 0; j < IT / VV; j++)
            *reinterpret_cast<v4u *>(vstage + j * (COL_WAVE * VV) + lane * VV) = vq[j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < IT; k++) val[k] = vstage[k * COL_WAVE + lane];
    }
    if (DIAG && (dbg & (1 << 28))) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    STAMP(1)      // ranking

    if (norank) {
#pragma unroll
        for (int k = 0; k < IT; k++) pos[k] = wbase + k * COL_WAVE;
        u32 cnt = 0;
        if (tid < RDIG) {
            const uint64_t flat = (uint64_t)tid * nblocks + b;
            cnt = (flat + 1 < (uint64_t)RDIG * nblocks ? offsets[flat + 1] : (u32)n) - my_offset;
        }
        u32 total;
        const u32 dstart = block_excl_scan<NT>(cnt, s_ws, &total);
        if (tid < RDIG) {
            s_goff[tid] = my_offset - dstart;
            for (int i = 0; i < NW; i++) s_cnt[i][tid] = 0;
            if (DIAG) s_dstart[tid] = dstart;
        }
    } else
    // digit `tid` (threads 0..255): exclusive over waves, then exclusive over digits; fold both into s_cnt
    {
        u32 c[NW], tot = 0;
        if (tid < RDIG) {
#pragma unroll
            for (int i = 0; i < NW; i++) { c[i] = s_cnt[i][tid]; tot += c[i]; }
        }
        u32 total;
        const u32 dstart = block_excl_scan<NT>(tot, s_ws, &total);      // threads >= 256 contribute 0
        if (tid < RDIG) {
            u32 run = dstart;
#pragma unroll
            for (int i = 0; i < NW; i++) { s_cnt[i][tid] = (CNT)run; run += c[i]; }
            // global position of tile-sorted slot i with digit d is s_goff[d] + i
            s_goff[tid] = my_offset - dstart;
            if (DIAG) s_dstart[tid] = dstart;
        }
    }
    __syncthreads();
    STAMP(2)      // digit scan + offsets

#pragma unroll
    for (int k = 0; k < IT; k++) {
        const u32 d = digit_of(key[k], shift);
        pos[k] += s_cnt[w][d];
        s_keys[pos[k]] = key[k];
        if (V_LDS) s_vals[pos[k]] = val[k];
    }
    if (HAS_V && !V_LDS) {
        // 16/32-byte values: each one is its own sector, scatter straight from HBM to HBM
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const u32 li = wbase + k * COL_WAVE;
            if (li < valid) vals_out[s_goff[digit_of(key[k], shift)] + pos[k]] = vals_in[tile_base + li];
        }
    }
    __syncthreads();
    STAMP(3)      // scatter into LDS

    // Read back and stream out: consecutive lanes take consecutive tile-sorted slots, i.e. consecutive
    // addresses inside each digit run.  (Measured and not kept: a lane taking 4 consecutive slots and
    // storing them as one 4-byte-aligned dwordx4 -- 8 store instructions per thread instead of 32 -- is
    // 9 % faster with the output forced coalesced and 2-10 % SLOWER on the real runs, even when every
    // store hits L2: misaligned 16-byte stores are split in the address unit.)
    if (DIAG && (dbg & (1 << 23))) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int k = 0; k < IT; k++) {
        const u32 i = k * NT + tid;
        if (DIAG && (dbg & (1 << 29)) && k) __builtin_amdgcn_s_sleep(2);
        if (i < valid) {
            const K kk = s_keys[i];
            u32 g = s_goff[digit_of(kk, shift)] + i;
            if (DIAG) {
                if (dbg & 2) g = (u32)tile_base + i;           // timing ablation: coalesced output
                if (dbg & 32768) g &= (1u << 20) - 1;          // timing ablation: every store lands in a 4 MiB window (L2)
                if (dbg & 8192) {      // timing ablation: every (tile, digit) run in a 128-byte cell of its own, digit-major
                    const u32 d = digit_of(kk, shift);         // (same spatial pattern as the real runs, but whole aligned lines)
                    g = (u32)(((u64)d * nblocks + b) * 32 + min(i - s_dstart[d], 31u));
                }
            }
            if (DIAG && (dbg & 16384)) {                       // timing ablation: (key, value) pairs interleaved in ONE output
                if constexpr (V_LDS && sizeof(K) == 4 && VB == 4)      // array of 8 n bytes at keys_out: a digit run is one 256-byte piece
                    reinterpret_cast<uint2 *>(keys_out)[g] = make_uint2(kk, s_vals[i]);
            } else if (DIAG && (dbg & 384)) {                  // timing ablations: stores at system (128) / agent (256) scope
                if (dbg & 128) {
                    st_scope<__HIP_MEMORY_SCOPE_SYSTEM>(&keys_out[g], kk);
                    if (V_LDS) st_scope<__HIP_MEMORY_SCOPE_SYSTEM>(&vals_out[g], s_vals[i]);
                } else {
                    st_scope<__HIP_MEMORY_SCOPE_AGENT>(&keys_out[g], kk);
                    if (V_LDS) st_scope<__HIP_MEMORY_SCOPE_AGENT>(&vals_out[g], s_vals[i]);
                }
            } else {
                keys_out[g] = kk;
                if (V_LDS) vals_out[g] = s_vals[i];
            }
        }
    }
    if (dbg & 32) { __builtin_amdgcn_s_waitcnt(0); }
    STAMP(4)      // read back + global stores (issue only)
}


// ---- the 16384-key tile (u32 keys WITHOUT values, from HUGE_N keys) ----
// Same ranking and the same tile-sorted image as k_scatter with twice the tile: digit runs of ~256 bytes (one whole
// line and two partial ones instead of two partial ones), 0.147 instead of 0.163 ms per 64 Mi-key pass.  Two 8-wave
// blocks per CU leave 128 VGPRs per thread, so the staged keys stay in LDS through the ranking (read back one item
// at a time) and come into registers only to cross the barrier after which the image -- the same 64 KB -- is
// overwritten in tile-sorted order.
// Not for (key, value) pairs: with the values taking their turn in the one image after the keys (a second 64 KB
// would leave one block per CU; ranks packed two to a register, the slots' digits four to a register, the values
// waiting in registers from the common load) the pass took 0.279 ms against 0.235 for the 8192-pair tile --
// bit-exact on the GPU suite, and dropped.
// RAGGED = the instance for the input's last, partial tile (one block, launched on its own: tile `first_tile`): its
// guarded element-wise loads would otherwise share the register allocation of the full-tile instance, whose asm
// loads must never be spilled (hipcc does not know they are in flight).
template <bool RAGGED>
__global__ __launch_bounds__(NT_HUGE, 4) void k_scatter_huge(const u32 *__restrict__ keys_in, u32 *__restrict__ keys_out,
                                                             uint64_t n, u32 nblocks, u32 first_tile, int shift,
                                                             const u32 *__restrict__ offsets) {
    constexpr int NT = NT_HUGE, IT = IT_HUGE, TILE = NT * IT, NW = NT / COL_WAVE, SLICE = COL_WAVE * IT;
    __shared__ __attribute__((aligned(16))) u32 s_img[TILE];
    __shared__ u32 s_cnt[NW][RDIG];
    __shared__ u32 s_goff[RDIG];
    __shared__ u32 s_ws[NW];
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const u32 tid = threadIdx.x, lane = tid & (COL_WAVE - 1), w = tid / COL_WAVE;
    u32 b = blockIdx.x;                                   // XCD-contiguous tile ranges, as in k_scatter
    if (!RAGGED) {
        const u32 q = gridDim.x / 8, r = gridDim.x % 8, xcd = b % 8;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    b += first_tile;
    const uint64_t tile_base = (uint64_t)b * TILE;
    const u32 valid = RAGGED ? (u32)(n - tile_base) : (u32)TILE;
    const u32 my_offset = tid < RDIG ? offsets[(uint64_t)tid * nblocks + b] : 0u;
    for (u32 i = lane; i < RDIG; i += COL_WAVE) s_cnt[w][i] = 0;      // (this wave's own counters)
    u32 *stage = s_img + w * SLICE;
    v4u kq[IT / 4];
    const u32 e0 = w * SLICE + lane * 4;                  // tile-relative index of this lane's first key of vector 0
    if (!RAGGED) {
#pragma unroll
        for (int j = 0; j < IT / 4; j++)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kq[j]) : "v"(keys_in + tile_base + e0 + j * (COL_WAVE * 4)) : "memory");
#pragma unroll
        for (int j = 0; j < IT / 4; j++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(kq[j])::"memory");
    } else {
#pragma unroll
        for (int j = 0; j < IT / 4; j++) {
            const u32 e = e0 + j * (COL_WAVE * 4);
            for (int c = 0; c < 4; c++) kq[j][c] = e + c < valid ? keys_in[tile_base + e + c] : 0xFFFFFFFFu;   // padding sorts last, never stored
        }
    }
#pragma unroll
    for (int j = 0; j < IT / 4; j++) *reinterpret_cast<v4u *>(stage + j * (COL_WAVE * 4) + lane * 4) = kq[j];
    __builtin_amdgcn_wave_barrier();          // (LDS operations of one wave execute in order; the slice and the counters are its own)

    // rank inside (wave, digit): k_scatter's scheme, the key read back from the slice item by item
    u32 pos[IT];
#pragma unroll
    for (int k = 0; k < IT; k++) {
        const u32 d = digit_of(stage[k * COL_WAVE + lane], shift);
        const u64 peers = match8(d);
        const u32 below = mbcnt(peers);
        const u32 prev = s_cnt[w][d];
        if (below == 0) s_cnt[w][d] = prev + (u32)__popcll(peers);
        pos[k] = prev + below;
    }
    __syncthreads();
    {
        u32 c[NW], tot = 0;
        if (tid < RDIG) {
#pragma unroll
            for (int i = 0; i < NW; i++) { c[i] = s_cnt[i][tid]; tot += c[i]; }
        }
        u32 total;
        const u32 dstart = block_excl_scan<NT>(tot, s_ws, &total);      // threads >= 256 contribute 0
        if (tid < RDIG) {
            u32 run = dstart;
#pragma unroll
            for (int i = 0; i < NW; i++) { s_cnt[i][tid] = run; run += c[i]; }
            s_goff[tid] = my_offset - dstart;      // global position of tile-sorted slot i with digit d is s_goff[d] + i
        }
    }
    __syncthreads();
    {
        u32 key[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) {
            key[k] = stage[k * COL_WAVE + lane];
            pos[k] += s_cnt[w][digit_of(key[k], shift)];
        }
        __syncthreads();                      // every wave holds its keys: the slices may be overwritten
#pragma unroll
        for (int k = 0; k < IT; k++) s_img[pos[k]] = key[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IT; k++) {
        const u32 i = k * NT + tid;
        if (i < valid) {
            const u32 kk = s_img[i];
            keys_out[s_goff[digit_of(kk, shift)] + i] = kk;
        }
    }
}

// ---- MSD finish: one block sorts one top-digit bucket on its remaining low bits ----
// col_radix_sort_msd (small inputs, launch-bound): ONE global pass on the top 8 significant bits
// (k_scatter, shift BS_SHIFT) leaves 256 buckets; a bucket of up to BS_CAP pairs is then sorted on
// the low BS_SHIFT bits entirely in LDS (three stable 8-bit LSD passes, ranking as in k_scatter), which
// replaces three global passes of three launches each.  A larger bucket (clustered codes) takes the
// same three passes through global memory, chunk by chunk, by its one block: correct but slow, and
// reported through *oversize so that the caller goes back to the LSD sort.
constexpr int BS_NT = 1024, BS_NW = BS_NT / COL_WAVE;
constexpr int BS_SHIFT = 22;                   // bucket digit = bits 22..29 of a 30-bit Morton code (pads: 255)
// BS_IT items per thread: 8 -> buckets of up to 8192 pairs (73 KB of LDS, inputs below ~2 M codes),
//                         16 -> 16384 pairs (145 KB: one workgroup per CU, inputs up to ~4 M codes)

template <int BS_IT> struct BucketLds {
    u32 keys[BS_NT * BS_IT];
    u32 vals[BS_NT * BS_IT];
    u32 cnt[BS_NW][RDIG];
    u32 base[RDIG];
    u32 ws[BS_NW];
};

// One stable pass over the (up to BS_CAP) pairs held lane-striped in registers: position of item k of
// this thread = w * row_len + k * 64 + lane (k * 64 < row_len); rows at or beyond `m` hold pads (key
// 0xFFFFFFFF) and are skipped.  On return pos[k] = rank of the item inside its (wave, digit) group and lds.cnt[i][d] = number
// of ranked items (pads of a partly valid row included) of wave i with digit d.
template <int BS_IT>
__device__ __forceinline__ void bucket_rank(BucketLds<BS_IT> &lds, const u32 (&key)[BS_IT], u32 (&pos)[BS_IT], u32 m, int shift,
                                            u32 w, u32 tid, u32 row_len) {
    for (u32 i = tid; i < BS_NW * RDIG; i += BS_NT) (&lds.cnt[0][0])[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BS_IT; k++) {
        pos[k] = 0;
        if ((u32)k * COL_WAVE >= row_len || w * row_len + k * COL_WAVE >= m) continue;    // wave-uniform: no such row / pads only
        const u32 d = digit_of(key[k], shift);
        const u64 peers = match8(d);
        const u32 below = mbcnt(peers);
        const u32 prev = lds.cnt[w][d];
        if (below == 0) lds.cnt[w][d] = prev + (u32)__popcll(peers);
        pos[k] = prev + below;
    }
    __syncthreads();
}

template <int BS_IT>
__global__ __launch_bounds__(BS_NT) void k_bucket_sort(u32 *__restrict__ k_a, u32 *__restrict__ v_a, u32 *__restrict__ k_b,
                                                        u32 *__restrict__ v_b, u32 n, const u32 *__restrict__ offsets,
                                                        u32 nblocks, u32 *oversize, const u32 *__restrict__ n_real_dev) {
    constexpr u32 BS_CAP = BS_NT * BS_IT;
    constexpr int BS_ROW = COL_WAVE * BS_IT;       // positions per wave
    __shared__ BucketLds<BS_IT> lds;
    const u32 tid = threadIdx.x, lane = tid & (COL_WAVE - 1);
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(tid / COL_WAVE));
    const u32 d = blockIdx.x;
    const u32 start = offsets[(uint64_t)d * nblocks], end = d + 1 < RDIG ? offsets[(uint64_t)(d + 1) * nblocks] : n;
    // DEVICE-SIDE COUNT (col_common.h): of the n pairs only the first *n_real_dev are real codes, the rest are pads
    // (0xFFFFFFFF) -- possibly far more than a bucket holds when n is a capacity.  They sit at the END of bucket 255 in the order
    // of their ids (the global pass is stable and they came last), i.e. sorted position j >= n_real holds the pad with id j:
    // bucket 255 sorts its real codes only, and every workgroup writes its 256th of the pads (one workgroup copying 3 * 10^5
    // of them through cost the one-rank step 0.14 ms).
    u32 pads = 0;
    if (n_real_dev) {
        const u32 n_real = min(n, *n_real_dev), all = n - n_real, per = (all + RDIG - 1) / RDIG;
        for (u32 i = threadIdx.x; i < per; i += BS_NT) {
            const u32 j = n_real + d * per + i;
            if (j < n) { k_b[j] = 0xFFFFFFFFu; v_b[j] = j; }
        }
        if (d == RDIG - 1) pads = min(end - start, all);
    }
    const u32 S = end - start - pads;
    if (S == 0) return;
    u32 key[BS_IT], val[BS_IT], pos[BS_IT];

    if (S <= BS_CAP) {
        // every wave takes an equal contiguous slice of the bucket (a multiple of 64 positions)
        const u32 L = ((S + BS_NW - 1) / BS_NW + COL_WAVE - 1) & ~(u32)(COL_WAVE - 1);
#pragma unroll
        for (int k = 0; k < BS_IT; k++) {
            const u32 p = w * L + k * COL_WAVE + lane;
            const bool ok = (u32)k * COL_WAVE < L && p < S;
            key[k] = ok ? k_a[start + p] : 0xFFFFFFFFu;
            val[k] = ok ? v_a[start + p] : 0u;
        }
        for (int shift = 0; shift < BS_SHIFT; shift += 8) {
            bucket_rank<BS_IT>(lds, key, pos, S, shift, w, tid, L);
            {   // exclusive over (digit, wave): digit `tid`
                u32 c[BS_NW], tot = 0;
                if (tid < RDIG) {
#pragma unroll
                    for (int i = 0; i < BS_NW; i++) { c[i] = lds.cnt[i][tid]; tot += c[i]; }
                }
                u32 total;
                const u32 dstart = block_excl_scan<BS_NT>(tot, lds.ws, &total);
                if (tid < RDIG) {
                    u32 run = dstart;
#pragma unroll
                    for (int i = 0; i < BS_NW; i++) { lds.cnt[i][tid] = run; run += c[i]; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < BS_IT; k++) {
                if ((u32)k * COL_WAVE >= L || w * L + k * COL_WAVE >= S) continue;
                const u32 q = pos[k] + lds.cnt[w][digit_of(key[k], shift)];
                lds.keys[q] = key[k];
                lds.vals[q] = val[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < BS_IT; k++) {
                const u32 p = w * L + k * COL_WAVE + lane;
                if ((u32)k * COL_WAVE >= L || w * L + k * COL_WAVE >= S) continue;
                key[k] = p < S ? lds.keys[p] : 0xFFFFFFFFu;       // pads of a partly valid row sorted last: drop them again
                val[k] = p < S ? lds.vals[p] : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < BS_IT; k++) {
            const u32 p = w * L + k * COL_WAVE + lane;
            if ((u32)k * COL_WAVE < L && p < S) { k_b[start + p] = key[k]; v_b[start + p] = val[k]; }
        }
        return;
    }

    // oversize bucket: the same three passes through global memory, a -> b -> a -> b
    if (tid == 0 && oversize) *oversize = S;
    int pass = 0;
    for (int shift = 0; shift < BS_SHIFT; shift += 8, pass++) {
        const u32 *sk = ((pass & 1) ? k_b : k_a) + start, *sv = ((pass & 1) ? v_b : v_a) + start;
        u32 *dk = ((pass & 1) ? k_a : k_b) + start, *dv = ((pass & 1) ? v_a : v_b) + start;
        if (tid < RDIG) lds.base[tid] = 0;
        __syncthreads();
        for (u32 i = tid; i < S; i += BS_NT) atomicAdd(&lds.base[digit_of(sk[i], shift)], 1u);
        __syncthreads();
        {
            u32 total;
            const u32 mine = tid < RDIG ? lds.base[tid] : 0u;
            const u32 ex = block_excl_scan<BS_NT>(mine, lds.ws, &total);
            if (tid < RDIG) lds.base[tid] = ex;                  // where digit `tid` starts in the bucket
        }
        __syncthreads();
        for (u32 c0 = 0; c0 < S; c0 += BS_CAP) {
            const u32 m = min(BS_CAP, S - c0);
#pragma unroll
            for (int k = 0; k < BS_IT; k++) {
                const u32 p = w * BS_ROW + k * COL_WAVE + lane;
                key[k] = p < m ? sk[c0 + p] : 0xFFFFFFFFu;
                val[k] = p < m ? sv[c0 + p] : 0u;
            }
            bucket_rank<BS_IT>(lds, key, pos, m, shift, w, tid, (u32)BS_ROW);
            if (tid < RDIG) {
                // this chunk's (wave, digit) groups start where the digit's running offset stands; pads (digit
                // 255 in every pass, ranked after the real items of their row) do not advance it
                u32 run = lds.base[tid];
#pragma unroll
                for (int i = 0; i < BS_NW; i++) { const u32 c = lds.cnt[i][tid]; lds.cnt[i][tid] = run; run += c; }
                if (tid == RDIG - 1) {
                    const u32 ranked = ((m + COL_WAVE - 1) / COL_WAVE) * COL_WAVE;     // rows are ranked whole
                    run -= ranked - m;
                }
                lds.base[tid] = run;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < BS_IT; k++) {
                const u32 p = w * BS_ROW + k * COL_WAVE + lane;
                if (p < m) {
                    const u32 q = pos[k] + lds.cnt[w][digit_of(key[k], shift)];
                    dk[q] = key[k];
                    dv[q] = val[k];
                }
            }
            __syncthreads();
        }
        // the next pass reads what other threads of this block just wrote (and, for `a`, what this block read
        // before): write back, then drop this CU's cached lines
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

// ---- reference-structured kernels (kernel-level parity only; one wave per block) ----
// radix.cl:48-102 block_sort: stable sort of each block of `block` elements by the digit,
// in place, + digit-major histogram.  Dynamic LDS: keys, values, 2^bits counters.
template <typename K>
__global__ __launch_bounds__(COL_WAVE) void k_ref_block_sort(K *keys, unsigned char *vals, u32 block, u32 vb,
                                                             int bits, int pass, u32 nblocks, u32 *hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 nd = 1u << bits;
    u32 *cnt = reinterpret_cast<u32 *>(smem);
    K *sk = reinterpret_cast<K *>(smem + ((nd * 4 + 15) & ~15u));
    unsigned char *sv = reinterpret_cast<unsigned char *>(sk + block);
    const u32 lane = threadIdx.x, b = blockIdx.x;
    K *bk = keys + (uint64_t)b * block;
    unsigned char *bv = vals ? vals + (uint64_t)b * block * vb : nullptr;
    const int shift = pass * bits;
    const u32 mask = nd - 1;
    for (u32 i = lane; i < nd; i += COL_WAVE) cnt[i] = 0;
    __syncthreads();
    for (u32 i = lane; i < block; i += COL_WAVE) atomicAdd(&cnt[(u32)(bk[i] >> shift) & mask], 1u);
    __syncthreads();
    for (u32 i = lane; i < nd; i += COL_WAVE) hist[(uint64_t)i * nblocks + b] = cnt[i];
    __syncthreads();
    if (lane == 0) {
        u32 acc = 0;
        for (u32 d = 0; d < nd; d++) { const u32 c = cnt[d]; cnt[d] = acc; acc += c; }
    }
    __syncthreads();
    for (u32 c = 0; c < block; c += COL_WAVE) {
        const u32 i = c + lane;
        const bool act = i < block;
        const K kk = act ? bk[i] : (K)0;
        const u32 d = act ? ((u32)(kk >> shift) & mask) : 0xFFFFFFFFu;
        // peers with the same digit among active lanes (digits up to 16 bits: compare via shuffles)
        u64 peers = 0;
        for (int l = 0; l < COL_WAVE; l++) {
            const u32 od = __shfl(d, l, COL_WAVE);
            if (od == d) peers |= 1ull << l;
        }
        if (act) {
            const u32 below = mbcnt(peers);
            const u32 p = cnt[d] + below;
            sk[p] = kk;
            if (bv) for (u32 q = 0; q < vb; q++) sv[(uint64_t)p * vb + q] = bv[(uint64_t)i * vb + q];
        }
        __syncthreads();
        if (act && mbcnt(peers) == 0) cnt[d] += (u32)__popcll(peers);
        __syncthreads();
    }
    for (u32 i = lane; i < block; i += COL_WAVE) bk[i] = sk[i];
    if (bv) for (u32 i = lane; i < block * vb; i += COL_WAVE) bv[i] = sv[i];
}

// radix.cl:104-139 scatter
template <typename K>
__global__ __launch_bounds__(COL_WAVE) void k_ref_scatter(const K *keys, K *keys_out, const unsigned char *vals,
                                                          unsigned char *vals_out, u32 block, u32 vb, int bits,
                                                          int pass, u32 nblocks, const u32 *offsets, const u32 *hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 nd = 1u << bits;
    u32 *loff = reinterpret_cast<u32 *>(smem);
    u32 *lstart = loff + nd;
    const u32 lane = threadIdx.x, b = blockIdx.x;
    const int shift = pass * bits;
    for (u32 i = lane; i < nd; i += COL_WAVE) {
        loff[i] = offsets[(uint64_t)i * nblocks + b];
        lstart[i] = hist[(uint64_t)i * nblocks + b];
    }
    __syncthreads();
    if (lane == 0) {
        u32 acc = 0;
        for (u32 d = 0; d < nd; d++) { const u32 c = lstart[d]; lstart[d] = acc; acc += c; }
    }
    __syncthreads();
    for (u32 i = lane; i < block; i += COL_WAVE) {
        const uint64_t src = (uint64_t)b * block + i;
        const K kk = keys[src];
        const u32 d = (u32)(kk >> shift) & (nd - 1);
        const u32 dst = loff[d] + i - lstart[d];
        keys_out[dst] = kk;
        if (vals && vals_out) for (u32 q = 0; q < vb; q++) vals_out[(uint64_t)dst * vb + q] = vals[src * vb + q];
    }
}

// Process-wide diagnostics switches (col_debug_radix / col_debug_radix_tile): they change the kernels of
// EVERY caller in the process and are not synchronised -- set them from one thread, with no sort in flight.
int g_radix_dbg = 0;
int g_radix_tile_override = 0;      // 0 = automatic, else 1024 / 4096 / 8192 / 16384
int g_radix_wide_block = 0;         // the 8192-pair tile with 1024 threads x 8 items (experiment, col_debug_radix_tile(8193))
uint64_t g_big_n = BIG_N_DEFAULT;    // A/B material (col_debug_radix_tile(16 << 20) = the round-2 threshold; buffers sized before a switch do not follow it)
#define BIG_N g_big_n
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline bool huge_ok(int key_bytes, int val_bytes) { return key_bytes == 4 && val_bytes == 0; }
inline u32 tile_auto(uint64_t n, int key_bytes, int val_bytes) {
    u32 t = n < SMALL_N ? (u32)(NT_SMALL * IT_SMALL) : n < BIG_N ? (u32)(NT_MID * IT_BIG) : (u32)(NT_BIG * IT_BIG);
    if (n >= HUGE_N && huge_ok(key_bytes, val_bytes)) t = (u32)(NT_HUGE * IT_HUGE);
    if (key_bytes == 8 && t > (u32)(NT_MID * IT_BIG)) t = (u32)(NT_MID * IT_BIG);
    return t;
}
inline u32 tile_for(uint64_t n, int key_bytes, int val_bytes) {
    if (!g_radix_tile_override) return tile_auto(n, key_bytes, val_bytes);
    u32 t = (u32)g_radix_tile_override;
    if (t > (u32)(NT_BIG * IT_BIG) && !huge_ok(key_bytes, val_bytes)) t = (u32)(NT_BIG * IT_BIG);
    if (key_bytes == 8 && t > (u32)(NT_MID * IT_BIG)) t = (u32)(NT_MID * IT_BIG);
    return t;
}
inline u32 tiles_of(uint64_t n, int key_bytes, int val_bytes) { return (u32)col_ceil_div(n, tile_for(n, key_bytes, val_bytes)); }
// The most tiles any n' <= n can have: the tile GROWS with n, so a smaller input may have more tiles (and a
// larger histogram) than n itself.  Scratch is sized with this bound, so one scratch buffer sized for n
// serves every n' <= n (col_collide on a varying number of owned spheres: collision_amd/multi.py).
inline size_t max_tiles_upto(uint64_t n, int key_bytes) {
    if (n == 0) return 1;
    size_t m = col_ceil_div(n < SMALL_N ? n : SMALL_N - 1, NT_SMALL * IT_SMALL);
    if (n >= SMALL_N) {
        const uint64_t top = (key_bytes == 8 || n < BIG_N) ? n : BIG_N - 1;
        m = std::max(m, (size_t)col_ceil_div(top, NT_MID * IT_BIG));
    }
    if (n >= BIG_N && key_bytes != 8) m = std::max(m, (size_t)col_ceil_div(n, NT_BIG * IT_BIG));
    if (g_radix_tile_override) m = std::max(m, (size_t)col_ceil_div(n, NT_SMALL * IT_SMALL));
    return m;
}

inline u32 hist_group(u32 nblocks) {
    // keep >= ~1024 histogram blocks when the input allows, else fewer tiles per block
    u32 g = HG;
    while (g > 1 && nblocks / g < 1024) g >>= 1;
    return g;
}

template <typename K>
int launch_hist(hipStream_t s, const void *keys, uint64_t n, int val_bytes, int pass, u32 *hist) {
    const u32 tile = tile_for(n, (int)sizeof(K), val_bytes);
    const u32 nb = tiles_of(n, (int)sizeof(K), val_bytes), g = hist_group(nb);
    dim3 grid((unsigned)col_ceil_div(nb, g)), block(RT);
    const K *k = (const K *)keys;
    if (tile == (u32)(NT_SMALL * IT_SMALL)) k_hist<K, NT_SMALL * IT_SMALL><<<grid, block, 0, s>>>(k, n, nb, g, pass * 8, hist);
    else if (tile == (u32)(NT_MID * IT_BIG)) k_hist<K, NT_MID * IT_BIG><<<grid, block, 0, s>>>(k, n, nb, g, pass * 8, hist);
    else if (tile == (u32)(NT_BIG * IT_BIG)) k_hist<K, NT_BIG * IT_BIG><<<grid, block, 0, s>>>(k, n, nb, g, pass * 8, hist);
    else k_hist<K, NT_HUGE * IT_HUGE><<<grid, block, 0, s>>>(k, n, nb, g, pass * 8, hist);
    COL_LAUNCH_OK();
    return COL_OK;
}

template <typename K, int VB, int IT, int NT>
int launch_scatter_one(hipStream_t s, const K *ki, K *ko, const void *vals, void *vals_out, uint64_t n, u32 nb, int shift,
                       const u32 *offsets) {
    dim3 grid(nb), block(NT);
    // the diagnostics instances exist for the profiled combinations only (u32 keys, no / 4-byte values)
    if constexpr (sizeof(K) == 4 && (VB == 0 || VB == 4)) {
        if (g_radix_dbg) {
            // timing ablation: unused dynamic LDS lowers the number of resident blocks per CU (modes 512 / 1024)
            const size_t dyn = (g_radix_dbg & 1024) ? 18432 : (g_radix_dbg & 512) ? 4608 : 0;
            k_scatter<K, VB, IT, NT, true><<<grid, block, dyn, s>>>(ki, ko, vals, vals_out, n, nb, shift, offsets, DiagOn{g_radix_dbg});
            COL_LAUNCH_OK();
            return COL_OK;
        }
    }
    k_scatter<K, VB, IT, NT, false><<<grid, block, 0, s>>>(ki, ko, vals, vals_out, n, nb, shift, offsets, DiagOff{});
    COL_LAUNCH_OK();
    return COL_OK;
}

template <typename K, int IT, int NT>
int launch_scatter_it(hipStream_t s, const void *keys, void *keys_out, const void *vals, void *vals_out,
                      uint64_t n, int vb, int shift, const u32 *offsets) {
    const u32 nb = (u32)col_ceil_div(n, NT * IT);
    const K *ki = (const K *)keys;
    K *ko = (K *)keys_out;
    if (!vals || !vals_out) vb = 0;
    switch (vb) {
    case 0: return launch_scatter_one<K, 0, IT, NT>(s, ki, ko, nullptr, nullptr, n, nb, shift, offsets);
    case 4: return launch_scatter_one<K, 4, IT, NT>(s, ki, ko, vals, vals_out, n, nb, shift, offsets);
    case 8: return launch_scatter_one<K, 8, IT, NT>(s, ki, ko, vals, vals_out, n, nb, shift, offsets);
    case 16: return launch_scatter_one<K, 16, IT, NT>(s, ki, ko, vals, vals_out, n, nb, shift, offsets);
    case 32: return launch_scatter_one<K, 32, IT, NT>(s, ki, ko, vals, vals_out, n, nb, shift, offsets);
    default: return COL_EINVAL;
    }
}

template <typename K>
int launch_scatter(hipStream_t s, const void *keys, void *keys_out, const void *vals, void *vals_out,
                   uint64_t n, int vb, int shift, const u32 *offsets) {
    if (!vals || !vals_out) vb = 0;
    const u32 tile = tile_for(n, (int)sizeof(K), vb);
    if (tile == (u32)(NT_SMALL * IT_SMALL))
        return launch_scatter_it<K, IT_SMALL, NT_SMALL>(s, keys, keys_out, vals, vals_out, n, vb, shift, offsets);
    if (tile == (u32)(NT_MID * IT_BIG))
        return launch_scatter_it<K, IT_BIG, NT_MID>(s, keys, keys_out, vals, vals_out, n, vb, shift, offsets);
    if constexpr (sizeof(K) == 4) {
        if (tile == (u32)(NT_HUGE * IT_HUGE)) {          // (u32 keys without values only: tile_for)
            constexpr uint64_t T = NT_HUGE * IT_HUGE;
            const u32 nb = (u32)col_ceil_div(n, T), nfull = (u32)(n / T);
            if (nfull) {
                k_scatter_huge<false><<<dim3(nfull), dim3(NT_HUGE), 0, s>>>((const u32 *)keys, (u32 *)keys_out, n, nb, 0, shift, offsets);
                COL_LAUNCH_OK();
            }
            if (nb > nfull) {                              // the partial last tile
                k_scatter_huge<true><<<dim3(1), dim3(NT_HUGE), 0, s>>>((const u32 *)keys, (u32 *)keys_out, n, nb, nfull, shift, offsets);
                COL_LAUNCH_OK();
            }
            return COL_OK;
        }
        // the 8192-pair tile as 1024 threads x 8 items (32 waves per CU instead of 16, same LDS): col_debug_radix_tile(8193)
        if (g_radix_wide_block && (vb == 0 || vb == 4 || vb == 8))
            return launch_scatter_it<K, IT_BIG / 2, NT_BIG * 2>(s, keys, keys_out, vals, vals_out, n, vb, shift, offsets);
        return launch_scatter_it<K, IT_BIG, NT_BIG>(s, keys, keys_out, vals, vals_out, n, vb, shift, offsets);
    }
    return COL_EINVAL;
}

// Value widths the scatter kernel moves itself; 1 / 2 / 64 / 128-byte values (the reference specialises for
// any NumPy dtype, radix.py:16-25: uint8 ... (float64, 16)) are sorted as (key, index) pairs and gathered
// once at the end (wide_value below).
inline bool native_value(int val_bytes) { return val_bytes == 0 || val_bytes == 4 || val_bytes == 8 || val_bytes == 16 || val_bytes == 32; }
inline bool wide_value(int val_bytes) { return val_bytes == 1 || val_bytes == 2 || val_bytes == 64 || val_bytes == 128; }
inline bool bad_sizes(uint64_t n, int key_bytes, int val_bytes) {
    if (key_bytes != 4 && key_bytes != 8) return true;
    if (!native_value(val_bytes)) return true;
    return n >= 0xFFFFFFFFull;   // offsets are uint32 (as in the reference)
}

__global__ __launch_bounds__(256) void k_iota(u32 *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (u32)i;
}
// uint8 / uint16 keys (radix.py:16-20 accepts every unsigned dtype): widened to u32 for the sort -- one 8-bit pass
// per key byte -- and narrowed again on the way out
template <typename N>
__global__ __launch_bounds__(256) void k_widen(const N *__restrict__ in, u32 *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (u32)in[i];
}
template <typename N>
__global__ __launch_bounds__(256) void k_narrow(const u32 *__restrict__ in, N *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (N)in[i];
}
inline bool narrow_key(int key_bytes) { return key_bytes == 1 || key_bytes == 2; }
// out[i] = in[idx[i]] for elements of CH chunks of type X (CH * sizeof(X) bytes): one chunk per thread
template <typename X, int CH>
__global__ __launch_bounds__(256) void k_gather_chunks(const X *__restrict__ in, const u32 *__restrict__ idx, X *__restrict__ out, uint64_t n) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t i = t / CH;
    if (i >= n) return;
    const u32 c = (u32)(t % CH);
    out[i * CH + c] = in[(uint64_t)idx[i] * CH + c];
}
int gather_values(hipStream_t s, const void *in, const u32 *idx, void *out, uint64_t n, int val_bytes) {
    const int ch = val_bytes >= 16 ? val_bytes / 16 : 1;
    dim3 grid((unsigned)col_ceil_div(n * ch, 256)), block(256);
    switch (val_bytes) {
    case 1: k_gather_chunks<uint8_t, 1><<<grid, block, 0, s>>>((const uint8_t *)in, idx, (uint8_t *)out, n); break;
    case 2: k_gather_chunks<uint16_t, 1><<<grid, block, 0, s>>>((const uint16_t *)in, idx, (uint16_t *)out, n); break;
    case 64: k_gather_chunks<uint4, 4><<<grid, block, 0, s>>>((const uint4 *)in, idx, (uint4 *)out, n); break;
    case 128: k_gather_chunks<uint4, 8><<<grid, block, 0, s>>>((const uint4 *)in, idx, (uint4 *)out, n); break;
    default: return COL_EINVAL;
    }
    COL_LAUNCH_OK();
    return COL_OK;
}

// ---- copy ceilings (diagnostics: bench.py's roofline.copy_ceiling; tools/micro/copy_ceiling.hip has the full set) ----
// What a pass that reads n bytes and writes n bytes can reach on this box, measured beside the scatter pass:
//   shape 0: the plain float4 copy, ONE 16-byte vector per thread, one-shot grid -- the fastest form found (more vectors per
//            thread, grid-strided or persistent grids are all slower: EXPERIMENTS.md R4.1);
//   shape 1: the scatter pass's own shape -- one 512-thread workgroup per 64 KB tile, all of its loads issued up front, the
//            tile staged through 64 KB of LDS, coalesced dword stores, XCD-contiguous tile order -- with no ranking at all.
__global__ __launch_bounds__(256) void k_dbg_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t nvec) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) out[i] = in[i];
}
__global__ __launch_bounds__(NT_BIG) void k_dbg_tile_copy(const u32 *__restrict__ in, u32 *__restrict__ out, uint64_t half_words) {
    constexpr int TILE = NT_BIG * IT_BIG;               // 8192 "pairs": 32 KB from each half of the buffer
    __shared__ __attribute__((aligned(16))) u32 s_k[TILE], s_v[TILE];
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 q = gridDim.x / 8, r = gridDim.x % 8, xcd = blockIdx.x % 8;
    const u32 b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + blockIdx.x / 8;
    const uint64_t base = (uint64_t)b * TILE;
    const u32 *ka = in + base + w * 1024 + lane * 4, *va = in + half_words + base + w * 1024 + lane * 4;
    v4u kq[4], vq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kq[j]) : "v"(ka + j * 256) : "memory");
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vq[j]) : "v"(va + j * 256) : "memory");
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(kq[j])::"memory");
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(vq[j])::"memory");
#pragma unroll
    for (int j = 0; j < 4; j++) {
        *reinterpret_cast<v4u *>(s_k + w * 1024 + j * 256 + lane * 4) = kq[j];
        *reinterpret_cast<v4u *>(s_v + w * 1024 + j * 256 + lane * 4) = vq[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IT_BIG; k++) {
        const u32 i = k * NT_BIG + tid;
        out[base + i] = s_k[i];
        out[half_words + base + i] = s_v[i];
    }
}

}  // namespace

extern "C" {

int col_debug_copy(void *stream, const void *in, void *out, uint64_t bytes, int shape) {
    if (!in || !out || bytes == 0 || bytes % (2 * NT_BIG * IT_BIG * 4) != 0) return COL_EINVAL;      // whole tiles in both halves
    if (shape == 0) k_dbg_copy<<<dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, col_stream(stream)>>>((const uint4 *)in, (uint4 *)out, bytes / 16);
    else if (shape == 1) k_dbg_tile_copy<<<dim3((unsigned)(bytes / (2 * NT_BIG * IT_BIG * 4))), dim3(NT_BIG), 0, col_stream(stream)>>>((const u32 *)in, (u32 *)out, bytes / 8);
    else return COL_EINVAL;
    COL_LAUNCH_OK();
    return COL_OK;
}

void col_debug_radix(int mode) { g_radix_dbg = mode; }

int col_debug_radix_tile(int tile) {
    if (tile == 8193 || tile == 8194) { g_radix_wide_block = tile == 8193; return COL_OK; }      // (8194: back to 512 x 16)
    if (tile == (16 << 20) || tile == (8 << 20)) { g_big_n = (uint64_t)tile; return COL_OK; }                    // the 4096 / 8192 threshold (A/B)
    if (tile != 0 && tile != NT_SMALL * IT_SMALL && tile != NT_MID * IT_BIG && tile != NT_BIG * IT_BIG && tile != NT_HUGE * IT_HUGE) return COL_EINVAL;
    g_radix_tile_override = tile;
    return COL_OK;
}

int col_debug_radix_stamps(uint64_t *out, int reset) {
    unsigned long long h[8] = {0};
    if (out) { COL_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h))); for (int i = 0; i < 8; i++) out[i] = h[i]; }
    if (reset) { unsigned long long z[8] = {0}; COL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z))); }
    return COL_OK;
}

uint32_t col_radix_tile(uint64_t n, int key_bytes, int val_bytes) {
    if (narrow_key(key_bytes)) key_bytes = 4;
    return tile_for(n, key_bytes, wide_value(val_bytes) ? 4 : val_bytes);       // (wide values are sorted as (key, index) pairs)
}

// Monotone in n: sized for the largest histogram any n' <= n can need (see max_tiles_upto), so a scratch
// buffer sized for n serves every smaller sort / col_collide call as well.
size_t col_radix_scratch_bytes(uint64_t n, int key_bytes, int val_bytes) {
    if (narrow_key(key_bytes))       // the u32 sort + the widened keys and their sorted image
        return col_radix_scratch_bytes(n, 4, val_bytes) + 2 * align256((size_t)n * 4);
    if (wide_value(val_bytes))       // (key, index) sort + the index ramp and its sorted image
        return col_radix_scratch_bytes(n, key_bytes, 4) + 2 * align256((size_t)n * 4);
    const size_t nb = max_tiles_upto(n, key_bytes);
    const size_t hist = align256((size_t)RDIG * nb * sizeof(u32));
    return hist + align256(col_scan_scratch_bytes((uint64_t)RDIG * nb)) + align256((size_t)n * key_bytes) +
           align256((size_t)n * val_bytes) + 256;
}

int col_radix_tile_override_active(void) { return g_radix_tile_override != 0; }

int col_radix_histogram(void *stream, const void *keys, uint64_t n, int key_bytes, int val_bytes, int pass,
                        uint32_t *hist) {
    if (bad_sizes(n, key_bytes, val_bytes) || pass < 0 || pass >= key_bytes) return COL_EINVAL;
    if (n == 0) return COL_OK;
    return key_bytes == 4 ? launch_hist<uint32_t>(col_stream(stream), keys, n, val_bytes, pass, hist)
                          : launch_hist<uint64_t>(col_stream(stream), keys, n, val_bytes, pass, hist);
}

int col_radix_scatter(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                      uint64_t n, int key_bytes, int val_bytes, int pass, const uint32_t *offsets) {
    if (bad_sizes(n, key_bytes, val_bytes) || pass < 0 || pass >= key_bytes) return COL_EINVAL;
    if (n == 0) return COL_OK;
    return key_bytes == 4
               ? launch_scatter<uint32_t>(col_stream(stream), keys, keys_out, vals, vals_out, n, val_bytes, pass * 8, offsets)
               : launch_scatter<uint64_t>(col_stream(stream), keys, keys_out, vals, vals_out, n, val_bytes, pass * 8, offsets);
}

// After an LSD sort of 30-bit codes: the size of the largest MSD bucket (codes sharing bits 22..29) read off the SORTED
// codes -- thread d finds where bucket d starts by a binary search -- published as 0x80000000 | max in a host-visible
// word.  A bucket above the MSD finish's capacity means the MSD plan WOULD take its slow path: the caller
// (collision_amd/collision.py) then does not try it -- clustered scenes never pay for a second probe.  One small launch as col_radix_bucket_report;
// inside col_collide the same code (col_bucket_report_block, col_common.h) is the last workgroup of k_chunk's grid (lbvh.hip).
__global__ __launch_bounds__(RDIG) void k_bucket_report(const u32 *__restrict__ sorted, u32 n, u32 *word) {
    static_assert(BS_SHIFT_REPORT == 22 && RDIG == 256, "col_bucket_report_block (col_common.h) is written for this digit");
    __shared__ u32 s_start[RDIG + 1];
    col_bucket_report_block(sorted, n, word, s_start);
}

// MSD sort of (u32 key, u32 value) pairs whose keys are 30-bit codes (or 0xFFFFFFFF pads), for inputs up to
// COL_MSD_MAX_N pairs.  The histogram of the bucket digit (bits 22..29, digit-major, one row entry per tile
// of col_radix_tile(n) pairs) must already be at the start of `scratch` (col_morton_tile writes it).
// Same result as col_radix_sort.  *oversize (device-visible, may be NULL) receives the size of a bucket
// that did not fit LDS, if any.
int col_radix_sort_msd(void *stream, const uint32_t *keys, uint32_t *keys_out, const uint32_t *vals, uint32_t *vals_out,
                       uint64_t n, void *scratch, uint32_t *oversize) {
    return col_radix_sort_msd_dev(stream, keys, keys_out, vals, vals_out, n, scratch, oversize, nullptr);
}

// ... with the number of real codes on the device (pads beyond it: see k_bucket_sort); internal (col_common.h)
int col_radix_sort_msd_dev(void *stream, const uint32_t *keys, uint32_t *keys_out, const uint32_t *vals, uint32_t *vals_out,
                           uint64_t n, void *scratch, uint32_t *oversize, const uint32_t *n_real_dev) {
    if (n == 0) return COL_OK;
    if (!scratch || !vals || !vals_out) return COL_EINVAL;
    if (n > COL_MSD_MAX_N) return COL_EINVAL;
    const u32 tile = tile_auto(n, 4, 4);                  // (a forced tile class does not apply here)
    hipStream_t s = col_stream(stream);
    const size_t nb = col_ceil_div(n, tile);
    char *p = (char *)scratch;
    u32 *hist = (u32 *)p;              p += align256((size_t)RDIG * nb * sizeof(u32));
    void *scan_scratch = p;            p += align256(col_scan_scratch_bytes((uint64_t)RDIG * nb));
    u32 *tmp_keys = (u32 *)p;          p += align256((size_t)n * 4);
    u32 *tmp_vals = (u32 *)p;
    int rc = col_scan_u32(stream, hist, (uint64_t)RDIG * nb, scan_scratch);
    if (rc) return rc;
    if (tile == (u32)(NT_SMALL * IT_SMALL)) rc = launch_scatter_it<u32, IT_SMALL, NT_SMALL>(s, keys, tmp_keys, vals, tmp_vals, n, 4, BS_SHIFT, hist);
    else rc = launch_scatter_it<u32, IT_BIG, NT_MID>(s, keys, tmp_keys, vals, tmp_vals, n, 4, BS_SHIFT, hist);
    if (rc) return rc;
    // a uniform scene puts n / 256 codes into a bucket: the 8192-pair finish up to ~1.9 M, the 16384-pair one above
    if (n <= COL_MSD_SMALL_N)
        k_bucket_sort<8><<<dim3(RDIG), dim3(BS_NT), 0, s>>>(tmp_keys, tmp_vals, keys_out, vals_out, (u32)n, hist, (u32)nb, oversize, n_real_dev);
    else
        k_bucket_sort<16><<<dim3(RDIG), dim3(BS_NT), 0, s>>>(tmp_keys, tmp_vals, keys_out, vals_out, (u32)n, hist, (u32)nb, oversize, n_real_dev);
    COL_LAUNCH_OK();
    return COL_OK;
}

static int sort_passes(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                       uint64_t n, int key_bytes, int val_bytes, void *scratch, int copy_back, int have_hist0, int passes);
// (u32 key, u32 value) pairs whose keys are below 2^(8 * passes): only that many digit passes (internal)
int col_radix_sort_low_passes(void *stream, const uint32_t *keys, uint32_t *keys_out, const uint32_t *vals,
                              uint32_t *vals_out, uint64_t n, void *scratch, int passes) {
    if (passes < 1 || passes > 4) return COL_EINVAL;
    return sort_passes(stream, keys, keys_out, vals, vals_out, n, 4, 4, scratch, 0, 0, passes);
}

int col_radix_bucket_report(void *stream, const uint32_t *sorted_codes, uint64_t n, uint32_t *word) {
    if (!sorted_codes || !word || n == 0 || n >= 0xFFFFFFFFull) return COL_EINVAL;
    k_bucket_report<<<dim3(1), dim3(RDIG), 0, col_stream(stream)>>>(sorted_codes, (u32)n, word);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_radix_sort(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                   uint64_t n, int key_bytes, int val_bytes, void *scratch, int copy_back) {
    return col_radix_sort_ex(stream, keys, keys_out, vals, vals_out, n, key_bytes, val_bytes, scratch, copy_back, 0);
}

static int sort_passes(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                       uint64_t n, int key_bytes, int val_bytes, void *scratch, int copy_back, int have_hist0, int passes);

int col_radix_sort_ex(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                      uint64_t n, int key_bytes, int val_bytes, void *scratch, int copy_back, int have_hist0) {
    if (!vals || !vals_out) val_bytes = 0;
    if (narrow_key(key_bytes)) {
        // uint8 / uint16 keys: widen, sort the low key_bytes digits as u32 keys (values ride along as usual), narrow
        if (n >= 0xFFFFFFFFull || have_hist0 || !(native_value(val_bytes) || wide_value(val_bytes))) return COL_EINVAL;
        if (n == 0) return COL_OK;
        if (!scratch) return COL_ENOSCRATCH;
        hipStream_t s = col_stream(stream);
        char *q = (char *)scratch + col_radix_scratch_bytes(n, 4, val_bytes);
        u32 *wide_in = (u32 *)q, *wide_out = (u32 *)(q + align256((size_t)n * 4));
        dim3 grid((unsigned)col_ceil_div(n, 256)), block(256);
        if (key_bytes == 1) k_widen<uint8_t><<<grid, block, 0, s>>>((const uint8_t *)keys, wide_in, n);
        else k_widen<uint16_t><<<grid, block, 0, s>>>((const uint16_t *)keys, wide_in, n);
        COL_LAUNCH_OK();
        int rc;
        if (wide_value(val_bytes)) {
            char *q2 = (char *)scratch + col_radix_scratch_bytes(n, 4, 4);
            u32 *iota = (u32 *)q2, *sorted_idx = (u32 *)(q2 + align256((size_t)n * 4));
            k_iota<<<grid, block, 0, s>>>(iota, n);
            COL_LAUNCH_OK();
            if ((rc = sort_passes(stream, wide_in, wide_out, iota, sorted_idx, n, 4, 4, scratch, 0, 0, key_bytes))) return rc;
            if ((rc = gather_values(s, vals, sorted_idx, vals_out, n, val_bytes))) return rc;
        } else if ((rc = sort_passes(stream, wide_in, wide_out, vals, vals_out, n, 4, val_bytes, scratch, 0, 0, key_bytes))) return rc;
        if (key_bytes == 1) k_narrow<uint8_t><<<grid, block, 0, s>>>(wide_out, (uint8_t *)keys_out, n);
        else k_narrow<uint16_t><<<grid, block, 0, s>>>(wide_out, (uint16_t *)keys_out, n);
        COL_LAUNCH_OK();
        if (copy_back) {
            COL_HIP(hipMemcpyAsync((void *)keys, keys_out, (size_t)n * key_bytes, hipMemcpyDeviceToDevice, s));
            if (val_bytes) COL_HIP(hipMemcpyAsync((void *)vals, vals_out, (size_t)n * val_bytes, hipMemcpyDeviceToDevice, s));
        }
        return COL_OK;
    }
    return sort_passes(stream, keys, keys_out, vals, vals_out, n, key_bytes, val_bytes, scratch, copy_back, have_hist0, key_bytes);
}

// `passes` 8-bit digit passes from the least significant byte (= key_bytes for a full sort; fewer when the upper
// key bytes are known to be zero); the last pass lands in *_out.
static int sort_passes(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                       uint64_t n, int key_bytes, int val_bytes, void *scratch, int copy_back, int have_hist0, int passes) {
    if (wide_value(val_bytes)) {
        // values of a width the scatter kernel does not move: sort (key, index), gather the values once
        if (n >= 0xFFFFFFFFull || (key_bytes != 4 && key_bytes != 8)) return COL_EINVAL;
        if (n == 0) return COL_OK;
        if (!scratch) return COL_ENOSCRATCH;
        hipStream_t s = col_stream(stream);
        char *q = (char *)scratch + col_radix_scratch_bytes(n, key_bytes, 4);
        u32 *iota = (u32 *)q, *sorted_idx = (u32 *)(q + align256((size_t)n * 4));
        k_iota<<<dim3((unsigned)col_ceil_div(n, 256)), dim3(256), 0, s>>>(iota, n);
        COL_LAUNCH_OK();
        int rc = sort_passes(stream, keys, keys_out, iota, sorted_idx, n, key_bytes, 4, scratch, 0, 0, passes);
        if (rc) return rc;
        if ((rc = gather_values(s, vals, sorted_idx, vals_out, n, val_bytes))) return rc;
        if (copy_back) {
            COL_HIP(hipMemcpyAsync((void *)keys, keys_out, (size_t)n * key_bytes, hipMemcpyDeviceToDevice, s));
            COL_HIP(hipMemcpyAsync((void *)vals, vals_out, (size_t)n * val_bytes, hipMemcpyDeviceToDevice, s));
        }
        return COL_OK;
    }
    if (bad_sizes(n, key_bytes, val_bytes)) return COL_EINVAL;
    if (n == 0) return COL_OK;
    if (!scratch) return COL_ENOSCRATCH;
    hipStream_t s = col_stream(stream);
    const size_t nb = tiles_of(n, key_bytes, val_bytes);
    char *p = (char *)scratch;
    u32 *hist = (u32 *)p;              p += align256((size_t)RDIG * nb * sizeof(u32));
    void *scan_scratch = p;            p += align256(col_scan_scratch_bytes((uint64_t)RDIG * nb));
    void *tmp_keys = p;                p += align256((size_t)n * key_bytes);
    void *tmp_vals = p;                // (carved for THIS n: never more than col_radix_scratch_bytes(n) in total)
    const void *src_k = keys, *src_v = vals;       // 8-bit digits; the destinations alternate so that the last pass lands in *_out
    for (int pass = 0; pass < passes; pass++) {
        void *dst_k = ((passes - 1 - pass) & 1) ? tmp_keys : keys_out;
        void *dst_v = ((passes - 1 - pass) & 1) ? tmp_vals : vals_out;
        int rc = (pass == 0 && have_hist0) ? COL_OK : col_radix_histogram(stream, src_k, n, key_bytes, val_bytes, pass, hist);
        if (rc) return rc;
        rc = col_scan_u32(stream, hist, (uint64_t)RDIG * nb, scan_scratch);
        if (rc) return rc;
        rc = col_radix_scatter(stream, src_k, dst_k, src_v, dst_v, n, key_bytes, val_bytes, pass, hist);
        if (rc) return rc;
        src_k = dst_k;
        src_v = dst_v;
    }
    if (copy_back) {   // radix.py:158-169 leaves a sorted copy in the input buffers too
        COL_HIP(hipMemcpyAsync((void *)keys, keys_out, (size_t)n * key_bytes, hipMemcpyDeviceToDevice, s));
        if (val_bytes) COL_HIP(hipMemcpyAsync((void *)vals, vals_out, (size_t)n * val_bytes, hipMemcpyDeviceToDevice, s));
    }
    return COL_OK;
}

int col_ref_block_sort(void *stream, void *keys, void *vals, uint64_t n, int key_bytes, int val_bytes,
                       uint32_t block, int bits, int pass, uint32_t *hist) {
    if ((key_bytes != 4 && key_bytes != 8) || block == 0 || n % block || bits < 1 || bits > 12) return COL_EINVAL;
    if (!vals) val_bytes = 0;
    if (n == 0) return COL_OK;
    const u32 nb = (u32)(n / block);
    const size_t lds = (((size_t)4 << bits) + 15 & ~(size_t)15) + (size_t)block * (key_bytes + val_bytes);
    if (lds > 64 * 1024) return COL_EINVAL;
    if (key_bytes == 4)
        k_ref_block_sort<uint32_t><<<dim3(nb), dim3(COL_WAVE), lds, col_stream(stream)>>>(
            (uint32_t *)keys, (unsigned char *)vals, block, (u32)val_bytes, bits, pass, nb, hist);
    else
        k_ref_block_sort<uint64_t><<<dim3(nb), dim3(COL_WAVE), lds, col_stream(stream)>>>(
            (uint64_t *)keys, (unsigned char *)vals, block, (u32)val_bytes, bits, pass, nb, hist);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_ref_scatter(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                    uint64_t n, int key_bytes, int val_bytes, uint32_t block, int bits, int pass,
                    const uint32_t *offsets, const uint32_t *hist) {
    if ((key_bytes != 4 && key_bytes != 8) || block == 0 || n % block || bits < 1 || bits > 12) return COL_EINVAL;
    if (n == 0) return COL_OK;
    const u32 nb = (u32)(n / block);
    const size_t lds = (size_t)8 << bits;
    if (key_bytes == 4)
        k_ref_scatter<uint32_t><<<dim3(nb), dim3(COL_WAVE), lds, col_stream(stream)>>>(
            (const uint32_t *)keys, (uint32_t *)keys_out, (const unsigned char *)vals, (unsigned char *)vals_out, block,
            (u32)val_bytes, bits, pass, nb, offsets, hist);
    else
        k_ref_scatter<uint64_t><<<dim3(nb), dim3(COL_WAVE), lds, col_stream(stream)>>>(
            (const uint64_t *)keys, (uint64_t *)keys_out, (const unsigned char *)vals, (unsigned char *)vals_out, block,
            (u32)val_bytes, bits, pass, nb, offsets, hist);
    COL_LAUNCH_OK();
    return COL_OK;
}

}  // extern "C"

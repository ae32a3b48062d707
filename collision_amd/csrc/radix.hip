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
// Tile classes (threads x items per thread), template parameters of the kernels:
//   SMALL 256 x 4  = 1024 pairs  below SMALL_N pairs: a pass is bound by the latency of one block, and
//                                more, shorter blocks finish sooner (the 1 M-sphere path);
//   MID   256 x 16 = 4096 pairs  up to BIG_N pairs, and for 8-byte keys (LDS);
//   BIG   512 x 16 = 8192 pairs  above: the digit runs of a tile are ~128 bytes, a full L2 line, so the
//                                stores no longer depend on the runs of neighbouring tiles meeting in L2.
constexpr int IT_BIG = 16, IT_SMALL = 4;
constexpr int NT_BIG = 512, NT_MID = 256, NT_SMALL = 256;
constexpr uint64_t SMALL_N = 1u << 20, BIG_N = 16u << 20;   // tools/radix_tile_sweep.py
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
template <typename K, int TILE>
__global__ __launch_bounds__(RT) void k_hist(const K *__restrict__ keys, uint64_t n, u32 nblocks, u32 g,
                                             int shift, u32 *__restrict__ hist) {
    constexpr int IT = TILE / RT;
    __shared__ u32 h[HG * RDIG];
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < g * RDIG; i += RT) h[i] = 0;
    __syncthreads();
    const u32 b0 = blockIdx.x * g;
    for (u32 t = 0; t < g && b0 + t < nblocks; t++) {
        const uint64_t base = (uint64_t)(b0 + t) * TILE;
        u32 *ht = h + t * RDIG;
        constexpr int VEC = 16 / sizeof(K);          // keys per 16-byte load
#pragma unroll
        for (int k = 0; k < IT / VEC; k++) {
            const uint64_t i = base + ((uint64_t)k * RT + tid) * VEC;
            if (i + VEC <= n) {
                K kk[VEC];
                *reinterpret_cast<uint4 *>(kk) = *reinterpret_cast<const uint4 *>(keys + i);
#pragma unroll
                for (int j = 0; j < VEC; j++) {
                    // sorted or low-entropy inputs put one digit in every lane; 64 LDS atomics on one
                    // address would serialise, so a wave-uniform digit is counted with a single add
                    const u32 d = digit_of(kk[j], shift);
                    const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
                    // (only the lanes inside this branch take part: the ragged last tile is partly elsewhere)
                    const u64 active = __ballot(true);
                    if (__ballot(d != d0) == 0) {
                        if (lane_id() == (u32)__builtin_ctzll(active)) atomicAdd(&ht[d0], (u32)__popcll(active));
                    } else atomicAdd(&ht[d], 1u);
                }
            } else {
                for (int j = 0; j < VEC; j++)
                    if (i + j < n) atomicAdd(&ht[digit_of(keys[i + j], shift)], 1u);
            }
        }
    }
    __syncthreads();
    // thread d writes its row of up to g entries (contiguous: full-sector writes)
    const u32 cnt = min(g, nblocks - b0);
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
    __shared__ u32 s_cnt[NW][RDIG];
    __shared__ u32 s_goff[RDIG];
    __shared__ u32 s_ws[NW];

    const V *vals_in = reinterpret_cast<const V *>(vals_in_);
    V *vals_out = reinterpret_cast<V *>(vals_out_);
    const u32 tid = threadIdx.x, lane = tid & (COL_WAVE - 1), w = tid / COL_WAVE;
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an
    // XCD), each with its own L2.  Consecutive tiles write ADJACENT runs of every digit, so give each
    // XCD a contiguous range of tiles: the ~64-byte runs of neighbouring tiles then meet in one L2 and
    // leave it as full lines instead of partial-line writes from 8 different L2s.  (Speed only.)
    u32 b = blockIdx.x;
    if (!(dbg & 4)) {
        const u32 q = nblocks / 8, r = nblocks % 8, xcd = b % 8;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    const uint64_t tile_base = (uint64_t)b * TILE;
    const u32 valid = (u32)min((uint64_t)TILE, n - tile_base);

    unsigned long long t_prev = (DIAG && (dbg & 32)) ? __builtin_amdgcn_s_memtime() : 0ull;
    (void)t_prev;
    // this tile's global offset of digit `tid`: one scattered 4-byte load per thread, issued now so that
    // its latency hides under the loads and the ranking instead of sitting between two barriers
    const u32 my_offset = tid < RDIG ? offsets[(uint64_t)tid * nblocks + b] : 0u;
    for (u32 i = tid; i < NW * RDIG; i += NT) (&s_cnt[0][0])[i] = 0;

    const u32 wbase = w * (COL_WAVE * IT) + lane;
    K key[IT];
    V val[V_LDS ? IT : 1];
    constexpr int KV = 16 / sizeof(K);                    // keys per 16-byte load
    if (valid == (u32)TILE && !(dbg & 8)) {
        // Full tile: 16-byte global loads (lane l takes KV consecutive keys), transposed to the
        // lane-striped order the ranking needs through this wave's own slice of the LDS staging
        // area.  LDS operations of one wave execute in order, so no barrier is needed.
        // All global loads of the tile (keys AND values) are issued before the first wait: one memory
        // round trip per tile instead of two.
        K *stage = s_keys + w * (COL_WAVE * IT);
        const K *src = keys_in + tile_base + w * (COL_WAVE * IT);
        constexpr int VV = 16 / sizeof(V);
        V *vstage = s_vals + w * (COL_WAVE * IT);
        const V *vsrc = vals_in + tile_base + w * (COL_WAVE * IT);
        // The loads are asm statements: hipcc sinks a plain second group of loads below the first
        // group's LDS writes (to save registers), which serialises two round trips.  hipcc does not
        // count asm loads, so every destination passes through an explicit vmcnt(0) statement before
        // its first use (cdna_hip_programming.md 5.7, form ii).
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        v4u kq[IT / KV], vq[V_LDS ? IT / VV : 1];
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
#pragma unroll
        for (int j = 0; j < IT / KV; j++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(kq[j])::"memory");
        if (V_LDS) {
#pragma unroll
            for (int j = 0; j < IT / VV; j++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(vq[j])::"memory");
        }
#pragma unroll
        for (int j = 0; j < IT / KV; j++)
            *reinterpret_cast<v4u *>(stage + j * (COL_WAVE * KV) + lane * KV) = kq[j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < IT; k++) key[k] = stage[k * COL_WAVE + lane];
        if (V_LDS) {
#pragma unroll
            for (int j = 0; j < IT / VV; j++)
                *reinterpret_cast<v4u *>(vstage + j * (COL_WAVE * VV) + lane * VV) = vq[j];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < IT; k++) val[k] = vstage[k * COL_WAVE + lane];
        }
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
    }
    __syncthreads();
    STAMP(0)      // load + transpose

    // rank inside (wave, digit): wave-private counters, program order keeps them consistent
    u32 pos[IT];
#pragma unroll
    for (int k = 0; k < IT; k++) {
        const u32 d = digit_of(key[k], shift);
        const u64 peers = match8(d);
        const u32 below = mbcnt(peers);
        // every lane of the group reads the counter (same address: a broadcast), then the group's
        // lowest lane bumps it; LDS operations of one wave execute in order, so no lane sees the bump
        const u32 prev = s_cnt[w][d];
        if (below == 0) s_cnt[w][d] = prev + (u32)__popcll(peers);
        pos[k] = prev + below;
    }
    __syncthreads();
    STAMP(1)      // ranking

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
            for (int i = 0; i < NW; i++) { s_cnt[i][tid] = run; run += c[i]; }
            // global position of tile-sorted slot i with digit d is s_goff[d] + i
            s_goff[tid] = my_offset - dstart;
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

#pragma unroll
    for (int k = 0; k < IT; k++) {
        const u32 i = k * NT + tid;
        if (i < valid) {
            const K kk = s_keys[i];
            u32 g = s_goff[digit_of(kk, shift)] + i;
            if (dbg & 2) g = (u32)tile_base + i;       // timing ablation: coalesced output
            if (dbg & 384) {                           // timing ablations: stores at system (128) / agent (256) scope
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


// ---- scatter, big inputs: persistent blocks, LDS-DMA double buffering ----
// k_scatter above is one tile per workgroup: load (a third of the block's cycles is the wait for HBM),
// rank, stage, store -- and nothing of the next tile is in flight meanwhile; two such blocks per CU
// overlap only by chance.  Here ONE 1024-thread workgroup per CU walks its tiles, and the NEXT tile is
// always on its way into the second LDS buffer:
//   * loads are LDS-DMA (global_load_lds_dwordx4: no VGPR destination, so nothing for the compiler to
//     spill or wait on), each wave fetching its own 512-pair slice; the lane-striped order the ranking
//     needs is then just a strided ds_read of that slice (no ds_write_b128 transposition pass);
//   * per tile: read slice -> rank -> [B1] digit scan [B2a, B2b] -> write (key, value) PAIRS at their tile-sorted
//     slot (one ds_write_b64) -> [B3] read back (ds_read_b64) -> [B4] wait for the DMA issued one
//     tile ago, issue the DMA of the tile after next into the buffer just freed, store this tile;
//   * barriers are raw s_barrier + lgkmcnt(0): a __syncthreads() would drain the stores and the DMA;
//   * the tile's 256 scanned offsets arrive by LDS-DMA too (one dword per lane, waves 0-3), so the
//     loop has no VGPR-destination load at all and the only vmcnt wait is the one above, for
//     operations issued a whole tile earlier.
// XCD-aware schedule as in k_scatter: every XCD owns a contiguous range of tiles and its workgroups
// stride through it together, so neighbouring tiles' ~128-byte runs meet in one L2.
constexpr int PT = 1024, PIT = 8, PNW = PT / COL_WAVE;
constexpr int PTILE = PT * PIT;                // 8192 pairs, the BIG tile
constexpr int PROW = COL_WAVE * PIT;           // pairs per wave

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ u32 lds_addr(const void *p) {
    return (u32)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// one LDS-DMA: lane l moves 16 (4) bytes from its own global address to lds_base + 16 (4) * l
__device__ __forceinline__ void glds16(const void *gsrc, u32 lds_base) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, u32 lds_base) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

template <bool HAS_V> struct ScatterPLds {
    static constexpr int BUFW = HAS_V ? 2 * PTILE : PTILE;      // dwords per buffer: keys, then values
    __attribute__((aligned(16))) u32 buf[2][BUFW];
    u32 cnt[PNW][RDIG];                                         // wave-private digit counters
    u32 raw[2][RDIG];                                           // the tile's scanned offsets, as fetched
    u32 goff[RDIG];
    u32 ws[RDIG / COL_WAVE];
};

// One tile of k_scatter_p from "my slice is in registers" to "sorted pairs and their global slots are in
// registers": rank, digit scan, stage through `buf` (in place of the slices), read back.
template <bool HAS_V, int VAR>
__device__ __forceinline__ void scatter_p_tile(ScatterPLds<HAS_V> &L, u32 *buf, const u32 *raw, const u32 (&key)[PIT],
                                               const u32 (&val)[HAS_V ? PIT : 1], int shift, u32 tile_base, u32 tid, u32 lane,
                                               u32 w, u32 (&okey)[PIT], u32 (&oval)[HAS_V ? PIT : 1], u32 (&g)[PIT]) {
    // rank inside (wave, digit): wave-private counters, program order keeps them consistent
    u32 pos[PIT];
#pragma unroll
    for (int k = 0; k < PIT; k++) {
        const u32 d = digit_of(key[k], shift);
        const u64 peers = match8(d);
        const u32 below = mbcnt(peers);
        const u32 prev = L.cnt[w][d];
        if (below == 0) L.cnt[w][d] = prev + (u32)__popcll(peers);
        pos[k] = prev + below;
    }
    lds_barrier();                                                            // B1: counters complete, slices read

    // digit `tid` (waves 0-3): exclusive over waves, then over digits
    u32 c[PNW], tot = 0, incl = 0;
    if (tid < RDIG) {
#pragma unroll
        for (int i = 0; i < PNW; i++) { c[i] = L.cnt[i][tid]; tot += c[i]; }
        incl = wave_incl_scan(tot);
        if (lane == COL_WAVE - 1) L.ws[w] = incl;
    }
    lds_barrier();                                                            // B2a
    if (tid < RDIG) {
        u32 run = incl - tot;
#pragma unroll
        for (int i = 0; i < RDIG / COL_WAVE; i++) run += ((u32)i < w ? L.ws[i] : 0u);
        L.goff[tid] = raw[tid] - run;       // global position of tile-sorted slot i with digit d is goff[d] + i
#pragma unroll
        for (int i = 0; i < PNW; i++) { L.cnt[i][tid] = run; run += c[i]; }
    }
    lds_barrier();                                                            // B2b

    // (key, value) PAIRS to their tile-sorted slot: one ds_write_b64 per item
#pragma unroll
    for (int k = 0; k < PIT; k++) {
        const u32 p = pos[k] + L.cnt[w][digit_of(key[k], shift)];
        if (HAS_V) reinterpret_cast<uint2 *>(buf)[p] = make_uint2(key[k], val[k]);
        else buf[p] = key[k];
    }
    lds_barrier();                                                            // B3

#pragma unroll
    for (int k = 0; k < PIT; k++) {
        const u32 i = k * PT + tid;
        if (HAS_V) {
            const uint2 kv = reinterpret_cast<const uint2 *>(buf)[i];
            okey[k] = kv.x; oval[k] = kv.y;
        } else okey[k] = buf[i];
        g[k] = L.goff[digit_of(okey[k], shift)] + i;
        if (VAR & 1) g[k] = tile_base + i;                                   // timing ablation: coalesced output
    }
    for (u32 i = lane; i < RDIG; i += COL_WAVE) L.cnt[w][i] = 0;            // for the next tile (wave-private)
    lds_barrier();                                                            // B4: `buf` is free again
}

// VAR (timing ablations, col_debug_radix): bit 0 = coalesced output, bit 1 = wait for everything (stores
// too) before the next DMA instead of the counted wait.
template <bool HAS_V, int VAR>
__global__ __launch_bounds__(PT) void k_scatter_p(const u32 *__restrict__ keys_in, u32 *__restrict__ keys_out,
                                                  const u32 *__restrict__ vals_in, u32 *__restrict__ vals_out,
                                                  uint64_t n, u32 nblocks, int shift, const u32 *__restrict__ offsets) {
    __shared__ ScatterPLds<HAS_V> L;
    const u32 tid = threadIdx.x, lane = tid & (COL_WAVE - 1);
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(tid / COL_WAVE));

    // this workgroup's tiles: first, first + step, ... < end
    u32 first, step, end;
    if (gridDim.x >= 8) {
        const u32 q = nblocks / 8, r = nblocks % 8, xcd = blockIdx.x % 8;
        const u32 lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        first = lo + blockIdx.x / 8;
        step = (gridDim.x + 7 - xcd) / 8;
        end = lo + q + (xcd < r ? 1u : 0u);
    } else {
        first = blockIdx.x; step = gridDim.x; end = nblocks;
    }
    // The pipelined loop takes FULL tiles only (it has no VGPR-destination load, so hipcc places no vmcnt
    // wait in it); the ragged last tile of the input, if this workgroup owns it, follows the loop.
    const u32 n_full = (u32)(n / PTILE);
    const u32 end_full = end < n_full ? end : n_full;

    // DMA of a full `tile` into buffer `sel`: this wave's slice of the keys (and values), plus (waves 0-3)
    // 64 of the tile's 256 scanned offsets
    auto issue = [&](u32 tile, u32 sel) {
        if (tile >= end_full) return;
        const uint64_t base = (uint64_t)tile * PTILE + w * PROW;
#pragma unroll
        for (int j = 0; j < PROW / 256; j++) {
            glds16(keys_in + base + j * 256 + lane * 4, lds_addr(&L.buf[sel][w * PROW + j * 256]));
            if (HAS_V) glds16(vals_in + base + j * 256 + lane * 4, lds_addr(&L.buf[sel][PTILE + w * PROW + j * 256]));
        }
        if (w < RDIG / COL_WAVE) glds4(offsets + (uint64_t)tid * nblocks + tile, lds_addr(&L.raw[sel][w * COL_WAVE]));
    };

    for (u32 i = lane; i < RDIG; i += COL_WAVE) L.cnt[w][i] = 0;
    issue(first, 0);
    issue(first + step, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    u32 sel = 0;
    for (u32 tile = first; tile < end_full; tile += step, sel ^= 1) {
        u32 *buf = L.buf[sel];
        // my slice, lane-striped: (wave, item, lane) order == memory order, stability is positional
        u32 key[PIT], val[HAS_V ? PIT : 1], okey[PIT], oval[HAS_V ? PIT : 1], g[PIT];
#pragma unroll
        for (int k = 0; k < PIT; k++) key[k] = buf[w * PROW + k * COL_WAVE + lane];
        if (HAS_V) {
#pragma unroll
            for (int k = 0; k < PIT; k++) val[k] = buf[PTILE + w * PROW + k * COL_WAVE + lane];
        }
        scatter_p_tile<HAS_V, VAR>(L, buf, L.raw[sel], key, val, shift, tile * (u32)PTILE, tid, lane, w, okey, oval, g);
        // The DMA of the next tile and the stores of the previous one were issued a whole tile ago, in that
        // order: waiting for all but the youngest 2 * PIT (the stores) operations waits for the DMA only.
        if (VAR & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HAS_V ? 2 * PIT : PIT) : "memory");
        issue(tile + 2 * step, sel);
#pragma unroll
        for (int k = 0; k < PIT; k++) {
            keys_out[g[k]] = okey[k];
            if (HAS_V) vals_out[g[k]] = oval[k];
        }
    }

    if (end == nblocks && n_full < nblocks && (nblocks - 1 - first) % step == 0 && nblocks - 1 >= first) {
        // the ragged last tile: loaded synchronously; pads sort last (positional stability) and are never stored
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32 tile = nblocks - 1;
        const uint64_t tile_base = (uint64_t)tile * PTILE;
        const u32 valid = (u32)(n - tile_base);
        u32 key[PIT], val[HAS_V ? PIT : 1], okey[PIT], oval[HAS_V ? PIT : 1], g[PIT];
#pragma unroll
        for (int k = 0; k < PIT; k++) {
            const u32 li = w * PROW + k * COL_WAVE + lane;
            key[k] = li < valid ? keys_in[tile_base + li] : 0xFFFFFFFFu;
            if (HAS_V) val[k] = li < valid ? vals_in[tile_base + li] : 0u;
        }
        if (tid < RDIG) L.raw[0][tid] = offsets[(uint64_t)tid * nblocks + tile];     // read back by the same thread
        scatter_p_tile<HAS_V, VAR>(L, L.buf[0], L.raw[0], key, val, shift, (u32)tile_base, tid, lane, w, okey, oval, g);
#pragma unroll
        for (int k = 0; k < PIT; k++) {
            if ((u32)k * PT + tid < valid) {
                keys_out[g[k]] = okey[k];
                if (HAS_V) vals_out[g[k]] = oval[k];
            }
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
constexpr int BS_NT = 1024, BS_IT = 8, BS_NW = BS_NT / COL_WAVE;
constexpr u32 BS_CAP = BS_NT * BS_IT;          // 8192 pairs
constexpr int BS_SHIFT = 22;                   // bucket digit = bits 22..29 of a 30-bit Morton code (pads: 255)
constexpr int BS_ROW = COL_WAVE * BS_IT;       // positions per wave

struct BucketLds {
    u32 keys[BS_CAP];
    u32 vals[BS_CAP];
    u32 cnt[BS_NW][RDIG];
    u32 base[RDIG];
    u32 ws[BS_NW];
};

// One stable pass over the (up to BS_CAP) pairs held lane-striped in registers: position of item k of
// this thread = w * row_len + k * 64 + lane (k * 64 < row_len); rows at or beyond `m` hold pads (key
// 0xFFFFFFFF) and are skipped.  On return pos[k] = rank of the item inside its (wave, digit) group and lds.cnt[i][d] = number
// of ranked items (pads of a partly valid row included) of wave i with digit d.
__device__ __forceinline__ void bucket_rank(BucketLds &lds, const u32 (&key)[BS_IT], u32 (&pos)[BS_IT], u32 m, int shift,
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

__global__ __launch_bounds__(BS_NT) void k_bucket_sort(u32 *__restrict__ k_a, u32 *__restrict__ v_a, u32 *__restrict__ k_b,
                                                        u32 *__restrict__ v_b, u32 n, const u32 *__restrict__ offsets,
                                                        u32 nblocks, u32 *oversize) {
    __shared__ BucketLds lds;
    const u32 tid = threadIdx.x, lane = tid & (COL_WAVE - 1);
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(tid / COL_WAVE));
    const u32 d = blockIdx.x;
    const u32 start = offsets[(uint64_t)d * nblocks], end = d + 1 < RDIG ? offsets[(uint64_t)(d + 1) * nblocks] : n;
    const u32 S = end - start;
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
            bucket_rank(lds, key, pos, S, shift, w, tid, L);
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
            bucket_rank(lds, key, pos, m, shift, w, tid, (u32)BS_ROW);
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
int g_radix_tile_override = 0;      // 0 = automatic, else 1024 / 4096 / 8192
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline u32 tile_auto(uint64_t n, int key_bytes) {
    u32 t = n < SMALL_N ? (u32)(NT_SMALL * IT_SMALL) : n < BIG_N ? (u32)(NT_MID * IT_BIG) : (u32)(NT_BIG * IT_BIG);
    if (key_bytes == 8 && t > (u32)(NT_MID * IT_BIG)) t = (u32)(NT_MID * IT_BIG);
    return t;
}
inline u32 tile_for(uint64_t n, int key_bytes) {
    if (!g_radix_tile_override) return tile_auto(n, key_bytes);
    u32 t = (u32)g_radix_tile_override;
    if (key_bytes == 8 && t > (u32)(NT_MID * IT_BIG)) t = (u32)(NT_MID * IT_BIG);
    return t;
}
inline u32 tiles_of(uint64_t n, int key_bytes) { return (u32)col_ceil_div(n, tile_for(n, key_bytes)); }
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
int launch_hist(hipStream_t s, const void *keys, uint64_t n, int pass, u32 *hist) {
    const u32 tile = tile_for(n, (int)sizeof(K));
    const u32 nb = tiles_of(n, (int)sizeof(K)), g = hist_group(nb);
    dim3 grid((unsigned)col_ceil_div(nb, g)), block(RT);
    const K *k = (const K *)keys;
    if (tile == (u32)(NT_SMALL * IT_SMALL)) k_hist<K, NT_SMALL * IT_SMALL><<<grid, block, 0, s>>>(k, n, nb, g, pass * 8, hist);
    else if (tile == (u32)(NT_MID * IT_BIG)) k_hist<K, NT_MID * IT_BIG><<<grid, block, 0, s>>>(k, n, nb, g, pass * 8, hist);
    else k_hist<K, NT_BIG * IT_BIG><<<grid, block, 0, s>>>(k, n, nb, g, pass * 8, hist);
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

// the persistent LDS-DMA kernel: u32 keys, no or 4-byte values, the BIG tile
constexpr int P_MAX_BLOCKS = 256;          // one 1024-thread workgroup (147 KB of LDS) per CU
template <bool HAS_V>
int launch_scatter_p(hipStream_t s, const u32 *ki, u32 *ko, const u32 *vi, u32 *vo, uint64_t n, int shift, const u32 *offsets) {
    const u32 nb = (u32)col_ceil_div(n, PTILE);
    dim3 grid(nb < (u32)P_MAX_BLOCKS ? nb : (u32)P_MAX_BLOCKS), block(PT);
    const int var = (g_radix_dbg & 2 ? 1 : 0) | (g_radix_dbg & 2048 ? 2 : 0);
    switch (var) {
    case 0: k_scatter_p<HAS_V, 0><<<grid, block, 0, s>>>(ki, ko, vi, vo, n, nb, shift, offsets); break;
    case 1: k_scatter_p<HAS_V, 1><<<grid, block, 0, s>>>(ki, ko, vi, vo, n, nb, shift, offsets); break;
    case 2: k_scatter_p<HAS_V, 2><<<grid, block, 0, s>>>(ki, ko, vi, vo, n, nb, shift, offsets); break;
    default: k_scatter_p<HAS_V, 3><<<grid, block, 0, s>>>(ki, ko, vi, vo, n, nb, shift, offsets); break;
    }
    COL_LAUNCH_OK();
    return COL_OK;
}

template <typename K>
int launch_scatter(hipStream_t s, const void *keys, void *keys_out, const void *vals, void *vals_out,
                   uint64_t n, int vb, int shift, const u32 *offsets) {
    const u32 tile = tile_for(n, (int)sizeof(K));
    if (tile == (u32)(NT_SMALL * IT_SMALL))
        return launch_scatter_it<K, IT_SMALL, NT_SMALL>(s, keys, keys_out, vals, vals_out, n, vb, shift, offsets);
    if (tile == (u32)(NT_MID * IT_BIG))
        return launch_scatter_it<K, IT_BIG, NT_MID>(s, keys, keys_out, vals, vals_out, n, vb, shift, offsets);
    if constexpr (sizeof(K) == 4) {
        const bool has_v = vals && vals_out && vb;
        if ((!has_v || vb == 4) && !(g_radix_dbg & 4096)) {      // mode 4096: the one-tile-per-workgroup kernel instead
            if (has_v) return launch_scatter_p<true>(s, (const u32 *)keys, (u32 *)keys_out, (const u32 *)vals, (u32 *)vals_out, n, shift, offsets);
            return launch_scatter_p<false>(s, (const u32 *)keys, (u32 *)keys_out, nullptr, nullptr, n, shift, offsets);
        }
        return launch_scatter_it<K, IT_BIG, NT_BIG>(s, keys, keys_out, vals, vals_out, n, vb, shift, offsets);
    }
    return COL_EINVAL;
}

inline bool bad_sizes(uint64_t n, int key_bytes, int val_bytes) {
    if (key_bytes != 4 && key_bytes != 8) return true;
    if (val_bytes != 0 && val_bytes != 4 && val_bytes != 8 && val_bytes != 16 && val_bytes != 32) return true;
    return n >= 0xFFFFFFFFull;   // offsets are uint32 (as in the reference)
}

}  // namespace

extern "C" {

void col_debug_radix(int mode) { g_radix_dbg = mode; }

int col_debug_radix_tile(int tile) {
    if (tile != 0 && tile != NT_SMALL * IT_SMALL && tile != NT_MID * IT_BIG && tile != NT_BIG * IT_BIG) return COL_EINVAL;
    g_radix_tile_override = tile;
    return COL_OK;
}

int col_debug_radix_stamps(uint64_t *out, int reset) {
    unsigned long long h[8] = {0};
    if (out) { COL_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h))); for (int i = 0; i < 8; i++) out[i] = h[i]; }
    if (reset) { unsigned long long z[8] = {0}; COL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z))); }
    return COL_OK;
}

uint32_t col_radix_tile(uint64_t n, int key_bytes, int val_bytes) { (void)val_bytes; return tile_for(n, key_bytes); }

// Monotone in n: sized for the largest histogram any n' <= n can need (see max_tiles_upto), so a scratch
// buffer sized for n serves every smaller sort / col_collide call as well.
size_t col_radix_scratch_bytes(uint64_t n, int key_bytes, int val_bytes) {
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
    return key_bytes == 4 ? launch_hist<uint32_t>(col_stream(stream), keys, n, pass, hist)
                          : launch_hist<uint64_t>(col_stream(stream), keys, n, pass, hist);
}

int col_radix_scatter(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                      uint64_t n, int key_bytes, int val_bytes, int pass, const uint32_t *offsets) {
    if (bad_sizes(n, key_bytes, val_bytes) || pass < 0 || pass >= key_bytes) return COL_EINVAL;
    if (n == 0) return COL_OK;
    return key_bytes == 4
               ? launch_scatter<uint32_t>(col_stream(stream), keys, keys_out, vals, vals_out, n, val_bytes, pass * 8, offsets)
               : launch_scatter<uint64_t>(col_stream(stream), keys, keys_out, vals, vals_out, n, val_bytes, pass * 8, offsets);
}

// MSD sort of (u32 key, u32 value) pairs whose keys are 30-bit codes (or 0xFFFFFFFF pads), for inputs
// sorted with the 1024-pair tile.  The histogram of the bucket digit (bits 22..29, digit-major, one row
// entry per 1024-pair tile) must already be at the start of `scratch` (col_morton_tile writes it).
// Same result as col_radix_sort.  *oversize (device-visible, may be NULL) receives the size of a bucket
// that did not fit LDS, if any.
int col_radix_sort_msd(void *stream, const uint32_t *keys, uint32_t *keys_out, const uint32_t *vals, uint32_t *vals_out,
                       uint64_t n, void *scratch, uint32_t *oversize) {
    if (n == 0) return COL_OK;
    if (!scratch || !vals || !vals_out) return COL_EINVAL;
    if (tile_auto(n, 4) != (u32)(NT_SMALL * IT_SMALL)) return COL_EINVAL;      // (a forced tile class does not apply here)
    hipStream_t s = col_stream(stream);
    const size_t nb = col_ceil_div(n, NT_SMALL * IT_SMALL);
    char *p = (char *)scratch;
    u32 *hist = (u32 *)p;              p += align256((size_t)RDIG * nb * sizeof(u32));
    void *scan_scratch = p;            p += align256(col_scan_scratch_bytes((uint64_t)RDIG * nb));
    u32 *tmp_keys = (u32 *)p;          p += align256((size_t)n * 4);
    u32 *tmp_vals = (u32 *)p;
    int rc = col_scan_u32(stream, hist, (uint64_t)RDIG * nb, scan_scratch);
    if (rc) return rc;
    rc = launch_scatter_it<u32, IT_SMALL, NT_SMALL>(s, keys, tmp_keys, vals, tmp_vals, n, 4, BS_SHIFT, hist);
    if (rc) return rc;
    k_bucket_sort<<<dim3(RDIG), dim3(BS_NT), 0, s>>>(tmp_keys, tmp_vals, keys_out, vals_out, (u32)n, hist, (u32)nb, oversize);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_radix_sort(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                   uint64_t n, int key_bytes, int val_bytes, void *scratch, int copy_back) {
    return col_radix_sort_ex(stream, keys, keys_out, vals, vals_out, n, key_bytes, val_bytes, scratch, copy_back, 0);
}

int col_radix_sort_ex(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                      uint64_t n, int key_bytes, int val_bytes, void *scratch, int copy_back, int have_hist0) {
    if (!vals || !vals_out) val_bytes = 0;
    if (bad_sizes(n, key_bytes, val_bytes)) return COL_EINVAL;
    if (n == 0) return COL_OK;
    if (!scratch) return COL_ENOSCRATCH;
    hipStream_t s = col_stream(stream);
    const size_t nb = tiles_of(n, key_bytes);
    char *p = (char *)scratch;
    u32 *hist = (u32 *)p;              p += align256((size_t)RDIG * nb * sizeof(u32));
    void *scan_scratch = p;            p += align256(col_scan_scratch_bytes((uint64_t)RDIG * nb));
    void *tmp_keys = p;                p += align256((size_t)n * key_bytes);
    void *tmp_vals = p;
    const int passes = key_bytes;      // 8-bit digits: 4 or 8 passes (even), so the last one lands in *_out
    const void *src_k = keys, *src_v = vals;
    for (int pass = 0; pass < passes; pass++) {
        void *dst_k = (pass & 1) ? keys_out : tmp_keys;
        void *dst_v = (pass & 1) ? vals_out : tmp_vals;
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

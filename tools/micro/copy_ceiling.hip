// Microbenchmark (diagnostics, not product): what does the MEMORY STRUCTURE of the radix scatter pass (csrc/radix.hip
// k_scatter, 64 Mi (u32 key, u32 value) pairs = 1 GiB moved per launch) allow on this box, with the ranking taken away?
//
//   A. plain copies of the same bytes: grid-stride float4 copy (one-shot and persistent grids, 1..8 vectors in flight per
//      thread, nt or not), hipMemcpyAsync D2D.
//   B. the pass's own shape -- one 512-thread workgroup per 8192-pair tile, every load of the tile issued up front, the tile
//      staged through 64 KB of LDS, dword stores -- replayed on the pass's tile-sorted image (so that an element's slot is its
//      index: no ranking), in these forms:
//        0 direct     loads -> dwordx4 stores at the same place (phased copy, no LDS)
//        1 coalesced  loads -> LDS -> dword stores at tile_base + i
//        2 scatter    loads -> LDS -> dword stores at the REAL addresses (256 runs of ~128 B per tile and array)
//        3 scatter, (key, value) pairs interleaved in ONE output array (a run is one ~256-byte piece)
//        4 pairs in -> pairs out, coalesced
//        5 pairs in -> pairs out, real addresses
//        6 pairs in -> separate arrays out, real addresses
//      each with XCD-contiguous tile ranges (blockIdx % 8 = XCD) or the plain order, as one workgroup per tile or as
//      PERSISTENT workgroups (2 per CU) that issue the loads of their next tile before they store the current one, and with
//      an optional spin between the loads and the stores (the ranking's time, with nothing in flight).
//
//   hipcc --offload-arch=gfx950 -O3 -o copy_ceiling tools/micro/copy_ceiling.hip && ./copy_ceiling [rounds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v2u __attribute__((ext_vector_type(2)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
    } while (0)

constexpr u32 N = 1u << 26;              // pairs
constexpr int TILE = 8192, NT = 512, IT = 16, NTILES = N / TILE;

__device__ __forceinline__ u32 hash32(u32 x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// ---------------- setup (not timed, simple) ----------------
__global__ void k_gen(u32 *keys, u32 *vals) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    keys[i] = hash32(i) & 0x3FFFFFFFu;
    vals[i] = i;
}
__global__ __launch_bounds__(NT) void k_tile_hist(const u32 *keys, u32 *hist_tile) {      // hist_tile[t][d]
    __shared__ u32 h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const u32 base = blockIdx.x * TILE;
    for (int k = 0; k < IT; k++) atomicAdd(&h[keys[base + k * NT + threadIdx.x] & 255u], 1u);
    __syncthreads();
    if (threadIdx.x < 256) hist_tile[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}
// the tile-sorted image: inside every tile the elements grouped by digit (order inside a group arbitrary)
__global__ __launch_bounds__(NT) void k_tile_sort(const u32 *keys, const u32 *vals, const u32 *hist_tile, u32 *sk, u32 *sv, v2u *skv) {
    __shared__ u32 start[256], cnt[256];
    if (threadIdx.x == 0) { u32 acc = 0; for (int d = 0; d < 256; d++) { start[d] = acc; acc += hist_tile[blockIdx.x * 256 + d]; } }
    if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
    __syncthreads();
    const u32 base = blockIdx.x * TILE;
    for (int k = 0; k < IT; k++) {
        const u32 i = base + k * NT + threadIdx.x;
        const u32 kk = keys[i], vv = vals[i], d = kk & 255u;
        const u32 slot = base + start[d] + atomicAdd(&cnt[d], 1u);
        sk[slot] = kk; sv[slot] = vv; skv[slot] = v2u{kk, vv};
    }
}
__global__ __launch_bounds__(NT) void k_reference(const u32 *sk, const u32 *sv, const u32 *goff, u32 *rk, u32 *rv) {
    const u32 base = blockIdx.x * TILE;
    for (int k = 0; k < IT; k++) {
        const u32 i = k * NT + threadIdx.x;
        const u32 kk = sk[base + i];
        const u32 g = goff[blockIdx.x * 256 + (kk & 255u)] + i;
        rk[g] = kk; rv[g] = sv[base + i];
    }
}
__global__ void k_compare(const u32 *a, const u32 *b, u64 n, u32 *bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(bad, 1u);
}
__global__ void k_compare_kv(const v2u *a, const u32 *bk, const u32 *bv, u64 n, u32 *bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (a[i].x != bk[i] || a[i].y != bv[i])) atomicAdd(bad, 1u);
}

// ---------------- A. plain copies ----------------
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const v4u *__restrict__ in, v4u *__restrict__ out, u64 nvec) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride * U) {
        v4u r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 j = i + u * stride;
            if (j < nvec) r[u] = NTL ? __builtin_nontemporal_load(in + j) : in[j];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 j = i + u * stride;
            if (j < nvec) { if (NTS) __builtin_nontemporal_store(r[u], out + j); else out[j] = r[u]; }
        }
    }
}

// ---------------- B. the pass's shape ----------------
__device__ __forceinline__ v4u gld4(const void *p) {
    v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ u32 gld1(const void *p) {
    u32 r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
template <int CNT> __device__ __forceinline__ void wait4(v4u &r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(CNT) : "memory"); }
template <int CNT> __device__ __forceinline__ void wait1(u32 &r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(CNT) : "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Args {
    const u32 *sk, *sv;      // tile-sorted image, separate arrays
    const v2u *skv;          // ... as pairs
    u32 *ok, *ov;            // outputs
    v2u *okv;
    const u32 *goff;         // goff[t][d]: global position of tile-sorted slot i with digit d is goff + i
    int spin;                // dummy VALU rounds between the loads and the stores
};

template <int MODE> struct ModeInfo {
    static constexpr bool KV_IN = MODE >= 4;
    static constexpr bool KV_OUT = MODE == 3 || MODE == 4 || MODE == 5;
    static constexpr bool SCATTER = MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6;
    static constexpr int STORES = MODE == 0 ? 8 : (KV_OUT ? IT : 2 * IT);      // store instructions per thread and tile
};

__device__ __forceinline__ u32 tile_of_block(u32 b, u32 nblocks, bool xcd) {
    if (!xcd) return b;
    const u32 q = nblocks / 8, x = b % 8;           // nblocks is a multiple of 8 here
    return x * q + b / 8;
}

// One tile: the loads are in q[] (issued by the caller), `pending` = vector-memory instructions issued after them.
template <int MODE, int PENDING>
__device__ __forceinline__ void finish_tile(const Args &a, u32 t, v4u (&q)[8], u32 &goff_reg, u32 *s_mem, u32 *s_goff) {
    typedef ModeInfo<MODE> M;
    const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u32 tile_base = t * TILE;
#pragma unroll
    for (int j = 0; j < 8; j++) wait4<PENDING>(q[j]);
    wait1<PENDING>(goff_reg);
    if (a.spin) {
        u32 dummy = tid;
        for (int s = 0; s < a.spin; s++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy));
        if (dummy == 0x12345678u) q[0].x ^= 1;
    }
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            *reinterpret_cast<v4u *>(a.ok + tile_base + w * 1024 + j * 256 + lane * 4) = q[j];
            *reinterpret_cast<v4u *>(a.ov + tile_base + w * 1024 + j * 256 + lane * 4) = q[4 + j];
        }
        return;
    }
    if (tid < 256) s_goff[tid] = goff_reg;
    if (!M::KV_IN) {
        u32 *s_k = s_mem, *s_v = s_mem + TILE;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            *reinterpret_cast<v4u *>(s_k + w * 1024 + j * 256 + lane * 4) = q[j];
            *reinterpret_cast<v4u *>(s_v + w * 1024 + j * 256 + lane * 4) = q[4 + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) *reinterpret_cast<v4u *>(s_mem + w * 2048 + j * 256 + lane * 4) = q[j];
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < IT; k++) {
        const u32 i = k * NT + tid;
        u32 kk, vv;
        if (!M::KV_IN) { kk = s_mem[i]; vv = s_mem[TILE + i]; }
        else { const v2u p = *reinterpret_cast<const v2u *>(s_mem + 2 * i); kk = p.x; vv = p.y; }
        const u32 g = M::SCATTER ? s_goff[kk & 255u] + i : tile_base + i;
        if (M::KV_OUT) a.okv[g] = v2u{kk, vv};
        else { a.ok[g] = kk; a.ov[g] = vv; }
    }
}

template <int MODE>
__device__ __forceinline__ void issue_tile(const Args &a, u32 t, v4u (&q)[8], u32 &goff_reg) {
    typedef ModeInfo<MODE> M;
    const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u32 tile_base = t * TILE;
    goff_reg = gld1(a.goff + t * 256 + (tid & 255));
    if (!M::KV_IN) {
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = gld4(a.sk + tile_base + w * 1024 + j * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < 4; j++) q[4 + j] = gld4(a.sv + tile_base + w * 1024 + j * 256 + lane * 4);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) q[j] = gld4(reinterpret_cast<const u32 *>(a.skv) + (u64)tile_base * 2 + w * 2048 + j * 256 + lane * 4);
    }
}

// one workgroup per tile
template <int MODE, bool XCD>
__global__ __launch_bounds__(NT) void k_tile(Args a) {
    __shared__ __attribute__((aligned(16))) u32 s_mem[2 * TILE];
    __shared__ u32 s_goff[256];
    const u32 t = tile_of_block(blockIdx.x, gridDim.x, XCD);
    v4u q[8];
    u32 goff_reg;
    issue_tile<MODE>(a, t, q, goff_reg);
    finish_tile<MODE, 0>(a, t, q, goff_reg, s_mem, s_goff);
}

// persistent workgroups: block j of XCD x takes tiles x * q + j, + per_xcd_blocks, ...; the loads of the next tile are
// issued before the current tile is stored, and waited for with a counted vmcnt (the stores stay in flight)
template <int MODE>
__global__ __launch_bounds__(NT, 4) void k_tile_pipe(Args a, u32 ntiles) {
    __shared__ __attribute__((aligned(16))) u32 s_mem[2 * TILE];
    __shared__ u32 s_goff[256];
    const u32 x = blockIdx.x % 8, j = blockIdx.x / 8, per = gridDim.x / 8, qn = ntiles / 8;
    v4u q0[8], q1[8];
    u32 g0, g1;
    u32 m = j;
    if (m >= qn) return;
    issue_tile<MODE>(a, x * qn + m, q0, g0);
    bool first = true;
    for (; m < qn; m += 2 * per) {
        // tile m in q0; prefetch m + per into q1
        const bool has1 = m + per < qn, has2 = m + 2 * per < qn;
        // wait for q0: behind it are the previous tile's stores (none the first time).  Counted waits need constants, so
        // the first tile waits for everything.
        if (first) { _Pragma("unroll") for (int z = 0; z < 8; z++) wait4<0>(q0[z]); wait1<0>(g0); first = false; }
        else { _Pragma("unroll") for (int z = 0; z < 8; z++) wait4<ModeInfo<MODE>::STORES>(q0[z]); wait1<ModeInfo<MODE>::STORES>(g0); }
        if (has1) issue_tile<MODE>(a, x * qn + m + per, q1, g1);
        finish_tile<MODE, 63>(a, x * qn + m, q0, g0, s_mem, s_goff);      // (63: its own waits are no-ops, done above)
        lds_barrier();
        if (!has1) break;
        _Pragma("unroll") for (int z = 0; z < 8; z++) wait4<ModeInfo<MODE>::STORES>(q1[z]);
        wait1<ModeInfo<MODE>::STORES>(g1);
        if (has2) issue_tile<MODE>(a, x * qn + m + 2 * per, q0, g0);
        finish_tile<MODE, 63>(a, x * qn + m + per, q1, g1, s_mem, s_goff);
        lds_barrier();
        if (!has2) break;
    }
}

// ---------------- C. flag hop latency ----------------
// A serial chain of workgroups: block b waits until block b - DIST has set its flag, then sets its own (relaxed agent-scope
// atomics: `sc1` accesses that are served at the memory side, the only inter-XCD-coherent level).  time / (blocks / DIST) = one
// store -> visible -> load-returns hop.  DIST = 1: neighbours on different XCDs; DIST = 8: the same XCD.  LOADED: every block
// first copies 64 KB, so that the chain runs under the traffic of a streaming kernel.  Spins are bounded (bad[1] counts give-ups).
template <int DIST, bool LOADED>
__global__ __launch_bounds__(256) void k_chain(u32 *flags, u32 epoch, const v4u *__restrict__ in, v4u *__restrict__ out, u32 *bad) {
    const u32 b = blockIdx.x;
    if (LOADED) {
        v4u r[16];
#pragma unroll
        for (int u = 0; u < 16; u++) r[u] = in[(u64)b * 4096 + u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 16; u++) out[(u64)b * 4096 + u * 256 + threadIdx.x] = r[u];
    }
    if (threadIdx.x == 0) {
        if (b >= DIST) {
            u32 spins = 0;
            while (__hip_atomic_load(&flags[b - DIST], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                if (++spins > (1u << 22)) { atomicAdd(bad + 1, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __hip_atomic_store(&flags[b], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------- host ----------------
struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <typename F> float run(F f, int warm, int reps) {
        for (int i = 0; i < warm; i++) f();
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; i++) f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        return ms / reps;
    }
};

static void report(const char *name, float ms) {
    const double bytes = (double)N * 16.0;
    printf("%-64s %8.4f ms  %7.1f GB/s  %.3f of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2;
    const int WARM = 10, REPS = 40;
    u32 *keys, *vals, *sk, *sv, *ok, *ov, *rk, *rv, *hist_tile, *goff, *bad;
    v2u *skv, *okv;
    CK(hipMalloc(&keys, (size_t)N * 4)); CK(hipMalloc(&vals, (size_t)N * 4));
    CK(hipMalloc(&sk, (size_t)N * 4)); CK(hipMalloc(&sv, (size_t)N * 4)); CK(hipMalloc(&skv, (size_t)N * 8));
    CK(hipMalloc(&ok, (size_t)N * 4)); CK(hipMalloc(&ov, (size_t)N * 4)); CK(hipMalloc(&okv, (size_t)N * 8));
    CK(hipMalloc(&rk, (size_t)N * 4)); CK(hipMalloc(&rv, (size_t)N * 4));
    CK(hipMalloc(&hist_tile, (size_t)NTILES * 256 * 4)); CK(hipMalloc(&goff, (size_t)NTILES * 256 * 4)); CK(hipMalloc(&bad, 8));
    k_gen<<<N / 256, 256>>>(keys, vals);
    k_tile_hist<<<NTILES, NT>>>(keys, hist_tile);
    CK(hipDeviceSynchronize());
    {
        std::vector<u32> h((size_t)NTILES * 256), g((size_t)NTILES * 256);
        CK(hipMemcpy(h.data(), hist_tile, h.size() * 4, hipMemcpyDeviceToHost));
        u64 acc = 0;
        std::vector<u32> off((size_t)NTILES * 256);                      // digit-major exclusive scan
        for (int d = 0; d < 256; d++)
            for (int t = 0; t < NTILES; t++) { off[(size_t)d * NTILES + t] = (u32)acc; acc += h[(size_t)t * 256 + d]; }
        for (int t = 0; t < NTILES; t++) {
            u32 ds = 0;
            for (int d = 0; d < 256; d++) { g[(size_t)t * 256 + d] = off[(size_t)d * NTILES + t] - ds; ds += h[(size_t)t * 256 + d]; }
        }
        CK(hipMemcpy(goff, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    }
    k_tile_sort<<<NTILES, NT>>>(keys, vals, hist_tile, sk, sv, skv);
    k_reference<<<NTILES, NT>>>(sk, sv, goff, rk, rv);
    CK(hipDeviceSynchronize());
    CK(hipFree(keys)); CK(hipFree(vals));

    Timer tm;
    Args a{sk, sv, skv, ok, ov, okv, goff, 0};
    auto check = [&](const char *name, bool kv, bool scattered) {
        CK(hipMemset(bad, 0, 4));
        const u32 *wk = scattered ? rk : sk, *wv = scattered ? rv : sv;
        if (kv) k_compare_kv<<<N / 256, 256>>>(okv, wk, wv, N, bad);
        else { k_compare<<<N / 256, 256>>>(ok, wk, N, bad); k_compare<<<N / 256, 256>>>(ov, wv, N, bad); }
        u32 hb;
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        if (hb) printf("   !! %s: %u mismatches\n", name, hb);
    };
    auto clear_out = [&]() { CK(hipMemset(ok, 0, (size_t)N * 4)); CK(hipMemset(ov, 0, (size_t)N * 4)); CK(hipMemset(okv, 0, (size_t)N * 8)); };

    for (int round = 0; round < rounds; round++) {
        printf("---- round %d ----\n", round);
        // A. plain copies: skv (512 MiB) -> okv
        const u64 nvec = (u64)N * 8 / 16;
        report("hipMemcpyAsync D2D 512 MiB", tm.run([&] { CK(hipMemcpyAsync(okv, skv, (size_t)N * 8, hipMemcpyDeviceToDevice, 0)); }, 3, 20));
#define COPY(U, NTL, NTS, GRID, LABEL) \
        report(LABEL, tm.run([&] { k_copy<U, NTL, NTS><<<dim3(GRID), dim3(256)>>>((const v4u *)skv, (v4u *)okv, nvec); }, WARM, REPS));
        COPY(1, false, false, (unsigned)(nvec / 256), "float4 copy, one-shot grid, 1 vector per thread");
        COPY(4, false, false, (unsigned)(nvec / 256 / 4), "float4 copy, one-shot grid, 4 vectors per thread");
        COPY(8, false, false, (unsigned)(nvec / 256 / 8), "float4 copy, one-shot grid, 8 vectors per thread");
        COPY(4, false, false, 256 * 8, "float4 copy, persistent 2048 blocks, 4 in flight");
        COPY(8, false, false, 256 * 8, "float4 copy, persistent 2048 blocks, 8 in flight");
        COPY(8, false, false, 256 * 4, "float4 copy, persistent 1024 blocks, 8 in flight");
        COPY(4, true, false, (unsigned)(nvec / 256 / 4), "float4 copy, one-shot, 4 per thread, nt loads");
        COPY(4, false, true, (unsigned)(nvec / 256 / 4), "float4 copy, one-shot, 4 per thread, nt stores");
        COPY(4, true, true, (unsigned)(nvec / 256 / 4), "float4 copy, one-shot, 4 per thread, nt loads + stores");
        COPY(8, true, true, 256 * 8, "float4 copy, persistent 2048 blocks, 8 in flight, nt both");

        // C. flag hops
        {
            u32 *flags;
            CK(hipMalloc(&flags, 65536 * 4));
            CK(hipMemset(flags, 0, 65536 * 4));
            CK(hipMemset(bad, 0, 8));
            u32 epoch = 1 + round * 100;
            const int NB = 8192;
#define CHAIN(DIST, LOADED, LABEL)                                                                                     \
            {                                                                                                          \
                float ms = tm.run([&] { k_chain<DIST, LOADED><<<NB, 256>>>(flags, epoch++, (const v4u *)skv, (v4u *)okv, bad); }, 2, 5); \
                printf("%-64s %8.4f ms  = %.3f us per hop (%d hops)\n", LABEL, ms, ms * 1e3 / (NB / DIST), NB / DIST);  \
            }
            CHAIN(1, false, "flag chain, block b waits for b-1 (other XCD), idle chip");
            CHAIN(8, false, "flag chain, block b waits for b-8 (same XCD), idle chip");
            CHAIN(1, true, "flag chain, b waits for b-1, every block copies 64 KB first");
            CHAIN(8, true, "flag chain, b waits for b-8, every block copies 64 KB first");
            u32 hb[2];
            CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
            if (hb[1]) printf("   !! flag chain: %u waits gave up\n", hb[1]);
            CK(hipFree(flags));
        }
        // B. tile kernels
#define TILEK(MODE, XCD, SPIN, LABEL)                                                           \
        {                                                                                       \
            a.spin = SPIN;                                                                      \
            if (round == 0) { clear_out(); k_tile<MODE, XCD><<<NTILES, NT>>>(a);                \
                              check(LABEL, ModeInfo<MODE>::KV_OUT, ModeInfo<MODE>::SCATTER); }  \
            report(LABEL, tm.run([&] { k_tile<MODE, XCD><<<NTILES, NT>>>(a); }, WARM, REPS));   \
        }
#define PIPEK(MODE, BLOCKS, SPIN, LABEL)                                                        \
        {                                                                                       \
            a.spin = SPIN;                                                                      \
            if (round == 0) { clear_out(); k_tile_pipe<MODE><<<BLOCKS, NT>>>(a, NTILES);        \
                              check(LABEL, ModeInfo<MODE>::KV_OUT, ModeInfo<MODE>::SCATTER); }  \
            report(LABEL, tm.run([&] { k_tile_pipe<MODE><<<BLOCKS, NT>>>(a, NTILES); }, WARM, REPS)); \
        }
        TILEK(0, true, 0, "tile 0 direct (no LDS), xcd");
        TILEK(1, true, 0, "tile 1 LDS, coalesced out, xcd");
        TILEK(1, false, 0, "tile 1 LDS, coalesced out, plain order");
        TILEK(2, true, 0, "tile 2 LDS, scattered out (separate arrays), xcd");
        TILEK(2, false, 0, "tile 2 LDS, scattered out (separate arrays), plain order");
        TILEK(2, true, 2000, "tile 2 scattered, xcd, spin 2000");
        TILEK(2, true, 6000, "tile 2 scattered, xcd, spin 6000");
        TILEK(3, true, 0, "tile 3 separate in, PAIRS out scattered, xcd");
        TILEK(3, true, 2000, "tile 3 separate in, PAIRS out scattered, xcd, spin 2000");
        TILEK(4, true, 0, "tile 4 pairs in, pairs out coalesced, xcd");
        TILEK(5, true, 0, "tile 5 pairs in, pairs out scattered, xcd");
        TILEK(5, true, 2000, "tile 5 pairs in, pairs out scattered, xcd, spin 2000");
        TILEK(6, true, 0, "tile 6 pairs in, separate out scattered, xcd");
        PIPEK(1, 512, 0, "pipe 1 coalesced, 512 persistent blocks");
        PIPEK(2, 512, 0, "pipe 2 scattered separate, 512 persistent blocks");
        PIPEK(2, 512, 2000, "pipe 2 scattered separate, 512 blocks, spin 2000");
        PIPEK(3, 512, 0, "pipe 3 separate in, pairs out scattered, 512 blocks");
        PIPEK(5, 512, 0, "pipe 5 pairs in, pairs out scattered, 512 blocks");
        PIPEK(5, 512, 2000, "pipe 5 pairs in/out scattered, 512 blocks, spin 2000");
        PIPEK(5, 1024, 0, "pipe 5 pairs in/out scattered, 1024 blocks (half resident)");
        PIPEK(6, 512, 0, "pipe 6 pairs in, separate out scattered, 512 blocks");
    }
    return 0;
}

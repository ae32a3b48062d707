// Microbenchmark (diagnostics, not product): do the scalar loads of ONE wave overlap?  Every wave follows K independent
// pointer chains through an array of 32-byte records (word 3 of a record = the next record), K s_load_dwordx8 per round
// issued back to back, one s_waitcnt per round.  Prints the time per round for K = 1..4 at 8 and 4 waves per SIMD and
// for an L2-resident and a 64 MB array (the record array of a 1 M-sphere scene).
//   hipcc --offload-arch=gfx950 -O2 -o smem_mlp tools/micro/smem_mlp.hip && ./smem_mlp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int K, bool VEC>
__global__ __launch_bounds__(1024) void k_chase(const char *__restrict__ base, uint32_t nrec, uint32_t rounds, uint32_t *out) {
    const uint32_t wave = (blockIdx.x * 1024 + threadIdx.x) / 64;
    uint32_t i0 = __builtin_amdgcn_readfirstlane((wave * 2654435761u) % nrec), i1 = __builtin_amdgcn_readfirstlane((wave * 40503u + 7) % nrec);
    uint32_t i2 = __builtin_amdgcn_readfirstlane((wave * 69069u + 13) % nrec), i3 = __builtin_amdgcn_readfirstlane((wave * 1664525u + 5) % nrec);
    if constexpr (VEC) {
        // the same chains through the vector memory path: lane k of the wave follows chain k
        uint32_t idx = threadIdx.x % 64 == 0 ? i0 : threadIdx.x % 64 == 1 ? i1 : threadIdx.x % 64 == 2 ? i2 : i3;
        const bool on = threadIdx.x % 64 < K;
        for (uint32_t r = 0; r < rounds; r++)
            if (on) idx = *reinterpret_cast<const uint32_t *>(base + 32ull * idx + 12);
        if (on && idx == 0xFFFFFFFFu) out[0] = idx;
        return;
    }
    for (uint32_t r = 0; r < rounds; r++) {
        asm volatile("s_lshl_b32 s40, %0, 5\n\t"
                     "s_load_dwordx8 s[40:47], %4, s40\n\t"
                     ".if %5 > 1\n\t"
                     "s_lshl_b32 s48, %1, 5\n\t"
                     "s_load_dwordx8 s[48:55], %4, s48\n\t"
                     ".endif\n\t"
                     ".if %5 > 2\n\t"
                     "s_lshl_b32 s56, %2, 5\n\t"
                     "s_load_dwordx8 s[56:63], %4, s56\n\t"
                     ".endif\n\t"
                     ".if %5 > 3\n\t"
                     "s_lshl_b32 s64, %3, 5\n\t"
                     "s_load_dwordx8 s[64:71], %4, s64\n\t"
                     ".endif\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_mov_b32 %0, s43\n\t"
                     ".if %5 > 1\n\ts_mov_b32 %1, s51\n\t.endif\n\t"
                     ".if %5 > 2\n\ts_mov_b32 %2, s59\n\t.endif\n\t"
                     ".if %5 > 3\n\ts_mov_b32 %3, s67\n\t.endif"
                     : "+s"(i0), "+s"(i1), "+s"(i2), "+s"(i3)
                     : "s"(base), "n"(K)
                     : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
                       "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "scc");
    }
    if ((i0 ^ i1 ^ i2 ^ i3) == 0xFFFFFFFFu) out[0] = i0;
}

template <int K, bool VEC>
static float run(const char *d, uint32_t nrec, int blocks, uint32_t rounds, uint32_t *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_chase<K, VEC><<<blocks, 1024>>>(d, nrec, rounds, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_chase<K, VEC><<<blocks, 1024>>>(d, nrec, rounds, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const uint32_t rounds = 2000;
    for (uint32_t nrec : {32768u, 2097152u, 33554432u}) {
        std::vector<uint32_t> h((size_t)nrec * 8);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 20) % nrec; }
        char *d; uint32_t *out;
        hipMalloc((void **)&d, h.size() * 4); hipMalloc((void **)&out, 4);
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int blocks : {512, 256, 64}) {
            const float t1 = run<1, false>(d, nrec, blocks, rounds, out), t2 = run<2, false>(d, nrec, blocks, rounds, out);
            const float t3 = run<3, false>(d, nrec, blocks, rounds, out), t4 = run<4, false>(d, nrec, blocks, rounds, out);
            const float v1 = run<1, true>(d, nrec, blocks, rounds, out), v4 = run<4, true>(d, nrec, blocks, rounds, out);
            printf("records %9u (%6.1f MB) blocks %3d (%d waves/SIMD on %s): ns per round, scalar K=1..4: %7.1f %7.1f %7.1f %7.1f | vector K=1, 4: %7.1f %7.1f"
                   " | scalar fetches per us per CU at K=1, 4: %.2f %.2f\n",
                   nrec, nrec * 32 / 1e6, blocks, blocks >= 512 ? 8 : 4, blocks >= 256 ? "256 CUs" : "64 CUs", t1 * 1e6 / rounds, t2 * 1e6 / rounds, t3 * 1e6 / rounds,
                   t4 * 1e6 / rounds, v1 * 1e6 / rounds, v4 * 1e6 / rounds,
                   16.0 * (blocks > 256 ? 2 : 1) * rounds / (t1 * 1e3), 4 * 16.0 * (blocks > 256 ? 2 : 1) * rounds / (t4 * 1e3));
        }
        hipFree(d); hipFree(out);
    }
    return 0;
}

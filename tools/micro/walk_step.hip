// Microbenchmark (diagnostics, not product): what does ONE step of the packet walk (csrc/bvh.hip phase 2) cost when
// every wave of the chip is walking?  Every wave follows pointer chains through an array of 32-byte records (word 3 = the
// next record) and runs, per record, the walk's own instruction sequence for "nobody overlaps":
//   MODE 0: fetch only (s_lshl, s_load_dwordx8, s_waitcnt, s_cmp, s_cbranch)
//   MODE 1: + the six v_cmpx against the lanes' boxes (the first one already fails for every lane), s_cbranch_execnz, EXEC restored
//   MODE 2: as 1, but every lane passes all six compares (EXEC stays full)
//   MODE 3: three chains per wave in lock step with the bookkeeping of the three-context walk (WALK = 3), MODE 1 per record
//   MODE 4: as 1 with only TWO v_cmpx per record
// Prints ns per record-step per wave, i.e. the walk's cost of a step at full occupancy.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o walk_step tools/micro/walk_step.hip && ./walk_step
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CMPX6(LX, LY, LZ, HX, HY, HZ)                \
    "v_cmpx_lt_f32_e32 vcc, " LX ", %[hx]\n\t"       \
    "v_cmpx_gt_f32_e32 vcc, " HX ", %[lx]\n\t"       \
    "v_cmpx_lt_f32_e32 vcc, " LY ", %[hx]\n\t"       \
    "v_cmpx_gt_f32_e32 vcc, " HY ", %[lx]\n\t"       \
    "v_cmpx_lt_f32_e32 vcc, " LZ ", %[hx]\n\t"       \
    "v_cmpx_gt_f32_e32 vcc, " HZ ", %[lx]\n\t"

template <int MODE>
__global__ __launch_bounds__(1024) void k_step(const char *__restrict__ base, uint32_t nrec, uint32_t rounds, uint32_t *out) {
    const uint32_t wave = (blockIdx.x * 1024 + threadIdx.x) / 64;
    uint32_t i0 = __builtin_amdgcn_readfirstlane((wave * 2654435761u) % nrec), i1 = __builtin_amdgcn_readfirstlane((wave * 40503u + 7) % nrec);
    uint32_t i2 = __builtin_amdgcn_readfirstlane((wave * 69069u + 13) % nrec);
    // record words are indices < 2^25, i.e. tiny positive floats as bit patterns: hx = -1 fails "lo < hx" for everybody,
    // hx = +1 / lx = -1 passes every compare
    float hx = MODE == 2 ? 1.0f : -1.0f, lx = -1.0f;
    asm volatile("" : "+v"(hx), "+v"(lx));
    uint64_t exec0;
    uint32_t r = rounds, t0;
    if constexpr (MODE == 3) {
        uint32_t e0 = 0xFFFFFFFFu, e1 = e0, e2 = e0;
        asm volatile("s_mov_b64 %[exec0], exec\n"
                     "60:\n\t"
#define ISSUE(K, C, R)                                           \
                     "s_cmp_eq_u32 " C ", 0x7fffffff\n\t"        \
                     "s_cbranch_scc1 19" K "f\n\t"               \
                     "s_lshl_b32 %[t0], " C ", 5\n\t"            \
                     "s_load_dwordx8 " R ", %[base], %[t0]\n"    \
                     "19" K ":\n\t"
                     ISSUE("0", "%[c0]", "s[40:47]") ISSUE("1", "%[c1]", "s[48:55]") ISSUE("2", "%[c2]", "s[56:63]")
                     "s_waitcnt lgkmcnt(0)\n\t"
#define CTX(K, C, E, LX, LY, LZ, SKIP, HX, HY, HZ)               \
                     "s_cmp_eq_u32 " SKIP ", 0x7ffffffe\n\t"     \
                     "s_cbranch_scc1 13" K "f\n\t"               \
                     CMPX6(LX, LY, LZ, HX, HY, HZ)                 \
                     "s_cbranch_execnz 13" K "f\n\t"             \
                     "s_mov_b64 exec, %[exec0]\n\t"              \
                     "s_cmp_eq_u32 " SKIP ", " E "\n\t"          \
                     "s_cbranch_scc1 13" K "f\n\t"               \
                     "s_mov_b32 " C ", " SKIP "\n"               \
                     "13" K ":\n\t"
                     CTX("0", "%[c0]", "%[e0]", "s40", "s41", "s42", "s43", "s44", "s45", "s46")
                     CTX("1", "%[c1]", "%[e1]", "s48", "s49", "s50", "s51", "s52", "s53", "s54")
                     CTX("2", "%[c2]", "%[e2]", "s56", "s57", "s58", "s59", "s60", "s61", "s62")
                     "s_cmp_lg_u32 %[nb], 0\n\t"
                     "s_cbranch_scc1 61f\n\t"
                     "s_sub_u32 %[r], %[r], 1\n\t"
                     "s_cmp_lg_u32 %[r], 0\n\t"
                     "s_cbranch_scc1 60b\n"
                     "61:\n\t"
                     "s_mov_b64 exec, %[exec0]"
                     : [c0] "+s"(i0), [c1] "+s"(i1), [c2] "+s"(i2), [e0] "+s"(e0), [e1] "+s"(e1), [e2] "+s"(e2), [r] "+s"(r),
                       [exec0] "=&s"(exec0), [t0] "=&s"(t0)
                     : [base] "s"(base), [hx] "v"(hx), [lx] "v"(lx), [nb] "s"(0u)
                     : "vcc", "scc", "memory", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53",
                       "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63");
    } else {
        asm volatile("s_mov_b64 %[exec0], exec\n\t"
                     "s_mov_b32 s43, %[c0]\n"
                     "1:\n\t"
                     "s_lshl_b32 %[t0], s43, 5\n\t"
                     "s_load_dwordx8 s[40:47], %[base], %[t0]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     ".if %[mode] == 1 || %[mode] == 2\n\t"
                     CMPX6("s40", "s41", "s42", "s44", "s45", "s46")
                     ".endif\n\t"
                     ".if %[mode] == 4\n\t"
                     "v_cmpx_lt_f32_e32 vcc, s40, %[hx]\n\t"
                     "v_cmpx_gt_f32_e32 vcc, s44, %[lx]\n\t"
                     ".endif\n\t"
                     ".if %[mode] == 1 || %[mode] == 4\n\t"
                     "s_cbranch_execnz 2f\n\t"
                     ".endif\n\t"
                     ".if %[mode] != 0\n\t"
                     "s_mov_b64 exec, %[exec0]\n\t"
                     ".endif\n\t"
                     "s_sub_u32 %[r], %[r], 1\n\t"
                     "s_cmp_lg_u32 %[r], 0\n\t"
                     "s_cbranch_scc1 1b\n"
                     "2:\n\t"
                     "s_mov_b64 exec, %[exec0]\n\t"
                     "s_mov_b32 %[c0], s43"
                     : [c0] "+s"(i0), [r] "+s"(r), [exec0] "=&s"(exec0), [t0] "=&s"(t0)
                     : [base] "s"(base), [hx] "v"(hx), [lx] "v"(lx), [mode] "n"(MODE)
                     : "vcc", "scc", "memory", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    }
    if ((i0 ^ i1 ^ i2) == 0xFFFFFFFFu) out[0] = i0 + r;
}

template <int MODE>
static float run(const char *d, uint32_t nrec, int blocks, uint32_t rounds, uint32_t *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_step<MODE><<<blocks, 1024>>>(d, nrec, rounds, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_step<MODE><<<blocks, 1024>>>(d, nrec, rounds, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e6f / rounds;
}

int main() {
    const uint32_t rounds = 3000;
    for (uint32_t nrec : {256u, 32768u, 2097152u}) {
        std::vector<uint32_t> h((size_t)nrec * 8);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 20) % nrec; }
        char *d; uint32_t *out;
        hipMalloc((void **)&d, h.size() * 4); hipMalloc((void **)&out, 4);
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int blocks : {512, 256}) {
            const float m0 = run<0>(d, nrec, blocks, rounds, out), m1 = run<1>(d, nrec, blocks, rounds, out), m2 = run<2>(d, nrec, blocks, rounds, out);
            const float m3 = run<3>(d, nrec, blocks, rounds, out), m4 = run<4>(d, nrec, blocks, rounds, out);
            printf("records %8u (%6.2f MB), %d waves/SIMD: ns per record per wave: fetch only %6.1f | + 6 v_cmpx (all fail at once) %6.1f | + 6 v_cmpx (all pass) %6.1f |"
                   " + 2 v_cmpx %6.1f | three chains in lock step, per ROUND %6.1f = per record %6.1f\n",
                   nrec, nrec * 32 / 1e6, blocks >= 512 ? 8 : 4, m0, m1, m2, m4, m3, m3 / 3);
        }
        hipFree(d); hipFree(out);
    }
    return 0;
}

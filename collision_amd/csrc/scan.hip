// Exclusive prefix sum of uint32 (wraps mod 2^32), in place.
// Replaces PrefixScanner.prefix_sum (collision/scan.py:75-112) and its kernels
// local_scan / block_scan / up_sweep / down_sweep (collision/scan.cl:5-36,
// collision/local_scan.cl:2-25).
//
// MI355X shape: reduce-then-scan instead of the reference's scan-then-propagate, so the
// array is read twice and written once (12 B/element instead of 16).  A tile is 2048
// elements per 256-thread block, 8 per thread as two 16-byte loads; inside a block the
// per-thread totals are scanned with wave shuffles and a 4-entry LDS hand-off between
// the block's waves (no Blelloch sweeps, no log2(n) barriers).
#include "col_common.h"
#include <atomic>

namespace {

constexpr int ST = 256;              // threads per block
constexpr int SI = 8;                // items per thread
constexpr int STILE = ST * SI;       // 2048

__device__ __forceinline__ void load_items(const u32 *data, uint64_t base, uint64_t n, u32 (&v)[SI]) {
    // thread-blocked: 8 consecutive elements per thread, two uint4 loads when fully in range
    if (base + SI <= n) {
        const uint4 a = *reinterpret_cast<const uint4 *>(data + base);
        const uint4 b = *reinterpret_cast<const uint4 *>(data + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < SI; k++) v[k] = (base + k < n) ? data[base + k] : 0u;
    }
}

__global__ __launch_bounds__(ST) void k_scan_reduce(const u32 *__restrict__ data, uint64_t n, u32 *__restrict__ sums) {
    __shared__ u32 ws[ST / COL_WAVE];
    const uint64_t base = (uint64_t)blockIdx.x * STILE + (uint64_t)threadIdx.x * SI;
    u32 v[SI];
    load_items(data, base, n, v);
    u32 t = 0;
#pragma unroll
    for (int k = 0; k < SI; k++) t += v[k];
    t = wave_sum(t);
    if (lane_id() == 0) ws[threadIdx.x / COL_WAVE] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 s = 0;
#pragma unroll
        for (int i = 0; i < ST / COL_WAVE; i++) s += ws[i];
        sums[blockIdx.x] = s;
    }
}

// Exclusive scan of each tile, plus bases[blockIdx] when bases != NULL.
__global__ __launch_bounds__(ST) void k_scan_apply(u32 *__restrict__ data, uint64_t n, const u32 *__restrict__ bases) {
    __shared__ u32 ws[ST / COL_WAVE];
    const uint64_t base = (uint64_t)blockIdx.x * STILE + (uint64_t)threadIdx.x * SI;
    u32 v[SI];
    load_items(data, base, n, v);
    u32 t = 0;
#pragma unroll
    for (int k = 0; k < SI; k++) t += v[k];
    u32 total;
    u32 run = block_excl_scan<ST>(t, ws, &total);
    if (bases) run += bases[blockIdx.x];
    u32 o[SI];
#pragma unroll
    for (int k = 0; k < SI; k++) { o[k] = run; run += v[k]; }
    if (base + SI <= n) {
        *reinterpret_cast<uint4 *>(data + base) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4 *>(data + base + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < SI; k++)
            if (base + k < n) data[base + k] = o[k];
    }
}

// Small arrays (the radix histogram of a ~1 M-element sort is 60 k counters = 31 tiles): ONE launch.
// Every block publishes its tile sum as an 8-byte {epoch, sum} granule (one relaxed agent-scope
// atomic store: the data is the flag, cdna_hip_programming.md guideline 16 R2), then wave 0 reads
// the granules of all predecessor tiles in parallel (lane = predecessor) and adds them up -- a
// look-back without a chain, so the depth is two hops whatever the tile count.  Forward progress: a
// block only waits for blocks with a LOWER index, and the dispatcher starts workgroups in index order,
// so whichever blocks are resident (other streams or processes may hold part of the GPU) always include
// one that waits for nothing -- the same assumption every decoupled look-back scan makes.  The epoch
// makes stale granules from earlier calls invisible without a memset launch.
// Round 4: up to 1024 tiles (the 2 Mi counters of a 64 Mi-pair sort pass: three launches -> one, ~9 us per pass): every thread
// polls up to CHAIN_PER_THREAD predecessors (tiles t, t + 256, ...), all of its loads in flight together.
constexpr u32 CHAIN_PER_THREAD = 4;
constexpr u32 CHAIN_MAX_TILES = ST * CHAIN_PER_THREAD;      // 1024
__global__ __launch_bounds__(ST) void k_scan_chain(u32 *__restrict__ data, u32 n, u64 *__restrict__ status, u32 epoch) {
    __shared__ u32 ws[ST / COL_WAVE];
    __shared__ u32 s_prefix;
    const u32 b = blockIdx.x;
    const uint64_t base = (uint64_t)b * STILE + (uint64_t)threadIdx.x * SI;
    u32 v[SI];
    load_items(data, base, n, v);
    u32 t = 0;
#pragma unroll
    for (int k = 0; k < SI; k++) t += v[k];
    u32 total;
    u32 run = block_excl_scan<ST>(t, ws, &total);
    if (threadIdx.x == 0)
        __hip_atomic_store(&status[b], ((u64)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {   // thread t polls predecessor tiles t, t + ST, ... (< b); ws is free again after the scan
        u32 pv = 0;
        u64 g[CHAIN_PER_THREAD];
#pragma unroll
        for (u32 k = 0; k < CHAIN_PER_THREAD; k++) {       // first look: all of the thread's loads in flight together
            const u32 t = threadIdx.x + k * ST;
            g[k] = t < b ? __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((u64)epoch << 32);
        }
#pragma unroll
        for (u32 k = 0; k < CHAIN_PER_THREAD; k++) {       // (b <= ST: only k = 0 ever polls -- the small scans of the 1 M path)
            const u32 t = threadIdx.x + k * ST;
            while ((u32)(g[k] >> 32) != epoch) g[k] = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pv += (u32)g[k];
        }
        pv = wave_sum(pv);
        if (lane_id() == 0) ws[threadIdx.x / COL_WAVE] = pv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 p = 0;
#pragma unroll
        for (int i = 0; i < ST / COL_WAVE; i++) p += ws[i];
        s_prefix = p;
    }
    __syncthreads();
    run += s_prefix;
    u32 o[SI];
#pragma unroll
    for (int k = 0; k < SI; k++) { o[k] = run; run += v[k]; }
    if (base + SI <= n) {
        *reinterpret_cast<uint4 *>(data + base) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4 *>(data + base + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < SI; k++)
            if (base + k < n) data[base + k] = o[k];
    }
}

// --- the reference's two kernels with its group structure (kernel-level parity only) ---
// scan.cl:5-30: one 64-lane wave per group of `block` elements, chunked with a carry.
__global__ __launch_bounds__(COL_WAVE) void k_ref_local_scan(u32 *data, u32 block, u32 *block_sums) {
    const uint64_t g0 = (uint64_t)blockIdx.x * block;
    u32 carry = 0;
    for (u32 c = 0; c < block; c += COL_WAVE) {
        const u32 i = c + threadIdx.x;
        const u32 v = i < block ? data[g0 + i] : 0u;
        const u32 incl = wave_incl_scan(v);
        if (i < block) data[g0 + i] = carry + incl - v;
        carry += __shfl(incl, COL_WAVE - 1, COL_WAVE);
    }
    if (block_sums && threadIdx.x == 0) block_sums[blockIdx.x] = carry;
}

// scan.cl:32-36
__global__ __launch_bounds__(256) void k_ref_block_scan(u32 *data, uint64_t n, u32 block, const u32 *block_sums) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) data[i] += block_sums[i / block];
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int scan_rec(hipStream_t s, u32 *data, uint64_t n, char *scratch) {
    if (n == 0) return COL_OK;
    const uint64_t nb = col_ceil_div(n, STILE);
    // Under stream capture the epoch argument would be frozen into the graph and every replay would
    // see the previous replay's granules as valid, so a captured scan takes the three-launch path.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (nb > 1 && nb <= CHAIN_MAX_TILES) (void)hipStreamIsCapturing(s, &capture);
    if (nb > 1 && nb <= CHAIN_MAX_TILES && capture == hipStreamCaptureStatusNone) {
        // never reused within a process (see k_scan_chain); atomic: host threads may drive different streams
        static std::atomic<u32> epoch_counter{0x5EED0000u};
        const u32 epoch = epoch_counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
        k_scan_chain<<<dim3((unsigned)nb), dim3(ST), 0, s>>>(data, (u32)n, (u64 *)scratch, epoch);
        COL_LAUNCH_OK();
        return COL_OK;
    }
    if (nb == 1) {
        k_scan_apply<<<dim3(1), dim3(ST), 0, s>>>(data, n, nullptr);
        COL_LAUNCH_OK();
        return COL_OK;
    }
    u32 *sums = (u32 *)scratch;
    k_scan_reduce<<<dim3((unsigned)nb), dim3(ST), 0, s>>>(data, n, sums);
    COL_LAUNCH_OK();
    int rc = scan_rec(s, sums, nb, scratch + align256(nb * sizeof(u32)));
    if (rc) return rc;
    k_scan_apply<<<dim3((unsigned)nb), dim3(ST), 0, s>>>(data, n, sums);
    COL_LAUNCH_OK();
    return COL_OK;
}

}  // namespace

extern "C" {

size_t col_scan_scratch_bytes(uint64_t n) {
    size_t total = 256 + CHAIN_MAX_TILES * sizeof(u64);
    while (n > (uint64_t)STILE) {
        n = col_ceil_div(n, STILE);
        total += align256(n * sizeof(u32));
    }
    return total;
}

int col_scan_u32(void *stream, uint32_t *data, uint64_t n, void *scratch) {
    if (n > (uint64_t)STILE && !scratch) return COL_ENOSCRATCH;
    if (n >= ((uint64_t)1 << 42)) return COL_EINVAL;
    return scan_rec(col_stream(stream), data, n, (char *)scratch);
}

int col_local_scan(void *stream, uint32_t *data, uint64_t n, uint32_t block, uint32_t *block_sums) {
    if (block == 0 || n % block) return COL_EINVAL;
    if (n == 0) return COL_OK;
    k_ref_local_scan<<<dim3((unsigned)(n / block)), dim3(COL_WAVE), 0, col_stream(stream)>>>(data, block, block_sums);
    COL_LAUNCH_OK();
    return COL_OK;
}

int col_block_scan(void *stream, uint32_t *data, uint64_t n, uint32_t block, const uint32_t *block_sums) {
    if (block == 0) return COL_EINVAL;
    if (n == 0) return COL_OK;
    k_ref_block_scan<<<dim3((unsigned)col_ceil_div(n, 256)), dim3(256), 0, col_stream(stream)>>>(data, n, block, block_sums);
    COL_LAUNCH_OK();
    return COL_OK;
}

}  // extern "C"

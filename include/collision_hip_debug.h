/*
 * collision_hip_debug.h -- diagnostics entry points of libcollision_hip.so.  NOT part of the drop-in ABI
 * (include/collision_hip.h): nothing a maintainer of the reference binds.  They exist for the
 * A/B tools under tools/, for bench.py's ablation legs and for a few tests that force a code path at a size
 * where the library would not choose it.  The col_debug_* switches are PROCESS-WIDE and unsynchronised: they
 * select separate diagnostics instances of the kernels for every caller in the process (the production
 * instances carry no diagnostics code).  Set them from one thread while no work is in flight;
 * col_collide / col_collide_plan refuse to run under a forced tile class.
 */
#ifndef COLLISION_HIP_DEBUG_H
#define COLLISION_HIP_DEBUG_H

#include "collision_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

int col_debug_xcc_census(void *stream, uint32_t *out, uint32_t nblocks);   /* diagnostics: XCC id per workgroup */
int col_debug_walk_profile(uint32_t *out, uint32_t npackets);   /* diagnostics: see csrc/bvh.hip */
void col_debug_traverse(int variant);   /* diagnostics: 0 packet walk (default), 1 lane-per-query, bit 1 vector record loads, bit 2 half grid;
                                         * the other bits are listed where they act (csrc/bvh.hip launch_traverse): 12 profiling instance,
                                         * 13 / 14 / 15 dynamic packet order forced / forbidden / at every size, 16 / 17 split units, 21 no walk
                                         * order, 22 the first batch drawn from the counter too, 23 long walks at ordinary priority,
                                         * 24 / 25 the walk with the lower / upper half of a packet's queries only (timing experiment:
                                         * the pair set is then a subset) */
void col_debug_lbvh(int mode);          /* diagnostics: timing ablations of k_chunk, 0 = off; 1024 alone = the production instance with the
                                         * round-3 code (shuffle scans, branchy Karras probes) for A/Bs, not a diagnostics instance; likewise
                                         * 2048 (float64 with shuffle scans), 8192 (Karras' searches for every node instead of the climb),
                                         * 16384 (24 KB of unused LDS per workgroup: half the resident workgroups); 4096: the traversal's
                                         * longest-first walk order whatever the scene (tests) */
void col_debug_leaf_blocks(float k);    /* leaf-block criterion of the fused LBVH build, process-wide: a node of <= 16 leaves is marked when it is at
                                         * most k leaf boxes wide on every axis; default 3, 0 = no marks, a huge k = every small node */
void col_debug_radix(int mode);         /* diagnostics: 2 = coalesced output, 4 = blockIdx tile order, 8 = dword loads, 32 = phase stamps,
                                           64 = non-temporal loads, 128/256 = system/agent-scope stores, 512/1024 = fewer blocks per CU,
                                           32768 = every store lands in a 4 MiB window, 65536 = ranking skipped on a tile-sorted input,
                                           1 << 21 = ranks from returning LDS atomics (experiment, csrc/radix.hip),
                                           1 << 22 | log2(G) << 24 = strips of G tiles go round the XCDs instead of one tile range per XCD */
int col_debug_radix_tile(int tile);     /* diagnostics: force the tile class (1024, 4096, 8192, 16384; 0 = automatic).  Set it BEFORE sizing
                                           scratch with col_radix_scratch_bytes / col_radix_tile: the histogram layout follows it.
                                           8 << 20 / 16 << 20: where the 8192-pair tile takes over from the 4096-pair one (default 8 Mi pairs;
                                           A/B material, not a forced class -- buffers sized before the switch do not follow it);
                                           8193 / 8194: the 8192-pair tile as 1024 x 8 / 512 x 16 (default) */
int col_debug_radix_stamps(uint64_t *out8, int reset);
/* copies `bytes` (a multiple of 128 KiB) from `in` to `out`: shape 0 = float4 copy, one vector per thread; shape 1 = the scatter
 * pass's tile shape (64 KB per 512-thread workgroup through LDS, coalesced stores) without any ranking: bench.py's
 * roofline.copy_ceiling, measured beside the pass */
int col_debug_copy(void *stream, const void *in, void *out, uint64_t bytes, int shape);   /* diagnostics: cycles per k_scatter phase, summed over blocks */
/* Diagnostics build of the traversal -- same pairs, same counter semantics -- that also counts its work.
 * stats: 8 x uint64, zeroed by the caller (steps, descents, leaf tests, leaf hits, steps within 1k/2k/4k/8k positions
 * of the block start) */
int col_traverse_stats(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity,
                       const void *bounds, uint32_t n, int coord_bytes, uint64_t *stats, int mode);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* COLLISION_HIP_DEBUG_H */

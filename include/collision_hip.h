/*
 * collision_hip.h -- C ABI of libcollision_hip.so, the MI355X (gfx950) engine
 * behind the kwohlfahrt/collision hot path.
 *
 * The reference has no FFI: its boundary is the Python class API over PyOpenCL
 * (SURVEY.md section 8b).  Each entry point below replaces one PyOpenCL kernel
 * enqueue (or a fixed group of them) of the reference; the reference interface
 * it stands in for is cited as file:line relative to /root/reference.  A
 * maintainer of the reference binds these with ctypes exactly as
 * collision_amd/_lib.py does.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - Every function returns 0 on success, a positive hipError_t value when the
 *     HIP runtime failed, or a negative COL_E* code; col_error_string() names it.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Every compute call is asynchronous on that stream, allocates nothing and
 *     never synchronises; scratch comes from the caller (col_*_scratch_bytes).
 *   - Device pointers are void* into hipMalloc'd (or torch-owned) HBM.
 *   - coord_bytes is 4 (float) or 8 (double); a "vec3" row is 4 scalars wide
 *     (collision/misc.py:62-71), lane w of *input* rows is ignored.
 */
#ifndef COLLISION_HIP_H
#define COLLISION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libcollision_hip.so is built with -fvisibility=hidden: exactly the functions declared in this header are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define COL_OK 0
#define COL_EINVAL (-1)      /* bad argument (size/dtype combination not supported) */
#define COL_ENOSCRATCH (-2)  /* scratch pointer missing */

/* element type codes for the generic reducer / fill */
#define COL_F32 0
#define COL_F64 1
#define COL_U32 2
#define COL_I32 3
#define COL_U64 4
#define COL_I64 5

/* reduction operators (collision/bounds.py:5, collision/summer.py:5) */
#define COL_OP_MINMAX 0   /* out = [min row, max row] */
#define COL_OP_SUM 1      /* out = [sum row] */

/* Node record, collision/collision.py:9 and collision/collision.cl:42-53. */
typedef struct col_node {
    uint32_t parent;      /* never written for the root */
    uint32_t right_edge;  /* last sorted leaf position covered */
    uint32_t data[2];     /* leaf: data[0] = sphere id; internal: children */
} col_node;

#define COL_NO_NODE 0xFFFFFFFFu   /* collision/collision.py:11 */

/* ---------------------------------------------------------------- runtime
 * Stand-ins for the PyOpenCL objects the reference's callers create
 * (tests/conftest.py:4-12: Context/CommandQueue; cl.Buffer, cl.enqueue_copy,
 * cl.enqueue_fill_buffer, cl.Event, cl.wait_for_events). */
const char *col_error_string(int code);
int col_version(void);
int col_device_count(int *count);
int col_set_device(int device);
int col_get_device(int *device);
int col_device_name(char *buf, int len);
int col_device_sync(void);
int col_malloc(void **ptr, size_t bytes);
int col_free(void *ptr);
int col_host_alloc(void **ptr, size_t bytes);   /* pinned host memory */
int col_host_free(void *ptr);
int col_memcpy_h2d(void *stream, void *dst, const void *src, size_t bytes);
int col_memcpy_d2h(void *stream, void *dst, const void *src, size_t bytes);
int col_memcpy_d2d(void *stream, void *dst, const void *src, size_t bytes);
/* cl.enqueue_fill_buffer with a 1/2/4/8/16-byte pattern */
int col_fill(void *stream, void *dst, const void *pattern, size_t pattern_bytes, size_t count);
int col_stream_create(void **stream);
int col_stream_destroy(void *stream);
int col_stream_sync(void *stream);
int col_stream_wait_event(void *stream, void *event);
int col_event_create(void **event);
int col_event_destroy(void *event);
int col_event_record(void *event, void *stream);
int col_event_sync(void *event);
int col_event_elapsed_ms(float *ms, void *start, void *stop);

/* ---------------------------------------------------------------- reduce
 * Replaces Reducer.reduce = bounds1 + bounds2 (collision/reduce.py:62-76,
 * collision/reduce.cl:5-58).  values: n rows of `width` scalars of type
 * `dtype`; out: width mins then width maxes (COL_OP_MINMAX) or width sums. */
size_t col_reduce_scratch_bytes(int dtype, int width);
int col_reduce(void *stream, const void *values, uint64_t n, int dtype, int width, int op,
               void *scratch, void *out);
/* Any accumulator list (collision/reduce.py:9-22 renders a list of (init, fn) pairs into reduce.cl): n_acc <=
 * COL_REDUCE_MAX_ACC accumulators, ops[k] one of COL_ACC_*, inits[k] the initial value (+-INFINITY = the
 * type's extreme).  int_inits (may be NULL): the same initial values as 64-bit integers, used for the integer
 * dtypes where a double is inexact beyond 2^53 (u64 values as their two's-complement bit pattern; an entry whose
 * inits[k] is +-INFINITY is ignored).  out: n_acc rows of `width` scalars, in list order (reduce.cl:34-37). */
#define COL_REDUCE_MAX_ACC 4
#define COL_ACC_MIN 0
#define COL_ACC_MAX 1
#define COL_ACC_ADD 2
#define COL_ACC_MUL 3
int col_reduce_list(void *stream, const void *values, uint64_t n, int dtype, int width, int n_acc, const int *ops,
                    const double *inits, const int64_t *int_inits, void *scratch, void *out);

/* ANY accumulator list, as the reference's template takes it (collision/reduce.py:9-22: every (initial value, binary
 * function name) pair is rendered into reduce.cl and compiled at run time): `source` is a HIP rendering of
 * reduce.cl:5-58 with two kernels, bounds1(values, n, group_accs) and bounds2(group_accs, output) -- the host side
 * (collision_amd/reduce.py) writes it -- compiled with hiprtc (loaded on first use) for the current device.
 * col_reduce_rtc launches it as reduce.py:62-76 does: bounds1 on ngroups x group_size work-items with acc_bytes of LDS
 * per work-item, bounds2 on one group of ngroups work-items; so a function that is not associative meets its operands
 * in the reference's order.  ngroups, group_size <= 1024 and acc_bytes * max(ngroups, group_size) <= 65536.
 * partials: ngroups * acc_bytes bytes.  out: acc_bytes bytes, one row per accumulator (reduce.cl:55-58).
 * log (may be NULL): the compiler's messages.  col_reduce_rtc_check compiles only (no device): for tests. */
int col_reduce_rtc_check(const char *source, const char *arch, char *log, size_t log_cap);
int col_reduce_rtc_create(const char *source, char *log, size_t log_cap, void **handle);
int col_reduce_rtc_destroy(void *handle);
int col_reduce_rtc(void *stream, void *handle, const void *values, uint64_t n, uint32_t ngroups, uint32_t group_size,
                   uint32_t acc_bytes, void *partials, void *out);

/* ---------------------------------------------------------------- morton
 * Replaces the `range` kernel, the padding fill and `calculateCodes`
 * (collision/collision.py:137-146,161-165; collision/collision.cl:8-40).
 * codes[i] = morton(coords[i]) for i < n, 0xFFFFFFFF for n <= i < padded;
 * ids[i] = i for i < padded (ids may be NULL).  range = 2 rows (min, max). */
int col_morton(void *stream, const void *coords, const void *range, uint32_t n, uint32_t padded,
               int coord_bytes, uint32_t *codes, uint32_t *ids);

/* ---------------------------------------------------------------- scan
 * Replaces PrefixScanner.prefix_sum = local_scan/block_scan levels
 * (collision/scan.py:75-112, collision/scan.cl:5-36): in-place exclusive
 * scan of n uint32 (sums wrap mod 2^32). */
size_t col_scan_scratch_bytes(uint64_t n);
int col_scan_u32(void *stream, uint32_t *data, uint64_t n, void *scratch);
/* The two reference kernels on their own, for the kernel-level parity tests
 * (tests/test_scan.py:24-103): block = 2*group_size elements per group. */
int col_local_scan(void *stream, uint32_t *data, uint64_t n, uint32_t block, uint32_t *block_sums);
int col_block_scan(void *stream, uint32_t *data, uint64_t n, uint32_t block, const uint32_t *block_sums);

/* ---------------------------------------------------------------- radix sort
 * Replaces RadixSorter.sort (collision/radix.py:118-170): stable LSD sort of
 * n keys (key_bytes 4|8) with optional values (val_bytes 0|4|8|16|32 moved by
 * the scatter passes; 1|2|64|128 sorted as (key, index) and gathered once) over
 * all key bits.  Result in keys_out/vals_out; keys/vals are left untouched
 * unless copy_back != 0, which also leaves a sorted copy there as the
 * reference does (radix.py:158-169).
 * Scratch sizes (here and col_collide_scratch_bytes) are monotone in n: a buffer
 * sized for n serves every call with n' <= n of the same key/value widths. */
size_t col_radix_scratch_bytes(uint64_t n, int key_bytes, int val_bytes);
int col_radix_sort(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                   uint64_t n, int key_bytes, int val_bytes, void *scratch, int copy_back);
/* One pass of the production sort, for profiling and per-pass parity:
 * histogram (digit-major, hist[d*nblocks+b]) -> scan -> scatter. */
uint32_t col_radix_tile(uint64_t n, int key_bytes, int val_bytes);   /* elements per block for an n-element sort: it depends
                                           on all three (1024 / 4096 / 8192, and 16384 for u32 key-only sorts from 32 Mi
                                           keys); pass the SAME val_bytes to the histogram and the scatter of a pass */
int col_radix_histogram(void *stream, const void *keys, uint64_t n, int key_bytes, int val_bytes,
                        int pass, uint32_t *hist);
int col_radix_scatter(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                      uint64_t n, int key_bytes, int val_bytes, int pass, const uint32_t *offsets);
/* The reference's own per-pass kernels with its block structure (block =
 * 2*group_size, `bits` per pass, digit-major histogram), for the kernel-level
 * parity tests (collision/radix.cl:48-139, tests/test_radix.py:61-351). */
int col_ref_block_sort(void *stream, void *keys, void *vals, uint64_t n, int key_bytes, int val_bytes,
                       uint32_t block, int bits, int pass, uint32_t *hist);
int col_ref_scatter(void *stream, const void *keys, void *keys_out, const void *vals, void *vals_out,
                    uint64_t n, int key_bytes, int val_bytes, uint32_t block, int bits, int pass,
                    const uint32_t *offsets, const uint32_t *hist);

/* ---------------------------------------------------------------- LBVH
 * col_bvh_build replaces fillInternal + generateBVH (collision/collision.py:
 * 171-180, collision/collision.cl:55-121): nodes[2n-1] from sorted codes/ids.
 * If `bounds` is not NULL the traversal links are also written into the unused
 * lane w of each node's Bound (min.w = next node when this subtree is skipped,
 * max.w = left child or, for a leaf, the sphere id) -- see DESIGN.md. */
int col_bvh_build(void *stream, const uint32_t *codes, const uint32_t *ids, col_node *nodes,
                  void *bounds, uint32_t n, int coord_bytes);
/* Replaces leafBounds + internalBounds (collision/collision.py:181-190,
 * collision/collision.cl:128-162).  flags: 2n-1 uint32, zeroed by the caller
 * (collision.py:147-150).  Lane w of bounds is preserved. */
int col_bvh_refit(void *stream, void *bounds, uint32_t *flags, const void *coords, const void *radii,
                  const col_node *nodes, uint32_t n, int coord_bytes);
/* Replaces traverse (collision/collision.py:191-196, collision/collision.cl:
 * 174-226).  counter: 1 uint32 zeroed by the caller, receives the TOTAL hit
 * count; pairs may be NULL when capacity == 0.  Needs the links written by
 * col_bvh_build(bounds != NULL).
 * Alignment of `bounds` (here and in col_traverse_chunked, col_collide*, col_traverse_ghost_slots): any 32-byte
 * aligned pointer is accepted; the fast walk (f32 records below 4 GB) is taken when it is 64-byte aligned -- what
 * col_malloc / hipMalloc / a torch allocation give -- because its leaf-block test uses 64-byte scalar loads, which
 * may read up to 32 bytes past the last record (never used, but inside the allocation of an aligned array).  Other
 * pointers take the generic walk: same pairs, slower. */
int col_traverse(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity,
                 const col_node *nodes, const void *bounds, uint32_t n, int coord_bytes);

/* col_traverse for scenes with millions of pairs (same list, same counter semantics): workgroups take list space in
 * chunks of 8192 pairs -- one atomic on the counter per chunk instead of one per 512 pairs, which is what bounds the walk
 * on BASELINE config 3 (50 000 atomics on one address) -- and a small kernel closes the holes the chunks leave; if the
 * chunks ran past `capacity` the exact walk runs again (4 launches instead of 1).  *counter must be 0 on entry.
  * record arrays below 4 GB and 64-byte aligned (f32 and, since round 4, f64); anything else takes col_traverse.  scratch: col_traverse_chunked_scratch_bytes(). */
size_t col_traverse_chunked_scratch_bytes(void);
int col_traverse_chunked(void *stream, uint32_t *pairs, uint32_t *counter, uint32_t capacity, const col_node *nodes,
                         const void *bounds, uint32_t n, int coord_bytes, void *scratch);

/* Fused production form of the three calls above minus the traversal: Karras topology, leaf and
 * internal AABBs and the traversal links in one pass, with no inter-workgroup hand-off (node
 * boxes are range queries over the sorted leaf boxes; see csrc/lbvh.hip).  Same `nodes` and
 * `bounds` contents as col_bvh_build(bounds != NULL) + col_bvh_refit. */
size_t col_lbvh_scratch_bytes(uint32_t n, int coord_bytes);
int col_lbvh(void *stream, const uint32_t *codes, const uint32_t *ids, const void *coords,
             const void *radii, col_node *nodes, void *bounds, void *scratch, uint32_t n,
             int coord_bytes);

/* ---------------------------------------------------------------- whole path
 * Replaces Collider.get_collisions (collision/collision.py:130-198): the whole
 * enqueue DAG as one call.  codes/ids: two buffers of `padded` uint32 each
 * (sorted result in codes[1]/ids[1], as in the reference); nodes 2n-1;
 * bounds (2n-1)*8 scalars; flags 2n-1 uint32; scratch from
 * col_collide_scratch_bytes. */
size_t col_collide_scratch_bytes(uint32_t n, uint32_t padded, int coord_bytes);
int col_collide(void *stream, const void *coords, const void *radii, uint32_t n, uint32_t padded,
                int coord_bytes, uint32_t *codes0, uint32_t *codes1, uint32_t *ids0, uint32_t *ids1,
                col_node *nodes, void *bounds, uint32_t *flags, void *scratch,
                uint32_t *counter, uint32_t *pairs, uint32_t capacity);

/* col_collide with a choice of sort for inputs of up to 4 000 000 spheres (above, the plan is ignored):
 *   COL_SORT_LSD  the four-pass LSD sort (what col_collide uses);
 *   COL_SORT_MSD  one global pass on the top 8 code bits, then every bucket finished inside one workgroup's
 *                 LDS -- 6 launches less.  Same outputs.  A bucket of more than 8192 pairs (16384 above 1.9 M
 *                 spheres; clustered scenes)
 *                 is still sorted correctly but slowly, and its size is stored to oversize[0] (TWO words of memory
 *                 the device can write, e.g. from col_host_alloc; may be NULL): the caller should then go back
 *                 to COL_SORT_LSD.  A call that takes COL_SORT_LSD where the MSD plan could apply stores
 *                 0x80000000 | g to oversize[1], g = the largest MSD bucket of ITS codes (read off the sorted
 *                 codes): g above the bucket capacity means the MSD plan would have met an oversize bucket.
 *                 oversize[2] receives 0x80000000 | (the pair count the counter held when this call zeroed it,
 *                 i.e. the PREVIOUS call's result on the same counter): a caller that sees millions of pairs
 *                 there ORs COL_TRAVERSE_CHUNKED into sort_plan.  `oversize` is THREE words.
 *                 collision_amd.collision.Collider follows all three on its own. */
/* The MSD plan's sort on its own: (u32 code, u32 id) pairs, codes 30-bit or the 0xFFFFFFFF pad, n up to
 * 4 000 000.  The digit-major histogram of the bucket digit (bits 22..29) per col_radix_tile(n)-code tile --
 * what the fused Morton kernel leaves -- must be at the start of `scratch` (col_radix_scratch_bytes(n, 4, 4)). */
int col_radix_sort_msd(void *stream, const uint32_t *keys, uint32_t *keys_out, const uint32_t *vals,
                       uint32_t *vals_out, uint64_t n, void *scratch, uint32_t *oversize);
#define COL_SORT_LSD 0
#define COL_SORT_MSD 1
/* ORed into sort_plan: the traversal takes pair-list space in chunks of 8192 pairs (col_traverse_chunked) -- for scenes
 * with millions of pairs, where one atomic on the pair counter per 512 pairs is what bounds the walk.  Same list. */
#define COL_TRAVERSE_CHUNKED 2
int col_collide_plan(void *stream, const void *coords, const void *radii, uint32_t n, uint32_t padded_size,
                     int coord_bytes, uint32_t *codes0, uint32_t *codes1, uint32_t *ids0, uint32_t *ids1,
                     col_node *nodes, void *bounds, uint32_t *flags, void *scratch, uint32_t *n_collisions,
                     uint32_t *collisions, uint32_t capacity, int sort_plan, uint32_t *oversize);
/* The same call with the first launch of the path -- the block partials of the scene bounds, [min row, max row] of
 * `coords` -- already made: col_minmax4_stage1_dev computes them with the row count read from DEVICE memory
 * (*n_dev, at most n_max; `partials` = 256 records of 8 scalars, *parts = the number written), so it can be enqueued
 * before the host knows the count (the multi-GPU step: right behind col_partition_unpack, while the host still polls
 * for the owned count; the device then has work when the rest of the path arrives).  partials == NULL: as
 * col_collide_plan.  (The fused front end applies at every size since the end of round 3.) */
int col_minmax4_stage1_dev(void *stream, const void *rows, const uint32_t *n_dev, uint32_t n_max, int coord_bytes,
                           void *partials, uint32_t *parts);
int col_collide_plan_partials(void *stream, const void *coords, const void *radii, uint32_t n, uint32_t padded_size,
                              int coord_bytes, uint32_t *codes0, uint32_t *codes1, uint32_t *ids0, uint32_t *ids1,
                              col_node *nodes, void *bounds, uint32_t *flags, void *scratch, uint32_t *n_collisions,
                              uint32_t *collisions, uint32_t capacity, int sort_plan, uint32_t *oversize,
                              const void *partials, uint32_t parts);
/* The same call with the NUMBER OF SPHERES ON THE DEVICE (round 4: the multi-GPU step, where a rank learns how many spheres it
 * owns from an exchange the host does not wait for).  *n_dev (device memory, written by earlier work on the stream) holds the
 * real count; n is a host-known upper bound -- the capacity of the arrays -- that sizes grids, scratch and the sort:
 * rows from *n_dev on are sorted as pads and every kernel works on min(n, *n_dev) spheres.  Outputs exactly those of
 * col_collide_plan_partials(n = *n_dev) for the first *n_dev sorted entries, all 2 * (*n_dev) - 1 nodes and boxes and the
 * pair list.  `partials` is required (col_minmax4_stage1_dev with the same word).  n_dev == NULL: as above. */
int col_collide_plan_dev(void *stream, const void *coords, const void *radii, uint32_t n, uint32_t padded_size,
                         int coord_bytes, uint32_t *codes0, uint32_t *codes1, uint32_t *ids0, uint32_t *ids1,
                         col_node *nodes, void *bounds, uint32_t *flags, void *scratch, uint32_t *n_collisions,
                         uint32_t *collisions, uint32_t capacity, int sort_plan, uint32_t *oversize,
                         const void *partials, uint32_t parts, const uint32_t *n_dev);

/* ---------------------------------------------------------------- multi-GPU helpers
 * New work (the reference is single-device): device side of the sphere
 * repartition / halo exchange / ghost queries driven by collision_amd/multi.py over RCCL.
 * coord_bytes 4 | 8.  Rows are 4 scalars (x, y, z, r); a transport record is those 4 scalars
 * followed by the 32-bit global id: 5 words (f32) or 9 words (f64). */
/* -- the repartition into Morton ranges of equal population: four calls between the collectives --
 * col_partition_sample (2 launches): payload = `samples` evenly strided rows of rows[0..n) (n == 0: rows at +inf),
 * then the min row and the max row of all n: what a rank contributes to the first all-gather.  scratch:
 * col_partition_scratch_bytes() bytes (block partials).  Also clears zero[0 .. zero_count) (zero_count <= 256;
 * the step's flag words). */
size_t col_partition_scratch_bytes(void);
int col_partition_sample(void *stream, const void *rows, uint32_t n, uint32_t samples, void *payload, void *scratch,
                         uint32_t *zero, uint32_t zero_count, int coord_bytes);
/* col_partition_plan (3 launches): gathered = [world][samples + 2] rows (every rank's payload; at most 16384 rows).
 * range8 = the global scene range; splitters = the world - 1 quantiles of the gathered rows' Morton codes;
 * dest[i] = owner of row i (number of splitters <= its code); hist (256 * ceil(n / col_radix_tile(n, 4, 4)) words)
 * = the scanned owner histogram col_partition_group needs; owner_counts[q] = rows of owner q.  An empty rank
 * (n == 0): hist is not touched. */
int col_partition_plan(void *stream, const void *gathered, uint32_t world, uint32_t samples, const void *rows, uint32_t n,
                       void *range8, uint32_t *splitters, uint32_t *dest, uint32_t *hist, uint32_t *owner_counts,
                       int coord_bytes);
/* col_partition_group (1 launch): stable grouping by owner and packing in one pass over the rows (dest / hist /
 * owner_counts as col_partition_plan left them; world <= 16): the rows this rank keeps go to the front of
 * own_rows / own_gids / own_radii (`capacity` rows each); the others into `send`: one SLOT of 1 + slot transport
 * records per other rank (rank order) -- a header record whose first word is the full length of the list, then
 * min(length, slot) records.  Rows of a list beyond its slot STAY with this rank: they follow the kept rows in the
 * owned arrays (owner order), so a slot that is too small never loses a sphere (ownership only decides load
 * balance and halo size).  flags[2] = max(flags[2], longest list). */
int col_partition_group(void *stream, const void *rows, const uint32_t *gids, uint32_t n, const uint32_t *dest,
                        const uint32_t *hist, const uint32_t *owner_counts, uint32_t world, uint32_t rank,
                        uint32_t slot, void *send, void *own_rows, uint32_t *own_gids, void *own_radii,
                        uint32_t capacity, uint32_t *flags, int coord_bytes);
/* col_partition_unpack (1 launch): recv = the slots received from the other ranks (laid out as `send`) appended to
 * the owned arrays behind the rows col_partition_group left there (owner_counts[rank] kept rows + the overflow of
 * the lists it sent: sum over q != rank of max(owner_counts[q] - slot, 0)); a received header longer than the slot
 * means the sender kept the rest.  owned[0] = min(m, capacity), owned[1] = m
 * (device words), *host_word (host-visible 64-bit word, e.g. col_host_alloc; may be NULL) = seq << 32 | m: the
 * host waits for that word only, then sizes the local pipeline.  flags[2] = max(flags[2], longest header). */
int col_partition_unpack(void *stream, const void *recv, uint32_t world, uint32_t rank, uint32_t slot,
                         const uint32_t *owner_counts, void *own_rows, uint32_t *own_gids, void *own_radii,
                         uint32_t capacity, uint32_t *owned, void *host_word, uint32_t seq, uint32_t *flags,
                         int coord_bytes);
int col_unpack_radii(void *stream, const void *rows, uint32_t n, void *radii, int coord_bytes);
/* A rank's REGION for the halo selection: 8 boxes, one per octant of the global scene range (the top three bits of
 * a centre's Morton code under range8, from
 * col_partition_plan; NULL = no repartition: one box, the other seven inverted), each (min centre - max r, 0,
 * max centre + max r, 0) over the spheres of rows[0..n) in that octant (conservative; an empty octant: an inverted
 * box nothing overlaps).  out = 8 x 2 rows of 4 scalars.  Two launches; scratch as for col_partition_sample (its own
 * buffer when the two run on different streams); also clears zero[0 .. zero_count) (the halo list counters). */
int col_region_boxes(void *stream, const void *rows, uint32_t n, const void *range8, void *scratch, void *out,
                     uint32_t *zero, uint32_t zero_count, int coord_bytes);
int col_region_boxes_dev(void *stream, const void *rows, uint32_t n, const void *range8, void *scratch, void *out,
                     uint32_t *zero, uint32_t zero_count, int coord_bytes,
        const uint32_t *n_dev);   /* ... with the row / tree count on the device: n is a bound, see col_collide_plan_dev */
/* halo selection in one launch: boxes = DEVICE array [world][8 boxes][8] (lo.xyz,-,hi.xyz,-) as produced by
 * the AABB all-gather of every rank's col_region_boxes; peers = HOST array of n_peers <= 8 rank numbers; lists[k*stride ...] and
 * counts[k] (zeroed beforehand: col_region_boxes does it) receive the spheres overlapping any box of peers[k] */
int col_select_overlap_multi(void *stream, const void *rows, uint32_t n, const void *boxes, const int *peers,
                             int n_peers, uint32_t stride, uint32_t *lists, uint32_t *counts, int coord_bytes);
int col_select_overlap_multi_dev(void *stream, const void *rows, uint32_t n, const void *boxes, const int *peers,
                             int n_peers, uint32_t stride, uint32_t *lists, uint32_t *counts, int coord_bytes,
        const uint32_t *n_dev);   /* ... with the row / tree count on the device: n is a bound, see col_collide_plan_dev */
/* the n_lists lists as fixed SLOTS of 1 + slot_records records each: a header record (first word = the list's
 * full length) followed by min(length, slot_records) records; list sizes are read on the device.  A fixed-size
 * exchange of such slots needs no count exchange and no host sync; the receiver learns the lengths -- and an
 * overflow -- from the headers. */
int col_pack_slots(void *stream, const void *rows, const uint32_t *gids, const uint32_t *lists, uint32_t stride,
                   const uint32_t *counts, int n_lists, uint32_t max_per_list, void *rec, uint32_t rec_capacity,
                   uint32_t slot_records, int coord_bytes);
/* ghost spheres (n_slots received slots) as queries against the local tree (bounds with links); emits
 * (ghost gid, local_gids[hit]); counter is NOT reset (it continues the local pair list);
 * flags[0] = max(flags[0], longest header length), flags[1] += ghosts queried (2 x uint32, zeroed by the caller).
 * scratch (col_ghost_scratch_bytes(n_slots, slot_records) bytes): the ghosts are ordered by a coarse Morton key (one
 * launch + two 8-bit sort passes) and walk the tree as PACKETS of 64 neighbours, like the local queries (4 + 6
 * launches); scratch == NULL: one launch, every lane walks its own ghost from the root (slower beyond a few
 * thousand ghosts). */
size_t col_ghost_scratch_bytes(uint32_t n_slots, uint32_t slot_records);
int col_traverse_ghost_slots(void *stream, const void *rec, uint32_t n_slots, uint32_t slot_records, const void *bounds,
                             uint32_t n, const uint32_t *local_gids, uint32_t *pairs, uint32_t *counter,
                             uint32_t capacity, uint32_t *flags, int coord_bytes, void *scratch);
int col_traverse_ghost_slots_dev(void *stream, const void *rec, uint32_t n_slots, uint32_t slot_records, const void *bounds,
                             uint32_t n, const uint32_t *local_gids, uint32_t *pairs, uint32_t *counter,
                             uint32_t capacity, uint32_t *flags, int coord_bytes, void *scratch,
        const uint32_t *n_dev);   /* ... with the row / tree count on the device: n is a bound, see col_collide_plan_dev */
/* pairs[first .. min(*count, capacity)) : index -> gids[index] */
int col_translate_pairs(void *stream, uint32_t *pairs, const uint32_t *count, uint32_t first,
                        uint32_t capacity, const uint32_t *gids);

/* ---------------------------------------------------------------- index / offset
 * collision/index.cl:1-13 (Indexer.gather/scatter, index.py:23-55) and
 * collision/offset.cl:3-12 (OffsetFinder.find_offsets, offset.py:37-49). */
int col_gather(void *stream, const void *in, const void *indices, void *out, uint64_t n,
               int val_bytes, int index_bytes);
int col_scatter(void *stream, const void *in, const void *indices, void *out, uint64_t n,
                int val_bytes, int index_bytes);
int col_find_offsets(void *stream, const void *values, uint64_t n_values, void *offsets,
                     uint64_t n_offsets, int value_bytes, int offset_bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* COLLISION_HIP_H */

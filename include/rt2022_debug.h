/*
 * rt2022_debug.h — probe entry points of librt2022.so used by the parity tests:
 * they run pieces of the device arithmetic (rt_math.h, the path RNG) on the GPU so
 * that tests can compare them bit for bit with the CPU oracle. Not part of the
 * drop-in boundary.
 */
#ifndef RT2022_DEBUG_H
#define RT2022_DEBUG_H
#include "rt2022.h"
#ifdef __cplusplus
extern "C" {
#endif

/* op: 0 sin, 1 cos, 2 acos, 3 atan2(a,b), 4 log, 5 sqrt, 6 a/b. Host buffers. */
int rt_debug_math_device(int op, const double *a, const double *b, double *out, uint64_t n);
/* n draws from Rng(state): mode 0 next_u64, 1 gen_f64 (bits), 2 gen_range(lo,hi) (bits), 3 gen_index(bound). */
int rt_debug_rng_device(uint64_t state, int mode, double lo, double hi, uint64_t bound, uint64_t *out, uint64_t n);
/* Scheduler knobs of the engines. node_quorum, a bit field:
 *   0-7   lanes that must want a BVH-node step before the wave takes the node fast path without a
 *         vote (1..64);
 *   8-15  (unused: two sphere tests per turn and a tail factor of 2 are built in);
 *   16-19 pool size of the wavefront engine: segments of 4096 path slots per resident traversal
 *         workgroup (1..8, default 8);
 *   20-23 s: every segment's ray list is ordered longest-first by (expected node steps) >> s, 0 = slot order;
 *   24-27 groups the pool is cut into, each alternating its passes on a stream of its own (1..8; 0, the default: the library's
 *         choice — two, one for scenes with triangle meshes; RT_FLAG_KERNEL_TIMES, the pass-timing probe and the partial-sum
 *         ring keep one);
 *   28    keep the BVH's node table out of LDS: the plain traversal kernels even where the node-table variant applies
 *         (rt_debug_trace_variant says which one a scene takes);
 *   29    run the pass-timing probe (rt_debug_pass_timing);
 *   30    take the literal AABB step only (test hook).
 * vote_weights: 4 bits per operation label from the lowest nibble up (node, sphere, rect, box, medium, misc,
 * ctx, done = publish + refill); the vote picks the label with the largest lanes * weight; 0 = the engine's
 * default (wavefront 0x24444442: node steps outside the fast path and the refill yield to the arms; megakernel
 * 0x22222221). All of this affects speed only, never results. */
int rt_debug_set_tuning(rt_scene *scene, uint32_t node_quorum, uint32_t vote_weights);
/* Engine behind rt_render*: 1 (default) = wavefront passes (pt_wavefront.hip), 0 = the single
 * megakernel (pt_kernel.hip). max_pool_blocks: segments of 4096 path slots in the pool (0 = the default of the tuning word).
 * Results are bit-identical between the two; only speed differs. */
int rt_debug_set_engine(rt_scene *scene, int engine, int max_pool_blocks);
/* Partial-sum ring of the default engine (one-sample work items): planes = 0 automatic (a ring of at most 24 GiB when all spp
 * planes would exceed 40 % of the device's memory), -1 never, n > 0 force a ring of n planes (rounded down to a power of two; tests go down to one
 * plane, where every sample waits for its predecessor). Memory and speed only: the sums are the same bit for bit. */
int rt_debug_set_partial_ring(rt_scene *scene, int planes);
/* Scheduler census of the last counter run (RT_FLAG_COUNTERS) of the wavefront traversal kernel:
 * per operation label (node, sphere, rect, box, medium, misc, ctx, done; [8] = node fast path) the
 * number of wave-rounds and the lanes they served: lanes / (64 * rounds) = lane utilisation. */
int rt_debug_census(const rt_scene *scene, uint64_t rounds[9], uint64_t lanes[9]);
/* Timing probe of the traversal passes of the last render made with tuning bit 29 set (the host then
 * synchronises after every pass): sums over passes, in 100 MHz ticks, of {pass span (first wave start
 * to last wave end), mean wave lifetime, mean wave time after the workgroup's ray list ran dry}, then
 * {passes, waves}. lifetime / span = how evenly the waves finish; dry / lifetime = the under-filled tail. */
int rt_debug_pass_timing(const rt_scene *scene, double out[5]);
/* HBM counter calibration: moves a known number of bytes in the path pool's access patterns over a buffer of
 * buffer_bytes (use several GiB: far beyond the Infinity Cache). mode 0 streaming read (16 B per lane, the whole
 * buffer), 1 the first 64 B of n_access random 128-byte records, 2 whole random records, 3 streaming write,
 * 4 32 B written at +64 of random records, 5 64 + 32 B written per random record, 6 single bytes at random places.
 * Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/traffic_calib.sh). */
int rt_debug_traffic_probe(int mode, uint64_t buffer_bytes, uint64_t n_access, uint64_t seed);
/* VALU counter calibration: a kernel whose vector pipes are saturated by construction — nothing but independent
 * vector instructions of one kind, 8 waves per SIMD on every CU, `iters` rounds of 64 of them per lane. mode 0 v_fma_f64,
 * 1 32-bit integer add / xor, 2 v_fma_f64 with half of every wave's lanes switched off, 3 f64 and 32-bit alternating,
 * 4 v_fma_f64 at one wave per SIMD. Run under rocprofv3 --pmc (tools/valu_calib.sh): what the SQ counters read for
 * it is the normalisation bench.py turns the traversal kernel's counters into a busy fraction with. */
int rt_debug_valu_probe(int mode, uint32_t iters);
/* Traversal-stack entries the scene needs and the persistent grid size used for it. */
int rt_debug_scene_info(const rt_scene *scene, uint32_t *stack_need, int32_t *grid_blocks);
/* The traversal kernel variant the scene's timed renders (no RT_FLAG_COUNTERS) take: threads per workgroup, stack
 * entries per lane, how many of the scene's node records the workgroup keeps in LDS (0: the plain kernels, every
 * node fetched through L1 / L2), and in *spheres_in_lds two flags: bit 0 — its Sphere / MovingSphere pools are there too (small
 * sphere-only scenes); bit 1 — it tests node boxes in single precision, with the double-precision test wherever the two could
 * differ (sphere scenes: 44-byte records in LDS, or 32-byte records from HBM / L2 and a partial LDS table). */
int rt_debug_trace_variant(const rt_scene *scene, uint32_t *workgroup_threads, uint32_t *stack_entries, uint32_t *nodes_in_lds,
                           uint32_t *spheres_in_lds);
/* The single-precision slab test of the traversal kernel for sphere-only scenes: out[3] = the build's
 * RT2022_F32_SLABS (0 off, 1 the all-in-LDS instance of sphere-only scenes, 2 every instance that keeps the whole node table
 * in LDS). A diagnostic build (-DRT2022_F32_CENSUS, out[2] = 1) also makes the double-precision test beside every verdict
 * the single-precision one takes and counts, over all renders since the last call: out[0] node steps of the fast path that
 * took the single-precision test, out[1] those it left undecided, out[4] verdicts that differed (must be 0). The call clears
 * the counters. */
int rt_debug_f32_slabs(uint64_t out[5]);

#ifdef __cplusplus
}
#endif
#endif

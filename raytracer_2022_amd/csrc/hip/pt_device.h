// pt_device.h — device-side views of the flattened scene and the launch interface
// between rt_api.hip (C ABI, validation, HBM residency) and pt_kernel.hip (kernels).
#ifndef RT2022_PT_DEVICE_H
#define RT2022_PT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../../include/rt2022.h"

namespace rt2022 {

// ConstantMedium as the traversal kernel wants it: one 64-byte record = one fetch, with the boundary's data inline
// when the boundary is a plain Sphere (the fog and the subsurface ball of the final scene) — rt_medium alone would cost
// a dependent second fetch of the sphere record.
struct MediumDev {
    double center[3], radius;          // boundary sphere (zeros when the boundary is something else)
    double neg_inv_density;
    uint32_t boundary, mat;            // mat = material index | slot kind << kMatKindShift, like the primitive pools
    uint32_t sphere_boundary;          // 1: the boundary is a plain Sphere
    uint32_t _pad[3];
};

// A material as the shading kernel wants it: one 80-byte record = one fetch, with the top-level record of its texture
// inline (kind, children / table ids, scale, colour) — a SolidColor albedo then needs no second, dependent fetch.
struct MaterialDev {
    uint32_t tex, tex_kind;            // rt_material::tex and that texture's kind
    double albedo[3];                  // Metal
    double param;                      // Metal fuzz / Dielectric ir
    double tex_color[3];               // rt_texture::color
    double tex_scale;                  // rt_texture::scale
    uint32_t tex_a, tex_b;             // rt_texture::a, b
};

// Pointers into HBM, one pool per kind (layouts = include/rt2022.h).
struct SceneDev {
    const rt_bvh_node *nodes;
    // The node table once more in single precision, 32 bytes per node: {min.x, max.x, min.y, max.y | min.z, max.z, left, push ref}
    // (floats rounded to nearest from the doubles of `nodes`, same numbering) — what the sphere-scene traversal kernels that cannot
    // keep the whole table in LDS fetch per node step (wf_trace, kF32G; the double-precision record serves the undecided steps).
    const uint32_t *nodes32;
    const rt_sphere *spheres;
    const rt_moving_sphere *moving_spheres;
    const rt_rect *rects;
    const rt_box *boxes;
    const rt_triangle *triangles;
    const rt_ring *rings;
    const rt_medium *media;
    const rt_xform *xforms;
    const rt_list *lists;
    const uint32_t *list_items;
    const uint32_t *lights;
    const rt_material *materials;
    const rt_texture *textures;
    const rt_image *images;
    const uint8_t *image_data;
    const rt_perlin *perlins;
    uint32_t root;
    uint32_t n_lights;
    const MaterialDev *materials_dev;  // [n_materials], shading kernel's view of `materials` + `textures`
    const MediumDev *media_dev;        // [n_media], traversal kernel's view of `media`
    uint32_t media_mode;               // 0: no medium has a plain-sphere boundary, 1: all have, 2: mixed (look at the record)
    uint32_t n_nodes;                  // records in `nodes`
    uint32_t n_xforms, n_media;        // records in `xforms`, `media_dev`
    uint32_t n_spheres, n_moving_spheres;
    uint32_t n_rects;                  // (0 with FEAT 0: every primitive is a sphere — the scenes that take the single-precision slab test)
};

// Counter block in HBM (same order as rt_stats' integer fields).
struct StatsDev {
    unsigned long long paths, rays, node_visits;
    unsigned long long prim_tests[RT_KIND_COUNT];
    unsigned long long light_pdf_tests, rng_draws;
    // Scheduler census of the traversal kernel (counter builds only): per operation label, how many
    // times a wave ran it and how many lanes it served; [8] = the node fast path.
    unsigned long long op_rounds[9], op_lanes[9];
};

struct RenderArgs {
    rt_camera cam;
    uint32_t width, height, spp, max_depth;
    uint32_t n_frames, n_rows;
    uint32_t chunk, n_chunks;          // samples per work item, items per pixel
    double background[3];
    double t_min;
    double split[3];                   // centre of the root's bounding box (ordering of the ray lists only)
    uint64_t seed;
    uint64_t n_pixels;                 // n_rows * width
    uint64_t n_items;                  // n_pixels * n_chunks
    const uint32_t *row_ids;           // device
    double *partial;                   // [n_chunks][n_pixels][3] (== out when n_chunks == 1)
    unsigned long long *work_counter;  // zeroed before launch
    double *tape;                      // bounce records: max_depth * 4 doubles per launched lane
    // Ring of partial-sum planes (r3; one-sample work items only). 0: `partial` holds all n_chunks planes and chunk_sum adds
    // them at the end. R > 0: the samples are taken in GROUPS of ring_group consecutive ones (a divisor of spp, R a multiple of
    // it): work items are numbered group-major — item = (group * n_pixels + pixel) * ring_group + sample within the group, so
    // that the samples of a pixel in a group are still neighbours on the work counter (their camera rays stay coherent: plain
    // sample-major order cost the traversal kernel 8 % on the headline and 30 % on C5) — sample c of a pixel goes to plane c mod R,
    // the host adds finished groups of planes to the output in sample order as the frame goes (ring_accumulate) and raises
    // *claim_limit — the number of work items that may be handed out — behind them: R planes instead of spp, the same sums bit
    // for bit (pixel_color += ..., main.rs:150, in sample order).
    uint32_t ring, ring_group;
    const unsigned long long *claim_limit;
    uint32_t node_quorum;              // lanes that must want a node step for the fast path (1..64)
    uint32_t vote_weights;             // 4 bits per operation label: the vote picks max(lanes * weight)
    StatsDev *stats;                   // may be null
};

// ---- wavefront engine (pt_wavefront.hip) -------------------------------------------
// What a path slot waits for.
enum SlotKind : uint32_t {
    SK_IDLE = 0,        // nothing left to do
    SK_FRESH = 1,       // no path yet: start the first sample
    SK_TRACE = 2,       // carries a ray: world.hit pending
    SK_MISS = 3,
    SK_LIGHT = 4,
    // Lambertian by albedo texture: a wave that holds one noise-textured hit pays seven octaves of
    // Perlin for all 64 lanes, so the texture kind is part of the sort key.
    SK_LAMB_SOLID = 5,
    SK_LAMB_CHECKER = 6,
    SK_LAMB_NOISE = 7,
    SK_LAMB_IMAGE = 8,
    SK_METAL = 9,
    SK_DIELECTRIC = 10,
    SK_ISOTROPIC = 11,
    SK_COUNT = 12
};

// `mat` of the device copies of the primitive pools = material index | slot kind of a hit on it << kMatKindShift.
constexpr uint32_t kMatKindShift = 24, kMatIndexMask = (1u << kMatKindShift) - 1u;

// A pool of path slots in HBM, one array of records per field; segment b (= shade workgroup b) owns
// the slots [b*kSlotsPerBlock, (b+1)*kSlotsPerBlock) for the whole frame.
#ifndef RT2022_SLOTS_PER_SEGMENT
#define RT2022_SLOTS_PER_SEGMENT 4096
#endif
constexpr int kSlotsPerBlock = RT2022_SLOTS_PER_SEGMENT;     // (a multiple of 256, at most 32768: list entries are u16)
constexpr uint64_t kRecBytes = 128, kRecDoubles = kRecBytes / 8, kRecWords = kRecBytes / 4;      // the slot record
struct WfPool {
    uint32_t n_slots;
    uint32_t n_blocks;      // segments
    uint8_t *kind;          // [P]    what a listed slot waits for (SlotKind), BY POSITION ON ITS SEGMENT'S RAY LIST (see `list`)
    // ONE 128-byte record per slot = one cache line, three parts (`ray`, `hit`, `state` point at their part of slot 0;
    // stride kRecDoubles doubles / kRecWords words): the shade pass, which visits slots in sorted order, then pulls one
    // line per slot instead of one line from each of three arrays (measured: shade's HBM reads per ray segment).
    //   +0   ray:   {ox oy oz dx dy dz tm, rng state}                                                        64 B
    //   +64  hit:   {t (f64), leaf ref, box face | movers << 4 | node steps << 16, 3 mover refs, 4th ref or material word}   32 B
    //   +96  state: {item = pixel slot * n_chunks + chunk (u64), next sample, end sample, remaining depth, px, py, frame}    32 B
    double *ray;
    uint32_t *hit;
    uint32_t *state;
    double *pixel_sum;      // [P][4]  running sum of the item (4th double unused)
    double *tape;           // [P][tape_cap][4] bounce records {w.x, w.y, w.z, p}
    uint32_t tape_cap;      // records per slot (>= max_depth)
    // Ray list of every segment (= the 4096 slots one shade workgroup owns), written by the shade pass:
    // local indices of the slots that carry a ray, longest expected traversal first. A trace workgroup
    // works through the lists of `segs` consecutive segments.
    uint16_t *list;         // [P]
    uint32_t *list_n;       // [n_blocks]
    uint32_t segs;          // segments per trace workgroup (n_blocks is a multiple of it)
    uint32_t *next_chunk;   // trace pass: the next chunk of list entries to hand out (cleared by the shade pass)
    uint32_t *max_list;     // [2] longest segment list of the pass, by pass parity (bounds the chunk ids)
    uint32_t n_cus;         // compute units of the device (size of the persistent trace grid)
    uint32_t *n_active;     // [2] rays handed to the next trace pass, by pass parity (polled by the host)
    uint32_t *fault;        // [1] engine invariants found broken on the device (bit 0: a slot reached the shade pass untraced)
    // Ring mode (RenderArgs::ring): the oldest sample GROUP with a path still in flight after a shade pass, by pass parity (every
    // group below it is finished: the host consumes their planes), and per segment the slots that asked for a work item and found
    // the ring full: listed behind the segment's rays, with kind FRESH, so that the next shade pass asks again.
    unsigned long long *oldest;     // [2]
    uint32_t *starved_n;            // [n_blocks]
    // Pass-timing probe (rt_debug_pass_timing; null otherwise): {first wave start, last wave end, sum of
    // wave lifetimes, sum of wave time after the list ran dry, waves} in wall_clock64 ticks.
    unsigned long long *dbg;
};

// Traversal-stack capacities the megakernel is instantiated for.
constexpr int kStackSmall = 22;   // 22 KiB of LDS per workgroup; the lean kernels run four workgroups per CU (VGPR-bound)
constexpr int kStackMid = 30;     // million-triangle meshes need ~26 entries; built for four workgroups per CU
constexpr int kStackLarge = 64;
constexpr int kBlock = 256;
// Vote weights of the traversal schedulers, four bits per operation label from the lowest nibble up: node, sphere, rect,
// box, medium, misc, ctx, done (publish + refill). The wave runs the label with the largest lanes x weight.
constexpr uint32_t kWfVoteWeights = 0x24444442u;      // wavefront engine: node and refill yield to the arms
constexpr uint32_t kMegaVoteWeights = 0x22222221u;    // megakernel
// Node-cache variant of the traversal kernel (pt_wavefront.hip): one workgroup of 1024 threads per CU, stacks of 16
// entries (64 KiB), and the first kNodeCache node records in the remaining LDS (56 bytes each: 97 440 B).
constexpr int kCacheBlock = 1024;
constexpr int kStackTiny = 16;
#ifndef RT2022_NODE_CACHE
#define RT2022_NODE_CACHE 1740
#endif
constexpr int kNodeCache = RT2022_NODE_CACHE;
// The all-in-LDS instance for small sphere-only scenes: 600 node records (33 600 B), 256 Sphere records (36 B each) and
// 512 MovingSphere records (80 B each) beside the 64 KiB of stacks.
constexpr int kPrimNodes = 600, kPrimSpheres = 256, kPrimMoving = 512;
// Four traversal workgroups per CU = 4 waves per SIMD = a budget of 128 VGPRs: the kernel then needs 116 and spills
// nothing. Five (96 VGPRs, 27 spilled, 84 B of scratch per lane) measured 3 % slower in the same run, three 8-9 %
// slower (profiles/r2_ab_occupancy.log): the kernel is bound by instruction issue far more than by latency.
#ifndef RT2022_TRACE_BLOCKS_PER_CU
#define RT2022_TRACE_BLOCKS_PER_CU 4
#endif
constexpr int kTraceBlocksPerCU = RT2022_TRACE_BLOCKS_PER_CU;   // resident traversal workgroups per CU the lean kernels are built for

// Launchers (pt_kernel.hip). `stack_need` = entries the scene needs (host-computed).
hipError_t launch_render(const SceneDev &scene, const RenderArgs &args, uint32_t stack_need, bool counters,
                         int n_blocks_hint, hipStream_t stream);
// Arms of the traversal kernel a scene can reach (template FEAT of wf_trace).
constexpr unsigned kFeatMisc = 1;      // triangles, rings
constexpr unsigned kFeatMovers = 2;    // Translate / RotateY / Zoom, HittableList objects
constexpr unsigned kFeatVolumes = 4;   // Boxes, ConstantMedium

// Streams the wavefront engine runs its groups of segments on (owned by the caller).
constexpr int kMaxGroups = 8;
struct WfStreams {
    int n = 0;                         // groups wanted (1 = everything on the caller's stream)
    hipStream_t stream[kMaxGroups] = {};
    hipEvent_t ev[kMaxGroups][2] = {};
    uint32_t *h_active = nullptr;      // pinned, [kMaxGroups][2]
    unsigned long long *h_work = nullptr;   // pinned, [kMaxGroups][2]: the work counter as of the same batches (progress callback, ring mode)
    unsigned long long *h_oldest = nullptr; // pinned, [kMaxGroups][2]: WfPool::oldest as of the same batches (ring mode)
};
// rt_params::progress_cb as the engine sees it (host side only).
struct Progress {
    void (*cb)(void *, uint32_t, uint64_t, uint64_t) = nullptr;
    void *user = nullptr;
    uint64_t total = 0;                // camera paths of the call
    uint32_t per_item = 1;             // samples per work item
};
// Per-kernel device time of one render (RT_FLAG_KERNEL_TIMES): HIP events on the launch stream around every pass.
struct KernelTimes {
    std::vector<hipEvent_t> ev;        // grown on demand, reused from call to call
    double shade_ms = 0.0, trace_ms = 0.0;
};
struct RingCtl;                        // (below)
// Wavefront engine: alternates shade / trace passes over the pool until it drains.
// Blocks the calling thread (polls `n_active`). d_args is the device-resident copy of `args`.
hipError_t launch_render_wavefront(const SceneDev &scene, const RenderArgs &args, const RenderArgs *d_args,
                                   const WfPool &pool, uint32_t stack_need, unsigned features, bool counters,
                                   const WfStreams &gs, hipStream_t stream, uint32_t *out_iterations,
                                   double *timing /* null, or [5]: see rt_debug_pass_timing */,
                                   uint32_t *out_fault /* WfPool::fault after the last pass */,
                                   KernelTimes *kt /* null, or where to put the per-kernel times (three HIP events per pass pair on its group's stream) */,
                                   const Progress *progress = nullptr, const RingCtl *ring = nullptr);
hipError_t launch_chunk_sum(const double *partial, double *out, uint64_t n_values, uint32_t n_chunks, hipStream_t stream);
// Ring mode: what launch_render_wavefront needs to consume planes as the frame goes.
struct RingCtl {
    uint32_t planes = 0;               // 0 = off
    double *out = nullptr;             // [n_pixels * 3] the call's output: finished planes are added to it in sample order
    unsigned long long *d_limit = nullptr;   // device word behind RenderArgs::claim_limit
    uint32_t max_passes = 0;           // watchdog: more pass pairs than this is an engine error, not a long frame
};
hipError_t launch_tonemap(const double *rgb_sum, uint64_t n_pixels, int32_t spp, uint8_t *rgb8, hipStream_t stream);
hipError_t launch_math_probe(int op, const double *a, const double *b, double *out, uint64_t n, hipStream_t stream);
hipError_t launch_rng_probe(uint64_t state, int mode, double lo, double hi, uint64_t bound, uint64_t *out, uint64_t n, hipStream_t stream);
// Which traversal variant the wavefront engine launches for a scene without counters: {threads per workgroup, stack entries, nodes kept in LDS}.
void trace_variant(const SceneDev &scene, uint32_t stack_need, uint32_t tuning, unsigned features, uint32_t out[4]);
// {fast-path node steps that took the single-precision slab test, those it left undecided, 1 if this build counts (-DRT2022_F32_CENSUS),
// RT2022_F32_SLABS of the build, verdicts that differed from the double-precision test's (census builds make both)}; clears the counters.
hipError_t f32_slab_census(unsigned long long out[5]);
// Occupancy-derived persistent grid size for the given variant.
int render_grid_blocks(uint32_t stack_need, bool counters);

} // namespace rt2022
#endif

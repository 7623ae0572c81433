// rt_api.hip — the device half of the C ABI (include/rt2022.h): scene validation,
// HBM residency, launches. No CPU fallback exists: every entry point here needs a
// HIP device and reports RT_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../../include/rt2022.h"
#include "../../../include/rt2022_debug.h"
#include "../host/rt_error.hpp"
#include "pt_device.h"

using namespace rt2022;

namespace {

struct Fail {
    int code;
    std::string msg;
};
#define RT_REQUIRE(cond, code, msg) do { if (!(cond)) throw Fail{code, msg}; } while (0)
#define RT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw Fail{RT_ERR_DEVICE, std::string(#expr ": ") + hipGetErrorString(e_)}; } while (0)

template <class F>
int guarded(F &&f) {
    try {
        return f();
    } catch (const Fail &e) {
        set_error(e.msg);
        return e.code;
    } catch (const std::exception &e) {
        set_error(e.what());
        return RT_ERR_INVALID;
    }
}

// ---- validation -------------------------------------------------------------------
struct Validator {
    const rt_scene_desc &d;
    std::vector<int32_t> node_need;      // memo: stack need of each node (-1 unknown, -2 on the DFS stack)
    std::vector<int32_t> node_xdepth;
    mutable bool general_boundaries = false;   // some medium boundary is more than a primitive under movers
    bool in_boundary = false;                  // need(): inside a medium's boundary (media do not nest there)
    explicit Validator(const rt_scene_desc &desc) : d(desc), node_need(desc.n_nodes, -1), node_xdepth(desc.n_nodes, 0) {}

    uint32_t pool_size(uint32_t kind) const {
        switch (kind) {
            case RT_KIND_NODE: return d.n_nodes;
            case RT_KIND_SPHERE: return d.n_spheres;
            case RT_KIND_MOVING_SPHERE: return d.n_moving_spheres;
            case RT_KIND_RECT: return d.n_rects;
            case RT_KIND_BOX: return d.n_boxes;
            case RT_KIND_TRIANGLE: return d.n_triangles;
            case RT_KIND_RING: return d.n_rings;
            case RT_KIND_MEDIUM: return d.n_media;
            case RT_KIND_TRANSLATE: case RT_KIND_ROTATE_Y: case RT_KIND_ZOOM: return d.n_xforms;
            case RT_KIND_LIST: return d.n_lists;
            default: return 0;
        }
    }
    void check_ref(uint32_t ref, const char *where) const {
        uint32_t kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
        RT_REQUIRE(kind < RT_KIND_COUNT, RT_ERR_INVALID, std::string(where) + ": ref with unknown kind");
        RT_REQUIRE(idx < pool_size(kind), RT_ERR_INVALID, std::string(where) + ": ref index out of range");
        if (kind >= RT_KIND_TRANSLATE && kind <= RT_KIND_ZOOM)
            RT_REQUIRE(d.xforms[idx].kind == kind, RT_ERR_INVALID, std::string(where) + ": mover ref kind does not match its record");
    }
    // Node boxes as the reference builds them (min / max of real coordinates): finite and ordered.
    bool boxes_plain() const {
        for (uint32_t i = 0; i < d.n_nodes; i++)
            for (int a = 0; a < 3; a++) {
                double lo = d.nodes[i].bmin[a], hi = d.nodes[i].bmax[a];
                if (!(std::isfinite(lo) && std::isfinite(hi) && lo <= hi)) return false;
            }
        return true;
    }
    void check_mat(uint32_t mat, const char *where) const {
        RT_REQUIRE(mat < d.n_materials, RT_ERR_INVALID, std::string(where) + ": material index out of range");
    }

    void check_pools() const {
#define RT_NONNULL(n, p) RT_REQUIRE(d.n == 0 || d.p != nullptr, RT_ERR_INVALID, #p " is null but " #n " > 0")
        RT_NONNULL(n_nodes, nodes); RT_NONNULL(n_spheres, spheres); RT_NONNULL(n_moving_spheres, moving_spheres);
        RT_NONNULL(n_rects, rects); RT_NONNULL(n_boxes, boxes); RT_NONNULL(n_triangles, triangles); RT_NONNULL(n_rings, rings);
        RT_NONNULL(n_media, media); RT_NONNULL(n_xforms, xforms); RT_NONNULL(n_lists, lists); RT_NONNULL(n_list_items, list_items);
        RT_NONNULL(n_lights, lights); RT_NONNULL(n_materials, materials); RT_NONNULL(n_textures, textures);
        RT_NONNULL(n_images, images); RT_NONNULL(image_data_bytes, image_data); RT_NONNULL(n_perlins, perlins);
#undef RT_NONNULL
        for (uint32_t i = 0; i < d.n_nodes; i++) { check_ref(d.nodes[i].left, "node.left"); check_ref(d.nodes[i].right, "node.right"); }
        for (uint32_t i = 0; i < d.n_spheres; i++) check_mat(d.spheres[i].mat, "sphere");
        for (uint32_t i = 0; i < d.n_moving_spheres; i++) check_mat(d.moving_spheres[i].mat, "moving_sphere");
        for (uint32_t i = 0; i < d.n_rects; i++) { check_mat(d.rects[i].mat, "rect"); RT_REQUIRE(d.rects[i].axis <= RT_RECT_YZ, RT_ERR_INVALID, "rect: bad axis"); }
        for (uint32_t i = 0; i < d.n_boxes; i++) check_mat(d.boxes[i].mat, "box");
        for (uint32_t i = 0; i < d.n_triangles; i++) check_mat(d.triangles[i].mat, "triangle");
        for (uint32_t i = 0; i < d.n_rings; i++) check_mat(d.rings[i].mat, "ring");
        for (uint32_t i = 0; i < d.n_media; i++) {
            check_mat(d.media[i].mat, "medium");
            RT_REQUIRE(d.materials[d.media[i].mat].kind == RT_MAT_ISOTROPIC, RT_ERR_INVALID, "medium: phase function must be Isotropic");
            check_ref(d.media[i].boundary, "medium.boundary");
            // A boundary that is one primitive under movers is what the megakernel engine handles;
            // anything else (a box of boxes, a BVH, a list) needs the wavefront engine's sub-queries.
            uint32_t ref = d.media[i].boundary;
            int lvl = 0;
            bool simple = true;
            while (RT_REF_KIND(ref) >= RT_KIND_TRANSLATE && RT_REF_KIND(ref) <= RT_KIND_ZOOM) {
                if (++lvl > RT_MAX_XFORM_DEPTH) { simple = false; break; }
                ref = d.xforms[RT_REF_INDEX(ref)].child;
                check_ref(ref, "medium.boundary chain");
            }
            uint32_t k = RT_REF_KIND(ref);
            if (!(k >= RT_KIND_SPHERE && k <= RT_KIND_RING)) simple = false;
            if (!simple) general_boundaries = true;
        }
        for (uint32_t i = 0; i < d.n_xforms; i++) {
            uint32_t k = d.xforms[i].kind;
            RT_REQUIRE(k >= RT_KIND_TRANSLATE && k <= RT_KIND_ZOOM, RT_ERR_INVALID, "xform: bad kind");
            check_ref(d.xforms[i].child, "xform.child");
        }
        for (uint32_t i = 0; i < d.n_lists; i++)
            RT_REQUIRE((uint64_t)d.lists[i].first + d.lists[i].count <= d.n_list_items, RT_ERR_INVALID, "list: items out of range");
        for (uint32_t i = 0; i < d.n_list_items; i++) check_ref(d.list_items[i], "list item");
        for (uint32_t i = 0; i < d.n_lights; i++) check_ref(d.lights[i], "light");
        for (uint32_t i = 0; i < d.n_materials; i++) {
            const rt_material &m = d.materials[i];
            RT_REQUIRE(m.kind <= RT_MAT_ISOTROPIC, RT_ERR_INVALID, "material: bad kind");
            if (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC)
                RT_REQUIRE(m.tex < d.n_textures, RT_ERR_INVALID, "material: texture index out of range");
        }
        for (uint32_t i = 0; i < d.n_textures; i++) {
            const rt_texture &t = d.textures[i];
            RT_REQUIRE(t.kind <= RT_TEX_IMAGE, RT_ERR_INVALID, "texture: bad kind");
            if (t.kind == RT_TEX_CHECKER) RT_REQUIRE(t.a < d.n_textures && t.b < d.n_textures, RT_ERR_INVALID, "checker: child out of range");
            if (t.kind == RT_TEX_NOISE) RT_REQUIRE(t.a < d.n_perlins, RT_ERR_INVALID, "noise: perlin index out of range");
            if (t.kind == RT_TEX_IMAGE) RT_REQUIRE(t.a < d.n_images, RT_ERR_INVALID, "image texture: image index out of range");
        }
        for (uint32_t i = 0; i < d.n_images; i++) {
            const rt_image &im = d.images[i];
            RT_REQUIRE(im.offset + (uint64_t)im.width * im.height * 3 <= d.image_data_bytes, RT_ERR_INVALID, "image: data out of range");
        }
        for (uint32_t i = 0; i < d.n_perlins; i++)
            for (int k = 0; k < 256; k++) {
                const rt_perlin &p = d.perlins[i];
                RT_REQUIRE((uint32_t)p.perm_x[k] < 256 && (uint32_t)p.perm_y[k] < 256 && (uint32_t)p.perm_z[k] < 256, RT_ERR_INVALID, "perlin: permutation entry out of range");
            }
    }

    // Stack entries trace<> needs while processing `ref` (its own slot included), and
    // the deepest nesting of movers below it. Cycles are rejected.
    void need(uint32_t ref, int depth, int32_t &out_need, int32_t &out_xdepth) {
        RT_REQUIRE(depth < 4096, RT_ERR_UNSUPPORTED, "scene graph too deep");
        uint32_t kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
        if (kind == RT_KIND_NODE) {
            RT_REQUIRE(!(ref & RT_REF_FLIP), RT_ERR_UNSUPPORTED, "FlipFace directly on a BvhNode ref: push the flip down to the leaves");
            RT_REQUIRE(node_need[idx] != -2, RT_ERR_INVALID, "cycle in the BVH");
            if (node_need[idx] >= 0) { out_need = node_need[idx]; out_xdepth = node_xdepth[idx]; return; }
            node_need[idx] = -2;
            int32_t nl, xl, nr, xr;
            need(d.nodes[idx].left, depth + 1, nl, xl);
            if (d.nodes[idx].right == d.nodes[idx].left) { nr = nl; xr = xl; }
            else need(d.nodes[idx].right, depth + 1, nr, xr);
            out_need = std::max(1 + nl, nr);
            out_xdepth = std::max(xl, xr);
            node_need[idx] = out_need;
            node_xdepth[idx] = out_xdepth;
            return;
        }
        if (kind >= RT_KIND_TRANSLATE && kind <= RT_KIND_ZOOM) {
            int32_t nc, xc;
            need(d.xforms[idx].child, depth + 1, nc, xc);
            out_need = 1 + nc;
            out_xdepth = 1 + xc;
            return;
        }
        if (kind == RT_KIND_LIST) {
            RT_REQUIRE(!(ref & RT_REF_FLIP), RT_ERR_UNSUPPORTED, "FlipFace directly on a HittableList ref: push the flip down to the items");
            const rt_list &l = d.lists[idx];
            int32_t best = std::max<int32_t>(1, (int32_t)l.count), bx = 0;
            for (uint32_t i = 0; i < l.count; i++) {
                int32_t ni, xi;
                need(d.list_items[l.first + i], depth + 1, ni, xi);
                best = std::max(best, (int32_t)(l.count - 1 - i) + ni);
                bx = std::max(bx, xi);
            }
            out_need = best;
            out_xdepth = bx;
            return;
        }
        if (kind == RT_KIND_MEDIUM) {
            // the medium's own slot becomes the sub-query sentinel while its boundary is traversed
            RT_REQUIRE(!in_boundary, RT_ERR_UNSUPPORTED, "a medium inside another medium's boundary");
            in_boundary = true;
            int32_t nb, xb;
            need(d.media[idx].boundary, depth + 1, nb, xb);
            in_boundary = false;
            out_need = 1 + nb;
            out_xdepth = xb;
            return;
        }
        out_need = 1;
        out_xdepth = 0;
    }
};

// The scene's buffers, pools and streams live on the device that was current at rt_scene_create. Every entry
// point that touches them makes that device current for the call and puts the caller's back afterwards, so one
// process may hold scenes on several GPUs (one host thread per scene and stream) whatever its current device is.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int want) {
        RT_HIP(hipGetDevice(&prev));
        if (prev != want) { RT_HIP(hipSetDevice(want)); switched = true; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// New index of every BVH node in the device copy: breadth-first from the root, through movers, lists and medium
// boundaries. The traversal kernels keep the FIRST records of the node table in LDS (pt_wavefront.hip); numbered this
// way those are the top levels of the BVHs — the nodes every ray goes through. (The flattener emits children before
// parents; the order of the records means nothing to the results.)
std::vector<uint32_t> breadth_first_nodes(const rt_scene_desc &d) {
    std::vector<uint32_t> new_of(d.n_nodes, 0xFFFFFFFFu), queue;
    std::vector<char> seen_x(d.n_xforms, 0), seen_l(d.n_lists, 0), seen_m(d.n_media, 0);
    uint32_t next = 0;
    queue.push_back(d.root);
    for (size_t h = 0; h < queue.size(); h++) {
        const uint32_t ref = queue[h], kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
        if (kind == RT_KIND_NODE) {
            if (new_of[idx] != 0xFFFFFFFFu) continue;
            new_of[idx] = next++;
            queue.push_back(d.nodes[idx].left);
            if (d.nodes[idx].right != d.nodes[idx].left) queue.push_back(d.nodes[idx].right);
        } else if (kind >= RT_KIND_TRANSLATE && kind <= RT_KIND_ZOOM) {
            if (!seen_x[idx]) { seen_x[idx] = 1; queue.push_back(d.xforms[idx].child); }
        } else if (kind == RT_KIND_LIST) {
            if (!seen_l[idx]) { seen_l[idx] = 1; for (uint32_t i = 0; i < d.lists[idx].count; i++) queue.push_back(d.list_items[d.lists[idx].first + i]); }
        } else if (kind == RT_KIND_MEDIUM) {
            if (!seen_m[idx]) { seen_m[idx] = 1; queue.push_back(d.media[idx].boundary); }
        }
    }
    for (uint32_t i = 0; i < d.n_nodes; i++)
        if (new_of[i] == 0xFFFFFFFFu) new_of[i] = next++;            // (unreachable nodes keep a place behind the others)
    return new_of;
}

template <class T>
T *upload(const T *src, uint64_t n, std::vector<void *> &owned) {
    // Never hand the kernels a null pool: an empty pool gets one zeroed element.
    // (128 bytes of zeroed slack behind every pool: the shading kernel fetches a fixed 80 bytes from the winning
    // primitive's record whatever its kind, the last record of a pool included)
    uint64_t bytes = (n ? n : 1) * sizeof(T) + 128;
    void *p = nullptr;
    RT_HIP(hipMalloc(&p, bytes));
    owned.push_back(p);
    RT_HIP(hipMemset(p, 0, bytes));
    if (n) RT_HIP(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return (T *)p;
}

struct Workspace {
    unsigned long long *work_counter = nullptr;
    StatsDev *stats = nullptr;
    double *partial = nullptr;
    uint64_t partial_bytes = 0;
    double *tape = nullptr;
    uint64_t tape_bytes = 0;
    // wavefront engine
    WfPool pool{};
    std::vector<void *> pool_owned;
    uint32_t pool_slots = 0, pool_depth = 0;
    unsigned long long *pool_dbg = nullptr;
    SceneDev *d_scene = nullptr;
    RenderArgs *d_args = nullptr;
    WfPool *d_pool = nullptr;
    WfStreams gs{};                   // group streams / events / pinned words (created on first use)
    uint32_t iterations = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    rt_stats *pending = nullptr;      // host stats to fill at rt_render_wait
    bool pending_counters = false;
    KernelTimes kt;                   // RT_FLAG_KERNEL_TIMES
    bool used_kt = false;
    uint32_t used_chunk = 0, used_passes = 0;
    uint64_t used_slots = 0;
    uint32_t *rows_max = nullptr;     // device word: largest row id of the call being checked
    uint32_t *h_rows_max = nullptr;   // ... and the pinned word it is copied to
    unsigned long long *d_limit = nullptr;   // ring mode: RenderArgs::claim_limit
    uint64_t used_partial_bytes = 0;
    // RT_FLAG_ASYNC: the host thread that drives the passes of the call in flight on this (scene, stream), and what it ended with
    // (read by rt_render_wait after the join).
    std::thread async_worker;
    int async_rc = RT_OK;
    std::string async_err;
};

} // namespace

struct rt_scene {
    SceneDev dev{};
    std::vector<void *> owned;
    uint32_t stack_need = 1;
    unsigned features = 7;
    bool general_boundaries = false;
    bool boxes_plain = false;         // every node box finite with min <= max: the short node step applies
    double split[3] = {0.0, 0.0, 0.0};         // centre of the root's box (list ordering)
    uint32_t node_quorum = 18u | (1u << 8) | (2u << 12) | ((uint32_t)(8 * 4096 / kSlotsPerBlock > 0 ? 8 * 4096 / kSlotsPerBlock : 1) << 16) | (2u << 20) | (0u << 24);   // fast-path quorum 18 lanes; one extra sphere test per turn; tail factor 2; pool of 8 segments per resident trace workgroup (4 per CU: 8192 segments = 33.5 M slots); list classes of 4 node steps; groups of segments: the library's choice (0)
    uint32_t vote_weights = 0;                 // 0: the engine's own default (kWfVoteWeights / kMegaVoteWeights, pt_device.h)
    int engine = 1;                   // 0 = megakernel, 1 = wavefront (shade / trace passes)
    unsigned long long census_rounds[9] = {}, census_lanes[9] = {};   // of the last counter run
    int max_pool_blocks = 0;          // 0 = 5 x CUs x segments per trace workgroup
    int partial_ring = 0;             // planes of the partial-sum ring: 0 automatic, -1 never, > 0 this many (rt_debug_set_partial_ring)
    int partial_ring_group = 0;       // largest sample group of the ring (0: 25); env RT2022_RING_GROUP at rt_scene_create (A/B)
    uint64_t ring_threshold_bytes = 64ull << 30;   // partial sums above this switch the ring on: 40 % of the device's memory (rt_scene_create)
    double pass_timing[5] = {};       // of the last render with tuning bit 29 (rt_debug_pass_timing)
    int device = 0;
    int n_cus = 0;                    // compute units of `device`
    std::mutex mu;
    std::map<hipStream_t, Workspace> ws;
};

namespace {

Workspace &workspace_for(rt_scene *sc, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(sc->mu);
    Workspace &w = sc->ws[stream];
    if (!w.work_counter) {
        RT_HIP(hipMalloc((void **)&w.work_counter, sizeof(unsigned long long)));
        RT_HIP(hipMalloc((void **)&w.stats, sizeof(StatsDev)));
        RT_HIP(hipEventCreate(&w.ev0));
        RT_HIP(hipEventCreate(&w.ev1));
    }
    return w;
}

void check_params(const rt_scene *scene, const rt_camera *cam, const rt_params *p) {
    RT_REQUIRE(scene && cam && p, RT_ERR_INVALID, "null argument");
    RT_REQUIRE(p->width > 0 && p->height > 0 && p->n_frames > 0, RT_ERR_INVALID, "empty image");
    RT_REQUIRE(cam->time0 < cam->time1, RT_ERR_INVALID, "camera time0 >= time1 (gen_range panics in the reference, camera.rs:71)");
    RT_REQUIRE(p->n_rows == 0 || p->row_ids, RT_ERR_INVALID, "row_ids is null");
    RT_REQUIRE((uint64_t)p->height * p->n_frames <= 0xFFFFFFFFull, RT_ERR_INVALID, "height * n_frames overflows a row id");
}

template <class T>
T *pool_alloc(Workspace &w, uint64_t count) {
    void *p = nullptr;
    RT_HIP(hipMalloc(&p, (count ? count : 1) * sizeof(T)));
    w.pool_owned.push_back(p);
    return (T *)p;
}

// (Re)allocate the wavefront pool for `blocks` workgroups and `depth` tape records.
void ensure_pool(Workspace &w, uint32_t blocks, uint32_t depth, hipStream_t stream) {
    uint32_t slots = blocks * (uint32_t)kSlotsPerBlock;
    if (depth == 0) depth = 1;
    if (w.d_scene == nullptr) {
        RT_HIP(hipMalloc((void **)&w.d_scene, sizeof(SceneDev)));
        RT_HIP(hipMalloc((void **)&w.d_args, sizeof(RenderArgs)));
        RT_HIP(hipMalloc((void **)&w.d_pool, sizeof(WfPool)));
        RT_HIP(hipHostMalloc((void **)&w.gs.h_active, 2 * kMaxGroups * sizeof(uint32_t)));
        RT_HIP(hipHostMalloc((void **)&w.gs.h_work, 2 * kMaxGroups * sizeof(unsigned long long)));
        RT_HIP(hipHostMalloc((void **)&w.gs.h_oldest, 2 * kMaxGroups * sizeof(unsigned long long)));
        RT_HIP(hipMalloc((void **)&w.d_limit, sizeof(unsigned long long)));
        for (int g = 0; g < kMaxGroups; g++) {
            RT_HIP(hipStreamCreateWithFlags(&w.gs.stream[g], hipStreamNonBlocking));
            for (int b = 0; b < 2; b++) RT_HIP(hipEventCreateWithFlags(&w.gs.ev[g][b], hipEventDisableTiming));
        }
    }
    if (slots <= w.pool_slots && depth <= w.pool_depth) { w.pool.n_blocks = blocks; w.pool.n_slots = w.pool_slots; return; }
    RT_HIP(hipStreamSynchronize(stream));
    for (int g = 0; g < kMaxGroups; g++) RT_HIP(hipStreamSynchronize(w.gs.stream[g]));    // (passes of an earlier call that failed half-way)
    for (void *p : w.pool_owned) RT_HIP(hipFree(p));
    w.pool_owned.clear();
    w.pool_slots = 0; w.pool_depth = 0;
    if (slots < w.pool.n_slots) slots = w.pool.n_slots;
    uint64_t tape_bytes = (uint64_t)slots * depth * 4 * sizeof(double);
    RT_REQUIRE(tape_bytes <= (64ull << 30), RT_ERR_UNSUPPORTED, "max_depth too large for the bounce tape");
    uint64_t P = slots;
    WfPool &q = w.pool;
    q.n_slots = slots;
    q.n_blocks = blocks;
    q.kind = pool_alloc<uint8_t>(w, P);
    q.ray = pool_alloc<double>(w, kRecDoubles * P);                       // the 128-byte slot records: ray | hit | state
    q.hit = reinterpret_cast<uint32_t *>(q.ray) + 16;
    q.state = reinterpret_cast<uint32_t *>(q.ray) + 24;
    q.pixel_sum = pool_alloc<double>(w, 4 * P);
    q.tape = pool_alloc<double>(w, (uint64_t)depth * 4 * P);
    q.tape_cap = depth;
    q.list = pool_alloc<uint16_t>(w, P);
    q.list_n = pool_alloc<uint32_t>(w, P / (uint64_t)kSlotsPerBlock);
    q.n_active = pool_alloc<uint32_t>(w, 2 * kMaxGroups);
    q.next_chunk = pool_alloc<uint32_t>(w, kMaxGroups);
    q.max_list = pool_alloc<uint32_t>(w, 2 * kMaxGroups);
    q.fault = pool_alloc<uint32_t>(w, 1);
    q.oldest = pool_alloc<unsigned long long>(w, 2);
    q.starved_n = pool_alloc<uint32_t>(w, P / (uint64_t)kSlotsPerBlock);
    w.pool_dbg = pool_alloc<unsigned long long>(w, 8 + 2 * 65536);
    w.pool_slots = slots;
    w.pool_depth = depth;
}

// Largest of n device-resident row ids (rt_render_device's twin of rt_render's host-side range check).
__global__ void __launch_bounds__(256) row_ids_max_kernel(const uint32_t *rows, uint32_t n, uint32_t *out) {
    uint32_t m = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = rows[i] > m ? rows[i] : m;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { uint32_t o = (uint32_t)__shfl_xor((int)m, d); m = o > m ? o : m; }
    if ((threadIdx.x & 63u) == 0 && m) atomicMax(out, m);
}
// Two halves, so that the check costs a render no synchronisation of its own: the reduction and the copy of
// its answer to a pinned word are enqueued with the call's other preparations, and the answer is looked at behind the
// synchronisation the call makes anyway before its first pass. (A row id out of range cannot fault a kernel — it only
// mis-keys a frame — so nothing is lost by finding out a moment later.)
void begin_check_device_rows(Workspace &w, const rt_params *p, hipStream_t stream) {
    if (p->n_rows == 0) return;
    if (!w.rows_max) {
        RT_HIP(hipMalloc((void **)&w.rows_max, sizeof(uint32_t)));
        RT_HIP(hipHostMalloc((void **)&w.h_rows_max, sizeof(uint32_t)));
    }
    RT_HIP(hipMemsetAsync(w.rows_max, 0, sizeof(uint32_t), stream));
    uint32_t blocks = (p->n_rows + 255u) / 256u;
    hipLaunchKernelGGL(row_ids_max_kernel, dim3(blocks > 64 ? 64 : blocks), dim3(256), 0, stream, p->row_ids, p->n_rows, w.rows_max);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(w.h_rows_max, w.rows_max, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
}
void end_check_device_rows(Workspace &w, const rt_params *p) {      // (after a synchronisation of the stream)
    if (p->n_rows == 0) return;
    RT_REQUIRE((uint64_t)*w.h_rows_max < (uint64_t)p->height * p->n_frames, RT_ERR_INVALID, "rt_render_device: row id out of range");
}

// Enqueue one render on `stream`; row ids and output are device pointers.
void enqueue(rt_scene *sc, const rt_camera *cam, const rt_params *p, const uint32_t *d_rows, double *d_out,
             hipStream_t stream, rt_stats *stats, bool check_rows = false /* the row ids came from the caller's HBM: range-check them */) {
    Workspace &w = workspace_for(sc, stream);
    RenderArgs a{};
    a.cam = *cam;
    a.width = p->width; a.height = p->height; a.spp = p->spp; a.max_depth = p->max_depth;
    a.n_frames = p->n_frames; a.n_rows = p->n_rows;
    uint32_t chunk = (p->spp_chunk == 0 || p->spp_chunk > p->spp) ? p->spp : p->spp_chunk;
    if (chunk == 0) chunk = 1;
    a.chunk = chunk;
    a.n_chunks = p->spp == 0 ? 1 : (p->spp + chunk - 1) / chunk;
    std::memcpy(a.background, p->background, sizeof a.background);
    a.t_min = p->t_min;
    std::memcpy(a.split, sc->split, sizeof a.split);
    a.seed = p->seed;
    a.n_pixels = (uint64_t)p->n_rows * p->width;
    a.n_items = a.n_pixels * a.n_chunks;
    a.row_ids = d_rows;
    bool counters = stats && (p->flags & RT_FLAG_COUNTERS);
    const bool want_kt = stats && (p->flags & RT_FLAG_KERNEL_TIMES) && sc->engine == 1;
    // Ring of partial-sum planes (pt_device.h, RenderArgs::ring): one-sample work items of the wavefront engine only. Automatic
    // when all spp planes would take more than 40 % of the device's memory (115 GB on an MI355X: C5's 99.5 GB stay below it —
    // the ring's work-item order costs its traversal 10 %, the headline's 1.4 %: profiles/r3j_ring.log): then at most 24 GiB
    // of planes; rt_debug_set_partial_ring forces a size (tests: down to one plane) or switches it off.
    a.ring = 0; a.ring_group = 1;
    if (sc->engine == 1 && a.chunk == 1 && a.n_chunks > 1 && a.n_pixels > 0 && sc->partial_ring >= 0) {
        uint32_t want = 0;
        const uint64_t plane = a.n_pixels * 3 * sizeof(double);
        if (sc->partial_ring > 0) want = (uint32_t)sc->partial_ring;
        else if (a.n_items * 3 * sizeof(double) > sc->ring_threshold_bytes) want = (uint32_t)std::max<uint64_t>(8, (24ull << 30) / plane);
        if (want && want < a.n_chunks) {
            // samples are taken in groups: the largest divisor of spp up to 25 (and up to a quarter of the ring, so that it holds
            // a few groups); the ring is a whole number of groups. (A group's planes are free again only when its last straggler
            // has ended, ~60 passes after its first claim: the ring must hold what is claimed meanwhile — measured on C5: five
            // groups of 50 stall every pool fill, twenty of 25 never.)
            uint32_t grp = 1;
            const uint32_t group_max = sc->partial_ring_group > 0 ? (uint32_t)sc->partial_ring_group : 25u;
            for (uint32_t d = 1; d <= group_max && d * 4u <= std::max(want, 4u); d++) if (a.n_chunks % d == 0) grp = d;
            if (grp > want) grp = 1;
            a.ring_group = grp;
            a.ring = want / grp * grp;
            if (a.ring == 0 || a.ring >= a.n_chunks) { a.ring = 0; a.ring_group = 1; }
        }
    }
    if (a.n_chunks > 1) {
        uint64_t bytes = (a.ring ? (uint64_t)a.ring * a.n_pixels : a.n_items) * 3 * sizeof(double);
        w.used_partial_bytes = bytes;
        if (bytes > w.partial_bytes) {
            RT_HIP(hipStreamSynchronize(stream));
            if (w.partial) RT_HIP(hipFree(w.partial));
            w.partial = nullptr; w.partial_bytes = 0;
            RT_HIP(hipMalloc((void **)&w.partial, bytes));
            w.partial_bytes = bytes;
        }
        a.partial = w.partial;
    } else {
        a.partial = d_out;
        w.used_partial_bytes = 0;
    }
    // bit 31: boxes are plain (see wf_trace's fast path); bit 30 of the tuning word forces the literal step
    a.node_quorum = (sc->node_quorum & 0x7FFFFFFFu) | ((sc->boxes_plain && !(sc->node_quorum & (1u << 30))) ? (1u << 31) : 0u);
    a.vote_weights = sc->vote_weights ? sc->vote_weights : (sc->engine == 1 ? kWfVoteWeights : kMegaVoteWeights);
    a.work_counter = w.work_counter;
    a.stats = counters ? w.stats : nullptr;
    if (sc->engine == 1) {
        // Wavefront engine: pool of path slots, shade / trace passes until it drains.
        // Pool = segments of 4096 path slots (one shade workgroup each). The trace pass is a persistent grid of
        // kTraceBlocksPerCU workgroups per CU that draws on all segments' ray lists; the pool holds `segs` segments
        // per such workgroup (default 8: 33.5 M slots, ~62 GB with a depth-50 tape — measured optimum of 2.5-10 K
        // segments on the headline scene; sized for 288 GB of HBM).
        uint32_t segs = (sc->node_quorum >> 16) & 0xFu;
        if (segs < 1) segs = 1;
        if (segs > 8) segs = 8;
        uint32_t max_blocks = sc->max_pool_blocks > 0 ? (uint32_t)sc->max_pool_blocks : (uint32_t)kTraceBlocksPerCU * (uint32_t)sc->n_cus * segs;
        // Use every workgroup slot of the chip even for small jobs (64 paths per workgroup at least).
        uint64_t want = (a.n_items + 63) / 64;
        uint32_t blocks = (uint32_t)(want < 1 ? 1 : (want > max_blocks ? max_blocks : want));
        // A job of about as many paths as the pool has slots is better served by half the pool: every path starts in the first pass
        // either way, the passes are as many (a path lives its dozen bounces), and each shade pass sweeps half the segments
        // (book-1 final 400x225x100, four frames per call — 36 M paths for 33.5 M slots: +11 %, profiles/r3zh_segs.log). Two paths per
        // slot at least, for jobs large enough to fill the chip anyway; larger jobs keep the whole pool (they lose with a smaller one).
        {
            const uint64_t two_per_slot = a.n_items / (2ull * (uint64_t)kSlotsPerBlock);
            const uint64_t floor_blocks = (uint64_t)kTraceBlocksPerCU * (uint64_t)sc->n_cus * 2ull;      // (never below two segments per resident traversal workgroup)
            if (sc->max_pool_blocks <= 0 && two_per_slot < blocks && blocks > floor_blocks)
                blocks = (uint32_t)(two_per_slot > floor_blocks ? two_per_slot : floor_blocks);
        }
        // (deep paths: keep the bounce tape under 56 GB by taking fewer segments)
        const uint64_t tape_per_block = (uint64_t)kSlotsPerBlock * (p->max_depth ? p->max_depth : 1) * 4 * sizeof(double);
        while (blocks > segs && (uint64_t)blocks * tape_per_block > (56ull << 30)) blocks -= segs;
        // (and never plan for more than 60 % of the memory that is free right now: other scenes, other users of the GPU)
        if ((uint64_t)blocks * kSlotsPerBlock > w.pool_slots) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const uint64_t per_block = (uint64_t)kSlotsPerBlock * 176 + tape_per_block;      // records + tape, bytes
                const uint64_t budget = (uint64_t)((double)free_b * 0.6) + (uint64_t)w.pool_slots / kSlotsPerBlock * per_block;
                while (blocks > segs && (uint64_t)blocks * per_block > budget) blocks -= segs;
            }
        }
        if (blocks < segs) segs = blocks;
        blocks = blocks / segs * segs;
        ensure_pool(w, blocks, p->max_depth, stream);
        a.claim_limit = w.d_limit;
        w.pool.segs = segs;
        w.pool.n_cus = (uint32_t)sc->n_cus;
        const bool timing = (sc->node_quorum & (1u << 29)) != 0;
        // Groups of pool segments passing independently, each on a stream of its own: one group's shade pass then runs beside another's
        // traversal pass and fills what its stragglers leave idle. 0 in the tuning word = the library's choice: two — measured
        // (profiles/r3ze_groups.log, bench.py --groups): 1e5 random spheres +18 %, Cornell box +3 %, random spheres +1 %, book-2 final
        // +0.4 % — except for meshes (wwscene: -3 % at two, -7 % at three), which keep one.
        w.gs.n = (int)((sc->node_quorum >> 24) & 0xFu);
        if (w.gs.n < 1) w.gs.n = (sc->features & kFeatMisc) ? 1 : 2;
        if (w.gs.n > kMaxGroups) w.gs.n = kMaxGroups;
        w.pool.dbg = timing ? w.pool_dbg : nullptr;
#if defined(RT2022_SHADE_PROBE) || defined(RT2022_TRACE_PROBE)
        w.pool.dbg = w.pool_dbg;                                  // (diagnostic builds: the section clocks of the shade / traversal kernels)
        RT_HIP(hipMemsetAsync(w.pool_dbg + 64, 0, 48 * sizeof(unsigned long long), stream));
#endif
        if (timing) for (double &t : sc->pass_timing) t = 0.0;
        a.tape = nullptr;
        RT_HIP(hipMemsetAsync(w.work_counter, 0, sizeof(unsigned long long), stream));
        if (counters) RT_HIP(hipMemsetAsync(w.stats, 0, sizeof(StatsDev), stream));
        RT_HIP(hipMemcpyAsync(w.d_args, &a, sizeof(RenderArgs), hipMemcpyHostToDevice, stream));
        if (check_rows) begin_check_device_rows(w, p, stream);
        RT_HIP(hipStreamSynchronize(stream));      // the three structs above live on this thread's stack
        if (check_rows) end_check_device_rows(w, p);
        RT_HIP(hipEventRecord(w.ev0, stream));
        Progress prog;
        prog.cb = p->progress_cb; prog.user = p->progress_user;
        prog.total = a.n_pixels * p->spp; prog.per_item = a.chunk;
        RingCtl ring;
        ring.planes = a.ring; ring.out = d_out; ring.d_limit = w.d_limit;
        // (watchdog of the ring's pass loop: a frame needs about items / slots pool fills of at most max_depth + 1 passes each)
        // (... plus one drain per ring-full of planes when the ring is small)
        ring.max_passes = (uint32_t)std::min<uint64_t>(1u << 26, 64 + 8 * (a.n_items / ((uint64_t)w.pool.n_blocks * kSlotsPerBlock) + 2 + (a.ring ? a.n_chunks / a.ring : 0)) *
                                                                     ((uint64_t)p->max_depth + 2));
        if (a.n_items > 0) {
            uint32_t fault = 0;
            RT_HIP(launch_render_wavefront(sc->dev, a, w.d_args, w.pool, sc->stack_need, sc->features, counters, w.gs, stream, &w.iterations,
                                           timing ? sc->pass_timing : nullptr, &fault, want_kt ? &w.kt : nullptr, &prog, a.ring ? &ring : nullptr));
            RT_REQUIRE(fault == 0, RT_ERR_DEVICE, "wavefront engine: a path slot reached the shade pass without having been traced (internal error; the frame is incomplete)");
            if (a.n_chunks > 1 && !a.ring) RT_HIP(launch_chunk_sum(a.partial, d_out, a.n_pixels * 3, a.n_chunks, stream));
        }
        RT_HIP(hipEventRecord(w.ev1, stream));
        w.pending = stats;
        w.pending_counters = counters;
        w.used_chunk = a.chunk; w.used_passes = w.iterations; w.used_slots = (uint64_t)w.pool.n_blocks * kSlotsPerBlock;
        w.used_kt = want_kt && a.n_items > 0;
        // (every pass of the frame has been issued and observed: the render is complete up to the chunk sums queued behind it)
        if (prog.cb) prog.cb(prog.user, 0u, prog.total, prog.total);
        return;
    }
    // Megakernel engine. Bounce tape: max_depth records of 4 doubles for every lane of the persistent grid.
    int blocks = render_grid_blocks(sc->stack_need, counters);
    uint64_t want_blocks = (a.n_items + kBlock - 1) / kBlock;
    if ((uint64_t)blocks > want_blocks) blocks = (int)(want_blocks ? want_blocks : 1);
    uint64_t tape_bytes = (uint64_t)blocks * kBlock * (uint64_t)(p->max_depth ? p->max_depth : 1) * 4 * sizeof(double);
    RT_REQUIRE(tape_bytes <= (32ull << 30), RT_ERR_UNSUPPORTED, "max_depth too large for the bounce tape");
    if (tape_bytes > w.tape_bytes) {
        RT_HIP(hipStreamSynchronize(stream));
        if (w.tape) RT_HIP(hipFree(w.tape));
        w.tape = nullptr; w.tape_bytes = 0;
        RT_HIP(hipMalloc((void **)&w.tape, tape_bytes));
        w.tape_bytes = tape_bytes;
    }
    a.tape = w.tape;
    if (check_rows) {                              // (the A/B engine enqueues without a synchronisation of its own)
        begin_check_device_rows(w, p, stream);
        RT_HIP(hipStreamSynchronize(stream));
        end_check_device_rows(w, p);
    }
    RT_HIP(hipMemsetAsync(w.work_counter, 0, sizeof(unsigned long long), stream));
    if (counters) RT_HIP(hipMemsetAsync(w.stats, 0, sizeof(StatsDev), stream));
    RT_HIP(hipEventRecord(w.ev0, stream));
    if (a.n_items > 0) {
        RT_HIP(launch_render(sc->dev, a, sc->stack_need, counters, blocks, stream));
        if (a.n_chunks > 1) RT_HIP(launch_chunk_sum(a.partial, d_out, a.n_pixels * 3, a.n_chunks, stream));
    }
    RT_HIP(hipEventRecord(w.ev1, stream));
    w.pending = stats;
    w.pending_counters = counters;
    w.used_chunk = a.chunk; w.used_passes = 0; w.used_slots = 0; w.used_kt = false;
    if (p->progress_cb) {                          // (the A/B engine is one launch: nothing to report in between)
        RT_HIP(hipStreamSynchronize(stream));
        p->progress_cb(p->progress_user, 0u, a.n_pixels * p->spp, a.n_pixels * p->spp);
    }
}

void finish(rt_scene *sc, hipStream_t stream) {
    Workspace &w = workspace_for(sc, stream);
    RT_HIP(hipStreamSynchronize(stream));
    if (w.pending) {
        rt_stats out;
        std::memset(&out, 0, sizeof out);
        if (w.pending_counters) {
            StatsDev h;
            RT_HIP(hipMemcpy(&h, w.stats, sizeof h, hipMemcpyDeviceToHost));
            out.paths = h.paths; out.rays = h.rays; out.node_visits = h.node_visits;
            for (int k = 0; k < RT_KIND_COUNT; k++) out.prim_tests[k] = h.prim_tests[k];
            out.light_pdf_tests = h.light_pdf_tests; out.rng_draws = h.rng_draws;
            {
                std::lock_guard<std::mutex> lock(sc->mu);
                for (int o = 0; o < 9; o++) { sc->census_rounds[o] = h.op_rounds[o]; sc->census_lanes[o] = h.op_lanes[o]; }
            }
        }
        float ms = 0.f;
        RT_HIP(hipEventElapsedTime(&ms, w.ev0, w.ev1));
        out.ms = (double)ms;
        out.spp_chunk = w.used_chunk; out.passes = w.used_passes; out.pool_slots = w.used_slots;
        out.partial_bytes = w.used_partial_bytes;
        if (w.used_kt) { out.trace_ms = w.kt.trace_ms; out.shade_ms = w.kt.shade_ms; }
        *w.pending = out;
        w.pending = nullptr;
    }
}

} // namespace

extern "C" {

int rt_scene_create(const rt_scene_desc *desc, rt_scene **out) {
    return guarded([&]() -> int {
        RT_REQUIRE(desc && out, RT_ERR_INVALID, "rt_scene_create: null argument");
        RT_REQUIRE(desc->abi_version == RT2022_ABI_VERSION, RT_ERR_INVALID, "rt_scene_create: abi_version mismatch");
        Validator v(*desc);
        v.check_pools();
        v.check_ref(desc->root, "root");
        int32_t need = 1, xdepth = 0;
        v.need(desc->root, 0, need, xdepth);
        RT_REQUIRE(need <= kStackLarge, RT_ERR_UNSUPPORTED, "scene needs a deeper traversal stack than the kernel provides");
        RT_REQUIRE(xdepth <= RT_MAX_XFORM_DEPTH, RT_ERR_UNSUPPORTED, "movers nested deeper than RT_MAX_XFORM_DEPTH");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        RT_REQUIRE(e == hipSuccess && ndev > 0, RT_ERR_DEVICE, "rt_scene_create: no HIP device available (the path has no CPU fallback)");
        rt_scene *sc = new rt_scene();
        if (const char *eg = getenv("RT2022_RING_GROUP")) sc->partial_ring_group = atoi(eg);
        try {
            RT_HIP(hipGetDevice(&sc->device));
            RT_HIP(hipDeviceGetAttribute(&sc->n_cus, hipDeviceAttributeMultiprocessorCount, sc->device));
            RT_REQUIRE(sc->n_cus > 0, RT_ERR_DEVICE, "rt_scene_create: device reports no compute units");
            {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) sc->ring_threshold_bytes = (uint64_t)((double)total_b * 0.4);
            }
            SceneDev &s = sc->dev;
            // Node refs of the device copy follow the breadth-first numbering (breadth_first_nodes).
            const std::vector<uint32_t> new_of = breadth_first_nodes(*desc);
            auto node_ref = [&](uint32_t ref) {
                return RT_REF_KIND(ref) == RT_KIND_NODE ? (ref & ~RT_REF_INDEX_MASK) | new_of[RT_REF_INDEX(ref)] : ref;
            };
            {
                std::vector<rt_bvh_node> nodes(desc->n_nodes);
                for (uint32_t i = 0; i < desc->n_nodes; i++) {
                    rt_bvh_node q = desc->nodes[i];
                    q.left = node_ref(q.left); q.right = node_ref(q.right);
                    // The PUSH REF of the node, for the wavefront traversal kernels (third word of the record's last 16 bytes): what
                    // goes on the stack when the box is hit — the right child, or "nothing" (14 << 27, their REF_EMPTY) for a span-1
                    // node holding the same plain primitive twice (bvh/mod.rs:44-47), whose second test finds the first one's hit again
                    // (counted, not repeated). Media, movers, lists and nodes are really visited twice: they draw from the RNG or recurse.
                    const uint32_t lk = RT_REF_KIND(q.left);
                    q._pad[0] = (q.left == q.right && lk >= RT_KIND_SPHERE && lk <= RT_KIND_RING) ? (14u << RT_REF_KIND_SHIFT) : q.right;
                    nodes[new_of[i]] = q;
                }
                s.nodes = upload(nodes.data(), nodes.size(), sc->owned);
                std::vector<uint32_t> n32((size_t)8 * nodes.size() + 8);         // (never empty: upload of nothing is a null pointer)
                for (size_t i = 0; i < nodes.size(); i++) {
                    const rt_bvh_node &q = nodes[i];
                    for (int ax = 0; ax < 3; ax++) {
                        const float lo = (float)q.bmin[ax], hi = (float)q.bmax[ax];
                        std::memcpy(&n32[8 * i + 2 * ax], &lo, 4);
                        std::memcpy(&n32[8 * i + 2 * ax + 1], &hi, 4);
                    }
                    n32[8 * i + 6] = q.left; n32[8 * i + 7] = q._pad[0];
                }
                s.nodes32 = upload(n32.data(), n32.size(), sc->owned);
            }
            // Primitive pools go up with the slot kind of their material packed above the material index (pt_device.h).
            RT_REQUIRE(desc->n_materials <= kMatIndexMask, RT_ERR_UNSUPPORTED, "more than 2^24 materials");
            auto packed = [&](auto *src, uint64_t n) {
                using T = std::remove_const_t<std::remove_pointer_t<decltype(src)>>;
                std::vector<T> v(src, src + n);
                for (T &q : v) {
                    const rt_material &m = desc->materials[q.mat];
                    uint32_t sk = m.kind == RT_MAT_DIFFUSE_LIGHT ? SK_LIGHT : m.kind == RT_MAT_METAL ? SK_METAL : m.kind == RT_MAT_DIELECTRIC ? SK_DIELECTRIC
                                : m.kind == RT_MAT_ISOTROPIC ? SK_ISOTROPIC : (uint32_t)SK_LAMB_SOLID + desc->textures[m.tex].kind;
                    q.mat |= sk << kMatKindShift;
                }
                return upload(v.data(), n, sc->owned);
            };
            s.spheres = packed(desc->spheres, desc->n_spheres);
            s.moving_spheres = packed(desc->moving_spheres, desc->n_moving_spheres);
            s.rects = packed(desc->rects, desc->n_rects);
            s.boxes = packed(desc->boxes, desc->n_boxes);
            s.triangles = packed(desc->triangles, desc->n_triangles);
            s.rings = packed(desc->rings, desc->n_rings);
            {
                std::vector<rt_medium> media(desc->media, desc->media + desc->n_media);
                for (rt_medium &m : media) m.boundary = node_ref(m.boundary);
                s.media = packed(media.data(), media.size());
            }
            {
                std::vector<MediumDev> md(desc->n_media);
                uint32_t n_sph = 0;
                for (uint32_t i = 0; i < desc->n_media; i++) {
                    const rt_medium &m = desc->media[i];
                    MediumDev &q = md[i];
                    std::memset(&q, 0, sizeof q);
                    q.neg_inv_density = m.neg_inv_density;
                    q.boundary = node_ref(m.boundary);
                    q.mat = m.mat | ((uint32_t)SK_ISOTROPIC << kMatKindShift);
                    if (RT_REF_KIND(m.boundary) == RT_KIND_SPHERE && !(m.boundary & RT_REF_FLIP)) {
                        const rt_sphere &sp = desc->spheres[RT_REF_INDEX(m.boundary)];
                        q.center[0] = sp.center[0]; q.center[1] = sp.center[1]; q.center[2] = sp.center[2]; q.radius = sp.radius;
                        q.sphere_boundary = 1;
                        n_sph++;
                    }
                }
                s.media_dev = upload(md.data(), md.size(), sc->owned);
                s.media_mode = n_sph == 0 ? 0u : n_sph == desc->n_media ? 1u : 2u;
            }
            {
                std::vector<rt_xform> xforms(desc->xforms, desc->xforms + desc->n_xforms);
                for (rt_xform &x : xforms) x.child = node_ref(x.child);
                s.xforms = upload(xforms.data(), xforms.size(), sc->owned);
                std::vector<uint32_t> items(desc->list_items, desc->list_items + desc->n_list_items);
                for (uint32_t &r : items) r = node_ref(r);
                s.list_items = upload(items.data(), items.size(), sc->owned);
            }
            s.lists = upload(desc->lists, desc->n_lists, sc->owned);
            s.lights = upload(desc->lights, desc->n_lights, sc->owned);
            s.materials = upload(desc->materials, desc->n_materials, sc->owned);
            s.textures = upload(desc->textures, desc->n_textures, sc->owned);
            {
                std::vector<MaterialDev> md(desc->n_materials);
                for (uint32_t i = 0; i < desc->n_materials; i++) {
                    const rt_material &m = desc->materials[i];
                    MaterialDev &q = md[i];
                    std::memset(&q, 0, sizeof q);
                    q.tex = m.tex;
                    std::memcpy(q.albedo, m.albedo, sizeof q.albedo);
                    q.param = m.param;
                    if (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC) {
                        const rt_texture &t = desc->textures[m.tex];
                        q.tex_kind = t.kind; q.tex_a = t.a; q.tex_b = t.b; q.tex_scale = t.scale;
                        std::memcpy(q.tex_color, t.color, sizeof q.tex_color);
                    }
                }
                s.materials_dev = upload(md.data(), md.size(), sc->owned);
            }
            s.images = upload(desc->images, desc->n_images, sc->owned);
            s.image_data = upload(desc->image_data, desc->image_data_bytes, sc->owned);
            s.perlins = upload(desc->perlins, desc->n_perlins, sc->owned);
            s.root = node_ref(desc->root);
            s.n_lights = desc->n_lights;
            s.n_nodes = desc->n_nodes;
            s.n_xforms = desc->n_xforms;
            s.n_media = desc->n_media;
            s.n_spheres = desc->n_spheres;
            s.n_moving_spheres = desc->n_moving_spheres;
            s.n_rects = desc->n_rects;
            sc->stack_need = (uint32_t)need;
            sc->general_boundaries = v.general_boundaries;
            sc->boxes_plain = v.boxes_plain();
            if (RT_REF_KIND(desc->root) == RT_KIND_NODE) {
                const rt_bvh_node &rn = desc->nodes[RT_REF_INDEX(desc->root)];
                for (int ax = 0; ax < 3; ax++) { const double c = 0.5 * (rn.bmin[ax] + rn.bmax[ax]); sc->split[ax] = std::isfinite(c) ? c : 0.0; }
            }
            sc->features = ((desc->n_triangles || desc->n_rings) ? kFeatMisc : 0u) |
                           ((desc->n_xforms || desc->n_lists) ? kFeatMovers : 0u) |
                           ((desc->n_boxes || desc->n_media) ? kFeatVolumes : 0u);
        } catch (...) {
            for (void *p : sc->owned) (void)hipFree(p);
            delete sc;
            throw;
        }
        *out = sc;
        return RT_OK;
    });
}

int rt_scene_destroy(rt_scene *scene) {
    return guarded([&]() -> int {
        if (!scene) return RT_OK;
        DeviceGuard guard(scene->device);
        for (auto &kv : scene->ws)                    // (asynchronous calls still in flight: let their host threads finish)
            if (kv.second.async_worker.joinable()) kv.second.async_worker.join();
        (void)hipDeviceSynchronize();                 // (every stream of the scene's device, the group streams included)
        for (auto &kv : scene->ws) {
            Workspace &w = kv.second;
            if (w.rows_max) (void)hipFree(w.rows_max);
            if (w.h_rows_max) (void)hipHostFree(w.h_rows_max);
            for (hipEvent_t ev : w.kt.ev) (void)hipEventDestroy(ev);
            if (w.work_counter) (void)hipFree(w.work_counter);
            if (w.stats) (void)hipFree(w.stats);
            if (w.partial) (void)hipFree(w.partial);
            if (w.tape) (void)hipFree(w.tape);
            for (void *p : w.pool_owned) (void)hipFree(p);
            if (w.d_scene) (void)hipFree(w.d_scene);
            if (w.d_args) (void)hipFree(w.d_args);
            if (w.d_pool) (void)hipFree(w.d_pool);
            if (w.gs.h_active) (void)hipHostFree(w.gs.h_active);
            if (w.gs.h_work) (void)hipHostFree(w.gs.h_work);
            if (w.gs.h_oldest) (void)hipHostFree(w.gs.h_oldest);
            if (w.d_limit) (void)hipFree(w.d_limit);
            for (int g = 0; g < kMaxGroups; g++) {
                if (w.gs.stream[g]) (void)hipStreamDestroy(w.gs.stream[g]);
                for (int b = 0; b < 2; b++) if (w.gs.ev[g][b]) (void)hipEventDestroy(w.gs.ev[g][b]);
            }
            if (w.ev0) (void)hipEventDestroy(w.ev0);
            if (w.ev1) (void)hipEventDestroy(w.ev1);
        }
        for (void *p : scene->owned) (void)hipFree(p);
        delete scene;
        return RT_OK;
    });
}

// Joins the worker of an RT_FLAG_ASYNC call in flight on (scene, stream), if any; returns what it ended with.
static int join_async(rt_scene *scene, hipStream_t stream, std::string *err, bool *joined = nullptr) {
    Workspace &w = workspace_for(scene, stream);
    if (joined) *joined = w.async_worker.joinable();
    if (!w.async_worker.joinable()) return RT_OK;
    w.async_worker.join();
    const int rc = w.async_rc;
    if (err) *err = w.async_err;
    w.async_rc = RT_OK; w.async_err.clear();
    return rc;
}

int rt_render_device(rt_scene *scene, const rt_camera *cam, const rt_params *params,
                     double *d_out_rgb_sum, void *hip_stream, rt_stats *stats) {
    return guarded([&]() -> int {
        check_params(scene, cam, params);
        RT_REQUIRE(d_out_rgb_sum || params->n_rows == 0, RT_ERR_INVALID, "rt_render_device: output is null");
        DeviceGuard guard(scene->device);
        const hipStream_t stream = (hipStream_t)hip_stream;
        {   // one call at a time per (scene, stream): an asynchronous one still in flight is finished first
            std::string err;
            bool joined = false;
            const int rc = join_async(scene, stream, &err, &joined);
            RT_REQUIRE(rc == RT_OK, rc, "rt_render_device: the previous asynchronous call on this stream failed: " + err);
            if (joined) finish(scene, stream);        // (its rt_stats, as rt_render_wait would have filled them)
        }
        if (!(params->flags & RT_FLAG_ASYNC)) {
            enqueue(scene, cam, params, params->row_ids, d_out_rgb_sum, stream, stats, true);
            return RT_OK;
        }
        // RT_FLAG_ASYNC: the engine drives its passes from a host thread (it polls one word per batch) — here a thread of the
        // library's own instead of the caller's. The arguments are copied; the device buffers are the caller's until the wait.
        Workspace &w = workspace_for(scene, stream);
        const rt_camera cam_copy = *cam;
        const rt_params params_copy = *params;
        w.async_rc = RT_OK; w.async_err.clear();
        w.async_worker = std::thread([scene, cam_copy, params_copy, d_out_rgb_sum, stream, stats, &w]() {
            try {
                DeviceGuard worker_guard(scene->device);
                enqueue(scene, &cam_copy, &params_copy, params_copy.row_ids, d_out_rgb_sum, stream, stats, true);
            } catch (const Fail &e) {
                w.async_rc = e.code; w.async_err = e.msg;
            } catch (const std::exception &e) {
                w.async_rc = RT_ERR_INVALID; w.async_err = e.what();
            }
        });
        return RT_OK;
    });
}

int rt_render_wait(rt_scene *scene, void *hip_stream) {
    return guarded([&]() -> int {
        RT_REQUIRE(scene, RT_ERR_INVALID, "rt_render_wait: null scene");
        DeviceGuard guard(scene->device);
        std::string err;
        const int rc = join_async(scene, (hipStream_t)hip_stream, &err);
        RT_REQUIRE(rc == RT_OK, rc, err);
        finish(scene, (hipStream_t)hip_stream);
        return RT_OK;
    });
}

int rt_render(rt_scene *scene, const rt_camera *cam, const rt_params *params, double *out_rgb_sum, rt_stats *stats) {
    return guarded([&]() -> int {
        check_params(scene, cam, params);
        RT_REQUIRE(out_rgb_sum || params->n_rows == 0, RT_ERR_INVALID, "rt_render: output is null");
        for (uint32_t i = 0; i < params->n_rows; i++)
            RT_REQUIRE(params->row_ids[i] < (uint64_t)params->height * params->n_frames, RT_ERR_INVALID, "rt_render: row id out of range");
        uint64_t n_values = (uint64_t)params->n_rows * params->width * 3;
        DeviceGuard guard(scene->device);
        uint32_t *d_rows = nullptr;
        double *d_out = nullptr;
        int rc = RT_OK;
        try {
            RT_HIP(hipMalloc((void **)&d_rows, (params->n_rows ? params->n_rows : 1) * sizeof(uint32_t)));
            RT_HIP(hipMalloc((void **)&d_out, (n_values ? n_values : 1) * sizeof(double)));
            if (params->n_rows) RT_HIP(hipMemcpy(d_rows, params->row_ids, params->n_rows * sizeof(uint32_t), hipMemcpyHostToDevice));
            // Poison the output so an unwritten pixel cannot pass for a result.
            RT_HIP(hipMemset(d_out, 0xFF, (n_values ? n_values : 1) * sizeof(double)));
            enqueue(scene, cam, params, d_rows, d_out, nullptr, stats);
            finish(scene, nullptr);
            if (n_values) RT_HIP(hipMemcpy(out_rgb_sum, d_out, n_values * sizeof(double), hipMemcpyDeviceToHost));
        } catch (const Fail &e) {
            set_error(e.msg);
            rc = e.code;
        }
        if (d_rows) (void)hipFree(d_rows);
        if (d_out) (void)hipFree(d_out);
        return rc;
    });
}

// ---- one call, several GPUs (rt2022.h) ----------------------------------------------------------------------
struct rt_scene_set {
    std::vector<int> devices;
    std::vector<rt_scene *> scenes;
};

int rt_scene_set_create(const rt_scene_desc *desc, uint64_t device_mask, rt_scene_set **out) {
    return guarded([&]() -> int {
        RT_REQUIRE(desc && out, RT_ERR_INVALID, "rt_scene_set_create: null argument");
        RT_REQUIRE(device_mask != 0, RT_ERR_INVALID, "rt_scene_set_create: empty device mask");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        RT_REQUIRE(e == hipSuccess && ndev > 0, RT_ERR_DEVICE, "rt_scene_set_create: no HIP device available (the path has no CPU fallback)");
        RT_REQUIRE(ndev >= 64 || (device_mask >> ndev) == 0, RT_ERR_DEVICE, "rt_scene_set_create: device_mask names a device this process cannot see");
        int prev = 0;
        RT_HIP(hipGetDevice(&prev));
        rt_scene_set *set = new rt_scene_set();
        int rc = RT_OK;
        for (int d = 0; d < ndev && d < 64 && rc == RT_OK; d++) {
            if (!((device_mask >> d) & 1ull)) continue;
            if (hipSetDevice(d) != hipSuccess) { set_error("rt_scene_set_create: hipSetDevice failed"); rc = RT_ERR_DEVICE; break; }
            rt_scene *sc = nullptr;
            rc = rt_scene_create(desc, &sc);                   // (leaves its own message on failure)
            if (rc == RT_OK) { set->devices.push_back(d); set->scenes.push_back(sc); }
        }
        (void)hipSetDevice(prev);
        if (rc != RT_OK) {
            std::string msg = rt_last_error();
            for (rt_scene *sc : set->scenes) (void)rt_scene_destroy(sc);
            delete set;
            set_error(msg);
            return rc;
        }
        *out = set;
        return RT_OK;
    });
}

int rt_scene_set_destroy(rt_scene_set *set) {
    return guarded([&]() -> int {
        if (!set) return RT_OK;
        for (rt_scene *sc : set->scenes) (void)rt_scene_destroy(sc);
        delete set;
        return RT_OK;
    });
}

int rt_render_multi(rt_scene_set *set, const rt_camera *cam, const rt_params *params, double *out_rgb_sum, rt_stats *stats) {
    return guarded([&]() -> int {
        RT_REQUIRE(set && !set->scenes.empty() && cam && params, RT_ERR_INVALID, "rt_render_multi: null argument");
        RT_REQUIRE(params->n_rows == 0 || (params->row_ids && out_rgb_sum), RT_ERR_INVALID, "rt_render_multi: null rows or output");
        const size_t n = set->scenes.size();
        const size_t row_doubles = (size_t)params->width * 3;
        struct Share {
            std::vector<uint32_t> rows;
            std::vector<double> out;
            rt_stats st;
            int rc = RT_OK;
            std::string err;
        };
        std::vector<Share> shares(n);
        for (uint32_t i = 0; i < params->n_rows; i++) shares[i % n].rows.push_back(params->row_ids[i]);
        std::vector<std::thread> workers;
        for (size_t k = 0; k < n; k++) {
            workers.emplace_back([&, k]() {
                Share &sh = shares[k];
                std::memset(&sh.st, 0, sizeof sh.st);
                rt_params p = *params;
                p.n_rows = (uint32_t)sh.rows.size();
                p.row_ids = sh.rows.data();
                // per-worker progress: this device's share under its place in the set (main.rs:124-127: one bar per thread)
                struct Relay { void (*cb)(void *, uint32_t, uint64_t, uint64_t); void *user; uint32_t worker; } relay{params->progress_cb, params->progress_user, (uint32_t)k};
                if (params->progress_cb) {
                    p.progress_user = &relay;
                    p.progress_cb = [](void *u, uint32_t, uint64_t done, uint64_t total) { const Relay *r = static_cast<const Relay *>(u); r->cb(r->user, r->worker, done, total); };
                }
                sh.out.resize(sh.rows.size() * row_doubles);
                // (rt_render makes the scene's device current for this thread and leaves the caller's alone)
                sh.rc = rt_render(set->scenes[k], cam, &p, sh.out.data(), &sh.st);
                if (sh.rc != RT_OK) sh.err = rt_last_error();
            });
        }
        for (std::thread &t : workers) t.join();
        for (size_t k = 0; k < n; k++)
            if (shares[k].rc != RT_OK) throw Fail{shares[k].rc, "rt_render_multi: device " + std::to_string(set->devices[k]) + ": " + shares[k].err};
        std::vector<size_t> taken(n, 0);
        for (uint32_t i = 0; i < params->n_rows; i++) {
            Share &sh = shares[i % n];
            std::memcpy(out_rgb_sum + (size_t)i * row_doubles, sh.out.data() + taken[i % n]++ * row_doubles, row_doubles * sizeof(double));
        }
        if (stats) {
            rt_stats tot;
            std::memset(&tot, 0, sizeof tot);
            for (const Share &sh : shares) {
                tot.paths += sh.st.paths; tot.rays += sh.st.rays; tot.node_visits += sh.st.node_visits;
                for (int k = 0; k < RT_KIND_COUNT; k++) tot.prim_tests[k] += sh.st.prim_tests[k];
                tot.light_pdf_tests += sh.st.light_pdf_tests; tot.rng_draws += sh.st.rng_draws;
                tot.ms = sh.st.ms > tot.ms ? sh.st.ms : tot.ms;
                tot.trace_ms = sh.st.trace_ms > tot.trace_ms ? sh.st.trace_ms : tot.trace_ms;
                tot.shade_ms = sh.st.shade_ms > tot.shade_ms ? sh.st.shade_ms : tot.shade_ms;
                tot.spp_chunk = sh.st.spp_chunk; tot.passes = sh.st.passes > tot.passes ? sh.st.passes : tot.passes;
                tot.pool_slots += sh.st.pool_slots;
            }
            *stats = tot;
        }
        return RT_OK;
    });
}

int rt_tonemap_device(const double *d_rgb_sum, uint64_t n_pixels, int32_t spp, uint8_t *d_rgb8, void *hip_stream) {
    return guarded([&]() -> int {
        RT_REQUIRE((d_rgb_sum && d_rgb8) || n_pixels == 0, RT_ERR_INVALID, "rt_tonemap_device: null argument");
        if (n_pixels) RT_HIP(launch_tonemap(d_rgb_sum, n_pixels, spp, d_rgb8, (hipStream_t)hip_stream));
        return RT_OK;
    });
}

// ---- probes (include/rt2022_debug.h) --------------------------------------------------
int rt_debug_math_device(int op, const double *a, const double *b, double *out, uint64_t n) {
    return guarded([&]() -> int {
        RT_REQUIRE(a && out, RT_ERR_INVALID, "rt_debug_math_device: null argument");
        double *da = nullptr, *db = nullptr, *dout = nullptr;
        int rc = RT_OK;
        try {
            RT_HIP(hipMalloc((void **)&da, n * 8 + 8));
            RT_HIP(hipMalloc((void **)&dout, n * 8 + 8));
            RT_HIP(hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice));
            if (b) { RT_HIP(hipMalloc((void **)&db, n * 8 + 8)); RT_HIP(hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice)); }
            RT_HIP(launch_math_probe(op, da, db, dout, n, nullptr));
            RT_HIP(hipDeviceSynchronize());
            RT_HIP(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
        } catch (const Fail &e) { set_error(e.msg); rc = e.code; }
        if (da) (void)hipFree(da);
        if (db) (void)hipFree(db);
        if (dout) (void)hipFree(dout);
        return rc;
    });
}

int rt_debug_rng_device(uint64_t state, int mode, double lo, double hi, uint64_t bound, uint64_t *out, uint64_t n) {
    return guarded([&]() -> int {
        RT_REQUIRE(out, RT_ERR_INVALID, "rt_debug_rng_device: null argument");
        uint64_t *dout = nullptr;
        int rc = RT_OK;
        try {
            RT_HIP(hipMalloc((void **)&dout, n * 8 + 8));
            RT_HIP(launch_rng_probe(state, mode, lo, hi, bound, dout, n, nullptr));
            RT_HIP(hipDeviceSynchronize());
            RT_HIP(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
        } catch (const Fail &e) { set_error(e.msg); rc = e.code; }
        if (dout) (void)hipFree(dout);
        return rc;
    });
}

} // extern "C"

// ---- HBM counter calibration (tools/traffic_calib.sh) -------------------------------------------------------------
// MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are calibrated for wide streaming accesses only ("calibrate on a known
// byte count in your own access pattern before trusting an absolute"). These kernels move a KNOWN number of bytes in
// the path pool's patterns — a 128-byte record per slot, slots visited in random order over a buffer far larger than
// the 256 MiB Infinity Cache — so that the counters read under rocprofv3 can be set against them.
namespace {
typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t probe_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
// MODE 0: streaming read, 16 B per lane; 1: 64 B (first half) of a random record; 2: all 128 B of a random record;
// 3: streaming write; 4: 32 B written at +64 of a random record (a winner); 5: 64 + 32 B written (ray + bookkeeping);
// 6: one byte written at a random position (the old per-slot kind array).
template <int MODE>
__global__ void __launch_bounds__(256) traffic_probe_kernel(probe_u32x4 *buf, uint64_t n_records, uint64_t n_access, uint64_t seed, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    probe_u32x4 acc = {0u, 0u, 0u, 0u};
    for (uint64_t i = gid; i < n_access; i += stride) {
        if (MODE == 0) { acc += buf[i]; continue; }
        if (MODE == 3) { buf[i] = (probe_u32x4){(uint32_t)i, 1u, 2u, 3u}; continue; }
        const uint64_t rec = probe_mix(i ^ seed) % n_records;
        probe_u32x4 *r = buf + rec * 8;                              // 128-byte record = 8 x 16 B
        if (MODE == 1) { acc += r[0]; acc += r[1]; acc += r[2]; acc += r[3]; }
        if (MODE == 2) { for (int k = 0; k < 8; k++) acc += r[k]; }
        if (MODE == 4) { r[4] = (probe_u32x4){(uint32_t)i, 1u, 2u, 3u}; r[5] = (probe_u32x4){4u, 5u, 6u, 7u}; }
        if (MODE == 5) { for (int k = 0; k < 4; k++) r[k] = (probe_u32x4){(uint32_t)i, (uint32_t)k, 2u, 3u}; r[6] = (probe_u32x4){1u, 1u, 1u, 1u}; r[7] = (probe_u32x4){2u, 2u, 2u, 2u}; }
        if (MODE == 6) { reinterpret_cast<uint8_t *>(buf)[probe_mix(i ^ seed ^ 0x5555) % (n_records * 128)] = (uint8_t)i; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = acc.x;      // (keeps the loads alive)
}
template <int MODE>
void launch_probe(probe_u32x4 *buf, uint64_t n_records, uint64_t n_access, uint64_t seed, uint32_t *sink) {
    hipLaunchKernelGGL((traffic_probe_kernel<MODE>), dim3(256 * 16), dim3(256), 0, nullptr, buf, n_records, n_access, seed, sink);
}
} // namespace

extern "C" {

int rt_debug_traffic_probe(int mode, uint64_t buffer_bytes, uint64_t n_access, uint64_t seed) {
    return guarded([&]() -> int {
        RT_REQUIRE(mode >= 0 && mode <= 6 && buffer_bytes >= 4096, RT_ERR_INVALID, "rt_debug_traffic_probe: bad arguments");
        probe_u32x4 *buf = nullptr;
        uint32_t *sink = nullptr;
        int rc = RT_OK;
        try {
            RT_HIP(hipMalloc((void **)&buf, buffer_bytes));
            RT_HIP(hipMalloc((void **)&sink, 4));
            RT_HIP(hipMemset(buf, 1, buffer_bytes));
            RT_HIP(hipDeviceSynchronize());
            const uint64_t n_records = buffer_bytes / 128;
            if (mode == 0 || mode == 3) n_access = buffer_bytes / 16;
            switch (mode) {
                case 0: launch_probe<0>(buf, n_records, n_access, seed, sink); break;
                case 1: launch_probe<1>(buf, n_records, n_access, seed, sink); break;
                case 2: launch_probe<2>(buf, n_records, n_access, seed, sink); break;
                case 3: launch_probe<3>(buf, n_records, n_access, seed, sink); break;
                case 4: launch_probe<4>(buf, n_records, n_access, seed, sink); break;
                case 5: launch_probe<5>(buf, n_records, n_access, seed, sink); break;
                default: launch_probe<6>(buf, n_records, n_access, seed, sink); break;
            }
            RT_HIP(hipGetLastError());
            RT_HIP(hipDeviceSynchronize());
        } catch (const Fail &e) { set_error(e.msg); rc = e.code; }
        if (buf) (void)hipFree(buf);
        if (sink) (void)hipFree(sink);
        return rc;
    });
}

} // extern "C"

// ---- VALU counter calibration (tools/valu_calib.sh) ------------------------------------------------------------------
// What do SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_THREAD_CYCLES_VALU / SQ_BUSY_CYCLES read for a kernel whose vector
// pipes are KNOWN to be saturated? These kernels issue nothing but independent vector instructions of one kind at 8
// waves per SIMD on every CU, so their issue-slot occupancy is 1 by construction; the counters read under rocprofv3 give
// the normalisation bench.py uses to turn the traversal kernel's counters into a measured busy fraction.
namespace {
// MODE 0: v_fma_f64, all lanes; 1: 32-bit integer VALU (v_add / v_xor), all lanes; 2: v_fma_f64 with half the lanes
// switched off (EXEC = low 32 lanes); 3: four f64 and four 32-bit instructions alternating; 4: v_fma_f64 at ONE wave per SIMD.
template <int MODE>
__global__ void __launch_bounds__(256) valu_probe_kernel(double *out, uint32_t iters, double b, double c, uint32_t k) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    double a0 = (double)gid, a1 = a0 + 1.0, a2 = a0 + 2.0, a3 = a0 + 3.0, a4 = a0 + 4.0, a5 = a0 + 5.0, a6 = a0 + 6.0, a7 = a0 + 7.0;
    uint32_t x0 = gid, x1 = gid + 1u, x2 = gid + 2u, x3 = gid + 3u, x4 = gid + 4u, x5 = gid + 5u, x6 = gid + 6u, x7 = gid + 7u;
    const bool on = MODE != 2 || (threadIdx.x & 63u) < 32u;
    if (on) {
        for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
          for (int rep = 0; rep < 8; rep++) {                // (64 vector instructions between two loop branches)
            if (MODE == 0 || MODE == 2 || MODE == 4) {
                a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
                a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
            } else if (MODE == 1) {
                x0 = (x0 + k) ^ x4; x1 = (x1 + k) ^ x5; x2 = (x2 + k) ^ x6; x3 = (x3 + k) ^ x7;
                x4 = (x4 + k) ^ x0; x5 = (x5 + k) ^ x1; x6 = (x6 + k) ^ x2; x7 = (x7 + k) ^ x3;
            } else {
                a0 = __builtin_fma(a0, b, c); x0 = (x0 + k) ^ x4; a1 = __builtin_fma(a1, b, c); x1 = (x1 + k) ^ x5;
                a2 = __builtin_fma(a2, b, c); x2 = (x2 + k) ^ x6; a3 = __builtin_fma(a3, b, c); x3 = (x3 + k) ^ x7;
            }
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));   // (no folding of the loop)
          }
        }
    }
    out[gid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7);
}
} // namespace

extern "C" {

int rt_debug_valu_probe(int mode, uint32_t iters) {
    return guarded([&]() -> int {
        RT_REQUIRE(mode >= 0 && mode <= 4 && iters > 0, RT_ERR_INVALID, "rt_debug_valu_probe: bad arguments");
        int dev = 0, cus = 0;
        RT_HIP(hipGetDevice(&dev));
        RT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const uint32_t blocks = (uint32_t)cus * (mode == 4 ? 1u : 8u);       // 8 workgroups of 4 waves per CU = 8 waves per SIMD (mode 4: one)
        double *out = nullptr;
        int rc = RT_OK;
        try {
            RT_HIP(hipMalloc((void **)&out, (uint64_t)blocks * 256 * sizeof(double)));
            const double b = 0.9999999, c = 1e-9;
            const uint32_t k = 0x9E3779B9u;
            switch (mode) {
                case 0: hipLaunchKernelGGL((valu_probe_kernel<0>), dim3(blocks), dim3(256), 0, nullptr, out, iters, b, c, k); break;
                case 1: hipLaunchKernelGGL((valu_probe_kernel<1>), dim3(blocks), dim3(256), 0, nullptr, out, iters, b, c, k); break;
                case 2: hipLaunchKernelGGL((valu_probe_kernel<2>), dim3(blocks), dim3(256), 0, nullptr, out, iters, b, c, k); break;
                case 3: hipLaunchKernelGGL((valu_probe_kernel<3>), dim3(blocks), dim3(256), 0, nullptr, out, iters, b, c, k); break;
                default: hipLaunchKernelGGL((valu_probe_kernel<4>), dim3(blocks), dim3(256), 0, nullptr, out, iters, b, c, k); break;
            }
            RT_HIP(hipGetLastError());
            RT_HIP(hipDeviceSynchronize());
        } catch (const Fail &e) { set_error(e.msg); rc = e.code; }
        if (out) (void)hipFree(out);
        return rc;
    });
}

int rt_debug_set_tuning(rt_scene *scene, uint32_t node_quorum, uint32_t vote_weights) {
    return guarded([&]() -> int {
        RT_REQUIRE(scene, RT_ERR_INVALID, "rt_debug_set_tuning: null scene");
        RT_REQUIRE((node_quorum & 0xFFu) >= 1 && (node_quorum & 0xFFu) <= 64 && true, RT_ERR_INVALID, "rt_debug_set_tuning: node_quorum must be 1..64 (+ extra sphere repeats << 8, + tail factor << 12, + segments per trace workgroup << 16, + long-first class shift << 20, + groups << 24, + 1 << 29: pass-timing probe, + 1 << 30: literal node step only)");
        if (vote_weights != 0)                  // (0 = the engine's default)
            for (int o = 0; o < 8; o++) RT_REQUIRE(((vote_weights >> (4 * o)) & 0xFu) != 0, RT_ERR_INVALID, "rt_debug_set_tuning: a vote weight is 0");
        scene->node_quorum = node_quorum;
        scene->vote_weights = vote_weights;
        return RT_OK;
    });
}

int rt_debug_set_engine(rt_scene *scene, int engine, int max_pool_blocks) {
    return guarded([&]() -> int {
        RT_REQUIRE(scene, RT_ERR_INVALID, "rt_debug_set_engine: null scene");
        RT_REQUIRE(engine == 0 || engine == 1, RT_ERR_INVALID, "rt_debug_set_engine: engine must be 0 (megakernel) or 1 (wavefront)");
        RT_REQUIRE(max_pool_blocks >= 0 && max_pool_blocks <= 65535, RT_ERR_INVALID, "rt_debug_set_engine: bad max_pool_blocks");
        RT_REQUIRE(engine == 1 || !scene->general_boundaries, RT_ERR_UNSUPPORTED,
                   "the megakernel engine only handles media whose boundary is one primitive under movers");
        scene->engine = engine;
        scene->max_pool_blocks = max_pool_blocks;
        return RT_OK;
    });
}

int rt_debug_set_partial_ring(rt_scene *scene, int planes) {
    return guarded([&]() -> int {
        RT_REQUIRE(scene, RT_ERR_INVALID, "rt_debug_set_partial_ring: null scene");
        RT_REQUIRE(planes >= -1, RT_ERR_INVALID, "rt_debug_set_partial_ring: planes must be -1 (never), 0 (automatic) or a plane count");
        scene->partial_ring = planes;
        return RT_OK;
    });
}

int rt_debug_pass_timing(const rt_scene *scene, double out[5]) {
    return guarded([&]() -> int {
        RT_REQUIRE(scene && out, RT_ERR_INVALID, "rt_debug_pass_timing: null argument");
        for (int i = 0; i < 5; i++) out[i] = scene->pass_timing[i];
        return RT_OK;
    });
}

int rt_debug_census(const rt_scene *scene, uint64_t rounds[9], uint64_t lanes[9]) {
    return guarded([&]() -> int {
        RT_REQUIRE(scene && rounds && lanes, RT_ERR_INVALID, "rt_debug_census: null argument");
        for (int o = 0; o < 9; o++) { rounds[o] = scene->census_rounds[o]; lanes[o] = scene->census_lanes[o]; }
        return RT_OK;
    });
}

int rt_debug_scene_info(const rt_scene *scene, uint32_t *stack_need, int32_t *grid_blocks) {
    return guarded([&]() -> int {
        RT_REQUIRE(scene, RT_ERR_INVALID, "rt_debug_scene_info: null scene");
        if (stack_need) *stack_need = scene->stack_need;
        if (grid_blocks) *grid_blocks = render_grid_blocks(scene->stack_need, false);
        return RT_OK;
    });
}

int rt_debug_trace_variant(const rt_scene *scene, uint32_t *workgroup_threads, uint32_t *stack_entries, uint32_t *nodes_in_lds,
                           uint32_t *spheres_in_lds) {
    return guarded([&]() -> int {
        RT_REQUIRE(scene, RT_ERR_INVALID, "rt_debug_trace_variant: null scene");
        uint32_t v[4] = {0, 0, 0, 0};
        if (scene->engine == 1) trace_variant(scene->dev, scene->stack_need, scene->node_quorum, scene->features, v);
        if (workgroup_threads) *workgroup_threads = v[0];
        if (stack_entries) *stack_entries = v[1];
        if (nodes_in_lds) *nodes_in_lds = v[2];
        if (spheres_in_lds) *spheres_in_lds = v[3];           // (bit 0: the sphere pools are in LDS; bit 1: node boxes are tested in single precision)
        return RT_OK;
    });
}

int rt_debug_f32_slabs(uint64_t out[5]) {
    return guarded([&]() -> int {
        RT_REQUIRE(out, RT_ERR_INVALID, "rt_debug_f32_slabs: null output");
        unsigned long long v[5] = {0, 0, 0, 0, 0};
        RT_HIP(f32_slab_census(v));
        for (int i = 0; i < 5; i++) out[i] = v[i];
        return RT_OK;
    });
}

} // extern "C"

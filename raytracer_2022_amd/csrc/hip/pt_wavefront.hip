// pt_wavefront.hip — the wavefront engine: the same per-path algorithm as the megakernel
// (pt_kernel.hip), cut at the one point where its register needs change character.
//
// Why two kernels: measured on MI355X, the single megakernel needs 256 VGPRs + scratch
// (traversal state and the shading temporaries — Perlin turbulence, ONB, light pdfs — are
// live together), which caps it at 2 waves/SIMD, and its spills land in the node loop
// (profiles/r1_megakernel_*.txt). Split at "closest hit found", the traversal kernel is
// lean and the shading kernel is wide, and each gets the occupancy it can use.
//
// Paths live in a pool of slots in HBM (WfPool, one record per slot and field). The pool is cut
// into segments of 4096 slots; a segment belongs to one shade workgroup for the whole frame; the trace
// pass is a persistent grid whose waves draw chunks of the segments' ray lists — so the only global
// atomics on the data path are the work counter (items) and the chunk counter (ray lists):
//   wf_shade  counting-sorts its slots by what they wait for (miss / light / lambertian by
//             texture / metal / dielectric / isotropic / fresh) in LDS and shades them in that
//             order — material dispatch by sorted type id, wave-uniform except at bin boundaries;
//             finished paths are unwound from the bounce tape, added to their pixel, and
//             replaced by the next sample / work item at once; last, it writes the segment's
//             ray list for the trace pass, longest expected traversal first;
//   wf_trace  runs the in-wave scheduled traversal over the lists: a wave claims 256 entries at a
//             time, its lanes pull the next ray as soon as theirs is done (__ballot / __popcll
//             refill), and the wave executes the operation most lanes wait for (node step, sphere
//             test, box, medium, ...).
// The host alternates the two until no slot carries a ray any more.
//
// Per-lane semantics never change: every path consumes its RNG stream and visits nodes
// in the reference's order, so results stay bit-identical to the oracle.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pt_common.hpp"

namespace rt2022 {

namespace {

constexpr int S = kSlotsPerBlock;
#ifndef RT2022_CHUNK
#define RT2022_CHUNK 256
#endif
#ifndef RT2022_LIST_OCTANTS
#define RT2022_LIST_OCTANTS 1
#endif
#ifndef RT2022_LIST_ORIGIN
#define RT2022_LIST_ORIGIN 1           // 0: no third key; 1: four classes (camera / sphere / box, rect / medium); 2: eight (the leaf kind itself)
#endif
// list order: 16 classes of expected length x classes of where the ray starts x 8 direction octants
constexpr uint32_t kOriginClasses = RT2022_LIST_ORIGIN == 2 ? 8u : RT2022_LIST_ORIGIN ? 4u : 1u;
#ifndef RT2022_LIST_SPATIAL
#define RT2022_LIST_SPATIAL 0          // a fourth key — 1: the quadrant (x, z about the centre of the root's box) the ray starts in; 2: the dominant axis of its direction
#endif
constexpr uint32_t kSpatialClasses = RT2022_LIST_SPATIAL ? 4u : 1u;
constexpr uint32_t kListBins = 16 * kOriginClasses * 8 * kSpatialClasses;
constexpr uint32_t kChunk = RT2022_CHUNK;           // list entries a wave claims at a time

// Records are fetched whole and at once — a few 16-byte loads issued back to back and waited for together — never
// field by field as the arithmetic gets to them: left to itself the compiler sinks each field's load into the branch
// that uses it, and an arm like Boxes::hit then waits for memory six to ten times in a row (seen in the ISA). The empty
// asm pins the value: the load cannot move below it, and everything pinned together shares one wait.
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));     // (records whose size is 8 mod 16)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef RT2022_F32_GLOBAL
#define RT2022_F32_GLOBAL 1            // the same test on 32-byte single-precision node records from HBM / L2 (wf_trace, kF32G): 1 the sphere scenes too large
                                       // for LDS and the triangle meshes, 2 every instance without boxes / media (A/B, census), 0 none
#endif
#ifndef RT2022_STASH
#define RT2022_STASH 1                 // keep 1/d.x, 1/d.z of the frame a RotateY is entered from (0: two divisions at its exit instead — five registers fewer)
#endif
#ifndef RT2022_F32_SLABS
#define RT2022_F32_SLABS 1             // node table in LDS: single-precision slab test with a double-precision second opinion (wf_trace, t_slabs32):
                                       // 1 the all-in-LDS instance of sphere-only scenes, 2 every instance that holds the whole table (A/B, census), 0 none
#endif
template <class T>
RT_DEV void t_pin(T &v) { asm volatile("" : "+v"(v)); }
// The wave's vote as the hardware gives it (a v_cmp into an SGPR pair); HIP's __ballot materialises the predicate as 0 / 1 first.
RT_DEV unsigned long long wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

struct PoolView {
    const WfPool &p;
    // Ray + RNG state: one 64-byte line per slot.
    RT_DEV Ray load_ray(uint32_t slot, uint64_t &rng_state) const {
        const double2 *q = reinterpret_cast<const double2 *>(p.ray + (uint64_t)slot * kRecDoubles);
        double2 a = q[0], b = q[1], c = q[2], d = q[3];
        rng_state = rtm::d2u(d.y);
        return Ray(Vec3(a.x, a.y, b.x), Vec3(b.y, c.x, c.y), d.x);
    }
    RT_DEV Ray load_ray(uint32_t slot) const { uint64_t unused; return load_ray(slot, unused); }
    RT_DEV void store_ray(uint32_t slot, const Ray &r, uint64_t rng_state) const {
        double2 *q = reinterpret_cast<double2 *>(p.ray + (uint64_t)slot * kRecDoubles);
        q[0] = make_double2(r.orig.x, r.orig.y);
        q[1] = make_double2(r.orig.z, r.dir.x);
        q[2] = make_double2(r.dir.y, r.dir.z);
        q[3] = make_double2(r.tm, rtm::u2d(rng_state));
    }
    RT_DEV void store_rng(uint32_t slot, uint64_t rng_state) const { p.ray[(uint64_t)slot * kRecDoubles + 7] = rtm::u2d(rng_state); }
    // Winner of the traversal: one 32-byte record per slot.
    // meta = box face | movers << 4 | node steps of the traversal << 16 (the shade pass orders the next
    // trace pass by them); a miss stores nothing (its path ends).
    // Second half = the movers enclosing the leaf; its last word holds the leaf's material word instead whenever the
    // chain leaves it free (fewer than four movers): the shade pass then needs no look at the primitive for it.
    RT_DEV void store_hit(uint32_t slot, double t, uint32_t leaf, uint32_t meta, const Chain &ch, uint32_t mat_word) const {
        u32x4 *q = reinterpret_cast<u32x4 *>(p.hit + (uint64_t)slot * kRecWords);
        uint64_t tb = rtm::d2u(t);
        q[0] = (u32x4){(uint32_t)tb, (uint32_t)(tb >> 32), leaf, meta};
        q[1] = (u32x4){ch.c0, ch.c1, ch.c2, ch.n >= 4u ? ch.c3 : mat_word};
    }
    RT_DEV static void decode_hit(u32x4 a, u32x4 b, Winner &w, uint32_t &steps, uint32_t &mat_word, bool &have_mat) {
        w.t = rtm::u2d(((uint64_t)a.y << 32) | a.x);
        w.leaf = a.z;
        w.face = a.w & 0xFu;
        w.chain.n = (a.w >> 4) & 0xFu;
        steps = a.w >> 16;
        have_mat = w.chain.n < 4u;
        mat_word = b.w;
        w.chain.c0 = b.x; w.chain.c1 = b.y; w.chain.c2 = b.z; w.chain.c3 = have_mat ? 0u : b.w;
    }
};

// Bounce tape of one slot (see Tape in pt_kernel.hip): its records are contiguous in HBM,
// record k = 4 doubles {w.x, w.y, w.z, p} at tape[(slot*cap + k)*4], so unwinding a path reads
// a few adjacent cache lines; the loads of four records are issued together before use.
struct SlotTape {
    double *base;          // this slot's first record
    RT_DEV void put(uint32_t k, Vec3 w, double p) const {
        double2 *q = reinterpret_cast<double2 *>(base + (uint64_t)k * 4);
        q[0] = make_double2(w.x, w.y);
        q[1] = make_double2(w.z, p);
    }
    RT_DEV static Vec3 step(Vec3 Lr, double2 a, double2 b) {
        Vec3 w(a.x, a.y, b.x);
        return Vec3(0.0, 0.0, 0.0) + (w * Lr) / b.y;           // emitted + ((att*spdf) * L) / pdf_val, main.rs:267-271
    }
    RT_DEV Vec3 unwind(uint32_t nb, Vec3 Lr) const {
        const double2 *q = reinterpret_cast<const double2 *>(base);
        uint32_t k = nb;
        while (k >= 4) {
            double2 a3 = q[2 * (k - 1)], b3 = q[2 * (k - 1) + 1], a2 = q[2 * (k - 2)], b2 = q[2 * (k - 2) + 1];
            double2 a1 = q[2 * (k - 3)], b1 = q[2 * (k - 3) + 1], a0 = q[2 * (k - 4)], b0 = q[2 * (k - 4) + 1];
            Lr = step(Lr, a3, b3); Lr = step(Lr, a2, b2); Lr = step(Lr, a1, b1); Lr = step(Lr, a0, b0);
            k -= 4;
        }
        for (; k > 0; k--) Lr = step(Lr, q[2 * (k - 1)], q[2 * (k - 1) + 1]);
        return Lr;
    }
};

// Path bookkeeping of one slot, one 32-byte record: {item's index in the partial sums (u64), smp, smp_end, depth, px, py, frame}.
struct SlotState {
    uint64_t item;
    uint32_t smp, smp_end, depth, px, py, frame;
};
RT_DEV SlotState load_state(const WfPool &p, uint32_t slot) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p.state + (uint64_t)slot * kRecWords);
    uint4 a = q[0], b = q[1];
    SlotState st;
    st.item = ((uint64_t)a.y << 32) | a.x;
    st.smp = a.z; st.smp_end = a.w; st.depth = b.x; st.px = b.y; st.py = b.z; st.frame = b.w;
    return st;
}
// (A bounce changes the depth only: the first half is rewritten when a new sample or item starts.)
RT_DEV void store_state(const WfPool &p, uint32_t slot, const SlotState &st, bool whole) {
    uint4 *q = reinterpret_cast<uint4 *>(p.state + (uint64_t)slot * kRecWords);
    if (whole) q[0] = make_uint4((uint32_t)st.item, (uint32_t)(st.item >> 32), st.smp, st.smp_end);
    q[1] = make_uint4(st.depth, st.px, st.py, st.frame);
}

// The device copies of the primitive pools carry, above the material index, the slot kind a hit on the primitive
// leads to (kMatKindShift; rt_scene_create): publishing a winner then costs one dependent load, not three.
RT_DEV bool t_finite_s(double x) { return (rtm::d2u(x) & 0x7FF0000000000000ull) != 0x7FF0000000000000ull; }

RT_DEV uint32_t leaf_material_word(const SceneDev &s, uint32_t leaf) {
    uint32_t idx = RT_REF_INDEX(leaf);
    switch (RT_REF_KIND(leaf)) {
        case RT_KIND_SPHERE: return s.spheres[idx].mat;
        case RT_KIND_MOVING_SPHERE: return s.moving_spheres[idx].mat;
        case RT_KIND_RECT: return s.rects[idx].mat;
        case RT_KIND_BOX: return s.boxes[idx].mat;
        case RT_KIND_TRIANGLE: return s.triangles[idx].mat;
        case RT_KIND_RING: return s.rings[idx].mat;
        default: return s.media[idx].mat;
    }
}


// The winning primitive's record as the shade pass fetches it: a fixed 80 bytes from the record's address, whatever
// the kind (the longest records — Triangle, MovingSphere — are 80 bytes; shorter ones run on into their neighbour or
// into the pool's zeroed slack, rt_scene_create) — one address computation, five loads, no branch.
struct PrimRegs { f64x2 r0, r1, r2, r3, r4; };
// ... with the pools' base addresses and record sizes taken from a 16-entry table in LDS, indexed by kind (wf_shade fills it
// once): seven pointer pairs need not sit in scalar registers through the whole shade loop (r3: they were being spilled), and
// the select chain below becomes one 8-byte LDS read.
struct PrimTable { unsigned long long base[16]; uint32_t stride[16]; };
RT_DEV void prim_table_fill(const SceneDev &s, PrimTable &t, uint32_t tid) {
    if (tid < 16) {
        const void *b = s.media; uint32_t st = (uint32_t)sizeof(rt_medium);
        if (tid == RT_KIND_SPHERE) { b = s.spheres; st = (uint32_t)sizeof(rt_sphere); }
        else if (tid == RT_KIND_MOVING_SPHERE) { b = s.moving_spheres; st = (uint32_t)sizeof(rt_moving_sphere); }
        else if (tid == RT_KIND_RECT) { b = s.rects; st = (uint32_t)sizeof(rt_rect); }
        else if (tid == RT_KIND_BOX) { b = s.boxes; st = (uint32_t)sizeof(rt_box); }
        else if (tid == RT_KIND_TRIANGLE) { b = s.triangles; st = (uint32_t)sizeof(rt_triangle); }
        else if (tid == RT_KIND_RING) { b = s.rings; st = (uint32_t)sizeof(rt_ring); }
        t.base[tid] = (unsigned long long)reinterpret_cast<uintptr_t>(b);
        t.stride[tid] = st;
    }
}
RT_DEV const f64x2_a8 *prim_address(const PrimTable &t, uint32_t leaf) {
    const uint32_t kind = RT_REF_KIND(leaf), idx = RT_REF_INDEX(leaf);
    return reinterpret_cast<const f64x2_a8 *>(static_cast<uintptr_t>(t.base[kind] + (unsigned long long)idx * t.stride[kind]));
}
// winner_record (pt_common.hpp) fed from registers: the HitRecord of the winning candidate, rebuilt from (leaf, t) in
// the leaf's own frame (sphere.rs:59-65,158-164, aarect.rs:51-71, boxes.rs:24-66, triangle.rs:54-76, ring.rs:49-52,
// constantmedium.rs:66-74) and then carried out through its movers.
RT_DEV void winner_record_regs(const SceneDev &s, const Ray &wr, const Winner &w, const PrimRegs &q, HitRec &rec, bool want_uv) {
    const XRay world{wr.orig, wr.dir};
    XRay r = ray_at_level(s, w.chain, w.chain.n, world);
    const uint32_t kind = RT_REF_KIND(w.leaf);
    const double t = w.t;
    rec.mat = 0;
    switch (kind) {
        case RT_KIND_SPHERE: case RT_KIND_MOVING_SPHERE: {
            Vec3 center; double radius;
            if (kind == RT_KIND_SPHERE) { center = Vec3(q.r0.x, q.r0.y, q.r1.x); radius = q.r1.y; }
            else {
                const Vec3 c0(q.r0.x, q.r0.y, q.r1.x), c1(q.r1.y, q.r2.x, q.r2.y);
                center = c0 + (c1 - c0) * ((wr.tm - q.r3.x) / (q.r3.y - q.r3.x));
                radius = q.r4.x;
            }
            Vec3 at = r.o + r.d * t;
            Vec3 outward_normal = (at - center) / radius;
            rec.u = 0.0; rec.v = 0.0;
            if (want_uv) sphere_uv(outward_normal, rec.u, rec.v);
            rec.p = at; rec.t = t;
            rec.set_face_normal(r.d, outward_normal);
            break;
        }
        case RT_KIND_RECT: {
            RectP rp{(uint32_t)rtm::d2u(q.r2.y), q.r0.x, q.r0.y, q.r1.x, q.r1.y, q.r2.x};
            rect_record(rp, 0u, r, t, rec);
            break;
        }
        case RT_KIND_BOX: {
            const double p0x = q.r0.x, p0y = q.r0.y, p0z = q.r1.x, p1x = q.r1.y, p1y = q.r2.x, p1z = q.r2.y;
            const uint32_t i = w.face;
            RectP rp;                                                // boxes.rs:24-66
            if (i < 2) rp = RectP{RT_RECT_XY, p0x, p1x, p0y, p1y, i == 0 ? p1z : p0z};
            else if (i < 4) rp = RectP{RT_RECT_XZ, p0x, p1x, p0z, p1z, i == 2 ? p1y : p0y};
            else rp = RectP{RT_RECT_YZ, p0y, p1y, p0z, p1z, i == 4 ? p1x : p0x};
            rect_record(rp, 0u, r, t, rec);
            break;
        }
        case RT_KIND_TRIANGLE: {
            const Vec3 a(q.r0.x, q.r0.y, q.r1.x), b(q.r1.y, q.r2.x, q.r2.y), c(q.r3.x, q.r3.y, q.r4.x);
            Vec3 n = rtm::to_unit(rtm::cross(b - a, c - a));
            Vec3 p = r.o + r.d * t;
            double a1 = a.x - b.x, b1 = a.x - c.x, c1 = a.x - p.x;
            double a2 = a.y - b.y, b2 = a.y - c.y, c2 = a.y - p.y;
            rec.u = (c1 * b2 - b1 * c2) / (a1 * b2 - b1 * a2);
            rec.v = (a1 * c2 - a2 * c1) / (a1 * b2 - b1 * a2);
            rec.p = p; rec.t = t;
            rec.set_face_normal(r.d, n);
            break;
        }
        case RT_KIND_RING: {
            rec.p = r.o + r.d * t; rec.t = t; rec.u = 0.0; rec.v = 0.0;
            rec.set_face_normal(r.d, Vec3(0.0, 1.0, 0.0));
            break;
        }
        default: {
            rec.p = r.o + r.d * t; rec.normal = Vec3(1.0, 0.0, 0.0); rec.t = t; rec.u = 0.0; rec.v = 0.0;
            rec.front_face = true;
            break;
        }
    }
    if (w.leaf & RT_REF_FLIP) rec.front_face = !rec.front_face;
    for (uint32_t lvl = w.chain.n; lvl > 0; lvl--) {
        XRay moved = ray_at_level(s, w.chain, lvl, world);
        xform_record(s, w.chain.at(lvl - 1), moved, rec);
    }
}
// Texture::value of a material's texture whose top-level record came with the material (MaterialDev): a SolidColor
// answers from registers; everything else goes the general way.
RT_DEV Vec3 texture_value_top(const SceneDev &s, uint32_t tex, uint32_t tex_kind, Vec3 color, double u, double v, Vec3 p) {
    if (tex_kind == RT_TEX_SOLID) return color;
    return texture_value(s, tex, u, v, p);
}

} // namespace

// =====================================================================================
// Shade pass.
// =====================================================================================
// Section clock of the shade pass (diagnostic build -DRT2022_SHADE_PROBE only; tools/shade_probe.sh): wave 0's lane 0 of
// every workgroup adds the wall-clock ticks it spent in each section to pool.dbg[64 + section].
#ifdef RT2022_SHADE_PROBE
#define SP_DECL unsigned long long sp_t = wall_clock64(), sp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define SP_MARK(i) do { unsigned long long sp_n = wall_clock64(); sp_acc[i] += sp_n - sp_t; sp_t = sp_n; } while (0)
#define SP_FLUSH() do { if (tid == 0 && pool.dbg) for (int sp_i = 0; sp_i < 10; sp_i++) atomicAdd(&pool.dbg[64 + sp_i], sp_acc[sp_i]); } while (0)
#else
#define SP_DECL do {} while (0)
#define SP_MARK(i) do {} while (0)
#define SP_FLUSH() do {} while (0)
#endif

#ifndef RT2022_SHADE_WAVES
#define RT2022_SHADE_WAVES 3           // resident shade workgroups per CU = waves per SIMD (168 VGPRs; four: 128 VGPRs, 51 spilled)
#endif
// RING: the partial-sum ring of RenderArgs::ring is in use (a build of its own: the default instance carries none of its
// bookkeeping — bounded claims, starved slots, the oldest item in flight).
template <bool STATS, bool RING = false>
__global__ void __launch_bounds__(kBlock, RT2022_SHADE_WAVES) wf_shade(const SceneDev s, const RenderArgs *__restrict__ ap, const WfPool pool, const uint32_t parity) {
    __shared__ uint32_t hist[SK_COUNT];
    __shared__ uint32_t cursor[SK_COUNT];
    __shared__ uint32_t sorted[S];
    __shared__ uint32_t n_sorted;
    __shared__ uint8_t new_kind[S];      // the slots' next state (| list class << 4), written back in one coalesced sweep
    __shared__ uint32_t bins[kListBins];
    __shared__ uint8_t new_oct[S];       // direction octant of the slot's next ray (second sort key of the list)
    __shared__ uint16_t fresh_q[S];      // slots that want a new path (| 0x8000: the slot holds an item whose state counts)
    static_assert(kSlotsPerBlock <= 32768 && kSlotsPerBlock % kBlock == 0, "a segment's slot index shares a u16 with one flag bit (fresh_q), and the sorts deal S / kBlock slots to every thread");
    static_assert(kSlotsPerBlock <= 65536, "`sorted` packs slot | kind << 16");
    __shared__ uint32_t n_fresh;
    const RenderArgs &a = *ap;
    const PoolView pv{pool};
    const uint32_t base = blockIdx.x * (uint32_t)S;
    const uint32_t tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    Counters<STATS> cnt;

    // The light list with its primitives' numbers, in LDS when it is short (it is one or two entries in every scene of
    // the reference): MixturePdf's two visits per bounce (generate + value, pdf.rs:94-104) then cost no memory round trip.
    constexpr uint32_t kLdsLights = 8;
    __shared__ LightRec lights_lds[kLdsLights];
    __shared__ PrimTable prim_tab;
    prim_table_fill(s, prim_tab, tid);
    if (tid < kLdsLights && tid < s.n_lights) lights_lds[tid] = fetch_light(s, tid);
    auto light_at = [&](uint32_t li) { return li < kLdsLights ? lights_lds[li] : fetch_light(s, li); };
    SP_DECL;
    if (tid < SK_COUNT) hist[tid] = 0;
    if (tid == 0) n_fresh = 0;
    __syncthreads();
    // Counting sort by kind of the slots that carried a ray through the trace pass: the entries of the segment's list
    // (written by the previous shade pass, or by wf_init: every slot in use, FRESH) with the kind the trace pass left
    // at the same position. Slots not on the list are idle. (Kinds live by list position, not by slot: the lanes of a
    // traversal wave take neighbouring entries, so their one-byte results land in the same cache lines at about the same
    // time instead of dirtying a line per byte all over the segment.)
    const uint32_t n_rays = pool.list_n[blockIdx.x] < (uint32_t)S ? pool.list_n[blockIdx.x] : (uint32_t)S;
    // (ring mode: behind the rays sit the slots that found the ring full last pass, kind FRESH: they ask again now)
    const uint32_t n_starved_in = RING ? (pool.starved_n[blockIdx.x] < (uint32_t)S - n_rays ? pool.starved_n[blockIdx.x] : (uint32_t)S - n_rays) : 0u;
    const uint32_t n_listed = n_rays + n_starved_in;
    uint32_t my_kind[S / kBlock], my_slot[S / kBlock];
#pragma unroll
    for (int i = 0; i < S / kBlock; i++) {
        const uint32_t e = (uint32_t)(i * kBlock) + tid;
        uint32_t k = SK_IDLE, ls = 0;
        if (e < n_listed) { k = pool.kind[base + e]; ls = pool.list[base + e]; }
        my_kind[i] = k; my_slot[i] = ls;
        new_kind[e] = (uint8_t)SK_IDLE;
        if (k != SK_IDLE) atomicAdd(&hist[k], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (uint32_t k = 0; k < SK_COUNT; k++) { cursor[k] = acc; acc += hist[k]; }
        n_sorted = acc;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < S / kBlock; i++) {
        uint32_t k = my_kind[i];
        if (k != SK_IDLE) sorted[atomicAdd(&cursor[k], 1u)] = my_slot[i] | (k << 16);
    }
    __syncthreads();
    SP_MARK(0);                                                      // 0: counting sort
    const uint32_t total = n_sorted;
    const Vec3 background = ld3(a.background);
    // One sample per work item (spp_chunk = 1): the item's running sum needs no place of its own in the pool.
    const bool single = a.chunk == 1 && a.spp > 0 && a.max_depth > 0;
    const bool small_job = a.n_items <= 0xFFFFFFFFull;
    const uint32_t step_shift = (a.node_quorum >> 20) & 0xFu;         // list class = expected steps >> shift (0 = slot order)

    unsigned long long my_oldest = ~0ull;                            // (ring mode) the oldest work item among this thread's paths that go on
    for (uint32_t j0 = 0; j0 < total; j0 += kBlock) {
        const uint32_t j = j0 + tid;
        const bool on = j < total;
        uint32_t slot = 0, kind = SK_IDLE;
        if (on) { uint32_t e = sorted[j]; slot = base + (e & 0xFFFFu); kind = e >> 16; }
        // Every slot that carried a ray has been through the trace pass by now. One that has not would lose its
        // path without a trace (it is neither shaded nor re-listed): report it instead — the render then fails.
        if (on && kind == SK_TRACE) atomicOr(pool.fault, 1u);
        bool alive = false;          // path continues with a new ray
        bool ended = false;          // path ended: add to pixel, start the next sample
        Ray r;
        Rng rng;
        Vec3 Lterm(0.0, 0.0, 0.0);
        SlotTape tape{pool.tape + (uint64_t)slot * pool.tape_cap * 4};
        // Everything the slot owns is fetched up front, side by side (the records are independent of
        // `kind`; a FRESH slot's are stale but mapped), instead of one latency after another.
        SlotState stt{};
        Winner w;
        w.t = 0.0; w.leaf = 0; w.face = 0; w.chain.n = 0; w.chain.c0 = w.chain.c1 = w.chain.c2 = w.chain.c3 = 0;
        uint32_t steps = 0;          // node steps of the ray that has just been traced
        uint32_t mat_word = 0;
        PrimRegs prim{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
        f64x2 md0{0.0, 0.0}, md1{0.0, 0.0}, md2{0.0, 0.0}, md3{0.0, 0.0}, md4{0.0, 0.0};      // MaterialDev
        if (on) {
            stt = load_state(pool, slot);
            if (kind > SK_MISS) {                                 // (a miss ends its path and a fresh slot has none: neither needs the ray or the winner)
                uint64_t rs;
                r = pv.load_ray(slot, rs);
                rng = Rng(rs);
                const u32x4 *hq = reinterpret_cast<const u32x4 *>(pool.hit + (uint64_t)slot * kRecWords);
                u32x4 ha = hq[0], hb = hq[1];
                t_pin(ha); t_pin(hb);
                bool have_mat;
                PoolView::decode_hit(ha, hb, w, steps, mat_word, have_mat);
                if (!have_mat) mat_word = leaf_material_word(s, w.leaf);      // (four movers deep: the chain needed the word's place)
                // second round trip, everything at once: the winning primitive's record and its material's
                const f64x2_a8 *pp = prim_address(prim_tab, w.leaf);
                const f64x2 *mp = reinterpret_cast<const f64x2 *>(s.materials_dev + (mat_word & kMatIndexMask));
                prim.r0 = pp[0]; prim.r1 = pp[1]; prim.r2 = pp[2]; prim.r3 = pp[3]; prim.r4 = pp[4];
                md0 = mp[0]; md1 = mp[1]; md2 = mp[2]; md3 = mp[3]; md4 = mp[4];
                t_pin(prim.r0); t_pin(prim.r1); t_pin(prim.r2); t_pin(prim.r3); t_pin(prim.r4);
                t_pin(md0); t_pin(md1); t_pin(md2); t_pin(md3); t_pin(md4);
            }
        }
        SP_MARK(1);                                                  // 1: slot, hit, primitive and material fetches
        // MaterialDev: tex, tex_kind | albedo | param | tex_color | tex_scale | tex_a, tex_b
        const uint32_t m_tex = (uint32_t)rtm::d2u(md0.x), m_tex_kind = (uint32_t)(rtm::d2u(md0.x) >> 32);
        const Vec3 m_albedo(md0.y, md1.x, md1.y), m_tex_color(md2.y, md3.x, md3.y);
        const double m_param = md2.x;
        // (top bit of the stored depth: some record of the path's tape is not "finite weight, pdf neither 0 nor NaN")
        uint32_t depth = stt.depth & 0x7FFFFFFFu;
        uint32_t tainted = stt.depth >> 31;
        // Expected length of the slot's next traversal, for the order of the trace pass's list: a bounce ray is taken
        // to resemble the ray before it; a new sample's camera ray goes with the short ones. (A per-slot record of the
        // previous camera ray's length predicts better, but costs a gather and a scatter per slot: measured -1.4 %.)
        // Ordering only: results never depend on it.
        uint32_t expect = steps;

        if (on && kind >= SK_MISS) {
            if (kind == SK_MISS) {
                Lterm = background;                                   // main.rs:275-276
                ended = true;
            } else {
                // (u, v) only matter to image textures (and to a checker that may select one).
                bool want_uv = false;
                const bool lambertian = kind >= SK_LAMB_SOLID && kind <= SK_LAMB_IMAGE;
                if (lambertian) {                                     // (the slot kind says which texture it is)
                    want_uv = kind == SK_LAMB_IMAGE || kind == SK_LAMB_CHECKER;
                } else if (kind == SK_LIGHT || kind == SK_ISOTROPIC) {
                    want_uv = m_tex_kind == RT_TEX_IMAGE || m_tex_kind == RT_TEX_CHECKER;
                }
                HitRec rec;
                winner_record_regs(s, r, w, prim, rec, want_uv);
                SP_MARK(2);                                           // 2: the winner's hit record
                if (kind == SK_LIGHT) {                               // emitted; scatter = None (material/mod.rs:16-18,174-180)
                    Lterm = rec.front_face ? texture_value_top(s, m_tex, m_tex_kind, m_tex_color, rec.u, rec.v, rec.p) : Vec3(0.0, 0.0, 0.0);
                    ended = true;
                } else {
                    Vec3 wgt;
                    double p = 1.0;
                    Vec3 dir;
                    double tm = r.tm;
                    if (lambertian) {                                 // material/mod.rs:51-65 + main.rs:263-271
                        Vec3 att = texture_value_top(s, m_tex, m_tex_kind, m_tex_color, rec.u, rec.v, rec.p);
                        rtm::Onb uvw = rtm::onb_from_w(rec.normal);
                        double cosv;
                        if (s.n_lights == 0) {                        // cosine-only mode
                            dir = uvw.local_vec(random_cosine_direction(rng));
                            cosv = rtm::dot(rtm::to_unit(dir), uvw.w);
                            p = cosv <= 0.0 ? 0.0 : cosv / rtm::PI;
                        } else {                                      // MixturePdf(lights, cos), pdf.rs:94-104
                            if (rng.gen_range(0.0, 1.0) < 0.5) dir = lights_random_of(s.n_lights, light_at, rec.p, rng);
                            else dir = uvw.local_vec(random_cosine_direction(rng));
                            double lp = lights_pdf_value_of<STATS>(s.n_lights, light_at, rec.p, dir, cnt);
                            cosv = rtm::dot(rtm::to_unit(dir), uvw.w);
                            double cp = cosv <= 0.0 ? 0.0 : cosv / rtm::PI;
                            p = 0.5 * lp + 0.5 * cp;
                        }
                        double cosine = rtm::dot(rec.normal, rtm::to_unit(dir));
                        double spdf = cosine < 0.0 ? 0.0 : cosine / rtm::PI;
                        wgt = att * spdf;
                    } else if (kind == SK_METAL) {                    // material/mod.rs:85-96
                        Vec3 reflected = rtm::reflect(rtm::to_unit(r.dir), rec.normal);
                        dir = reflected + random_in_unit_sphere(rng) * m_param;
                        wgt = m_albedo;
                        tm = 0.0;                                     // time = 0., mod.rs:91
                    } else if (kind == SK_DIELECTRIC) {               // material/mod.rs:120-147
                        double refraction_ratio = rec.front_face ? 1.0 / m_param : m_param;
                        Vec3 unit_direction = rtm::to_unit(r.dir);
                        double cos_theta = rtm::fmin_(rtm::dot(-unit_direction, rec.normal), 1.0);
                        double sin_theta = rtm::sqrt_(1.0 - cos_theta * cos_theta);
                        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
                        double random_double = rng.gen_range(0.0, 1.0);
                        dir = (cannot_refract || reflectance(cos_theta, refraction_ratio) > random_double)
                                  ? rtm::reflect(unit_direction, rec.normal)
                                  : rtm::refract(unit_direction, rec.normal, refraction_ratio);
                        wgt = Vec3(1.0, 1.0, 1.0);
                    } else {                                          // Isotropic, material/mod.rs:207-213
                        wgt = texture_value_top(s, m_tex, m_tex_kind, m_tex_color, rec.u, rec.v, rec.p);
                        dir = random_in_unit_sphere(rng);
                    }
                    uint32_t nb = a.max_depth - depth;
                    tape.put(nb, wgt, p);
                    tainted |= (t_finite_s(wgt.x) && t_finite_s(wgt.y) && t_finite_s(wgt.z) && p == p && p != 0.0) ? 0u : 1u;
                    r = Ray(rec.p, dir, tm);
                    depth--;
                    if (depth == 0) ended = true;                     // the next ray_color returns (0,0,0), main.rs:240-242
                    else alive = true;
                }
            }
            SP_MARK(3);                                               // 3: emitted / scatter / pdfs / tape record
            if (ended) {
                uint32_t nb = a.max_depth - depth;
                // Unwinding from an exact zero through records with finite weights and usable pdfs gives 0 + (w * 0) / p =
                // +0 at every step (a black background, a light seen from behind, an exhausted depth): the tape need not
                // be read. Anything else — a pdf of 0, an infinite weight: the reference's NaN pixels — is unwound.
                Vec3 Lp(0.0, 0.0, 0.0);
                if (!(nb >= 1 && !tainted && Lterm.x == 0.0 && Lterm.y == 0.0 && Lterm.z == 0.0)) Lp = tape.unwind(nb, Lterm);
                if (single) {                                         // the item's one sample: 0 + L goes straight to its place
                    double *o = a.partial + stt.item * 3;                 // (ring mode: its plane is sample mod R — worked out when the path began)
                    o[0] = 0.0 + Lp.x; o[1] = 0.0 + Lp.y; o[2] = 0.0 + Lp.z;   // pixel_color = 0; pixel_color += ..., main.rs:143,150
                } else {
                    double2 *ps = reinterpret_cast<double2 *>(pool.pixel_sum + (uint64_t)slot * 4);
                    double2 s0 = ps[0], s1 = ps[1];
                    ps[0] = make_double2(s0.x + Lp.x, s0.y + Lp.y);       // pixel_color += ..., main.rs:150
                    ps[1] = make_double2(s1.x + Lp.z, 0.0);
                }
            }
            cnt.draws(rng.draws);                                     // words drawn while scattering
            rng.draws = 0;
        }

        SP_MARK(4);                                                   // 4: unwinding and the pixel
        // A slot whose path has ended (or that never had one) gets its next path in the second sweep below, where all
        // such slots of the segment sit side by side: aiming a camera ray (three hashes, the lens rejection loop, five
        // divisions) is the longest stretch of this kernel, and here it would run for the fifth of the lanes that need it.
        const bool want_path = on && (kind == SK_FRESH || ended);
        {
            const unsigned long long wm = wballot(want_path);
            if (wm) {
                const int leader = __ffsll((long long)wm) - 1;
                uint32_t qbase = 0;
                if ((int)lane == leader) qbase = atomicAdd(&n_fresh, (uint32_t)__popcll(wm));
                qbase = (uint32_t)__shfl((int)qbase, leader);
                if (want_path) fresh_q[qbase + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull))] =
                    (uint16_t)((slot - base) | ((kind != SK_FRESH && !single) ? 0x8000u : 0u));
            }
        }

        if (RING && on && alive) { const unsigned long long grp = (stt.smp - 1u) / a.ring_group; my_oldest = grp < my_oldest ? grp : my_oldest; }      // (smp - 1: the sample in flight)
        if (on && alive) {
            cnt.ray();                                                // world.hit(r, 0.001, f64::MAX), main.rs:243
            pv.store_ray(slot, r, rng.s);
            stt.depth = depth | (tainted << 31);
            store_state(pool, slot, stt, false);
            uint32_t cls = step_shift ? (expect >> step_shift) : 0u;
            new_kind[slot - base] = (uint8_t)(SK_TRACE | ((cls > 15u ? 15u : cls) << 4));
            // (third key, RT2022_LIST_ORIGIN: what the ray starts from — a sphere, a box / rect, a medium)
            const uint32_t lk = RT_REF_KIND(w.leaf);
            const uint32_t org = RT2022_LIST_ORIGIN == 2 ? (lk & 7u)
                               : RT2022_LIST_ORIGIN ? ((lk == RT_KIND_SPHERE || lk == RT_KIND_MOVING_SPHERE) ? 1u : (lk == RT_KIND_BOX || lk == RT_KIND_RECT) ? 2u : lk == RT_KIND_MEDIUM ? 3u : 0u) : 0u;
            const double adx = rtm::fabs_(r.dir.x), ady = rtm::fabs_(r.dir.y), adz = rtm::fabs_(r.dir.z);
            const uint32_t quad = RT2022_LIST_SPATIAL == 2 ? ((adx >= ady && adx >= adz) ? 0u : (ady >= adz ? 1u : 2u))
                                : RT2022_LIST_SPATIAL ? ((r.orig.x < a.split[0] ? 1u : 0u) | (r.orig.z < a.split[2] ? 2u : 0u)) : 0u;
            new_oct[slot - base] = (uint8_t)((RT2022_LIST_OCTANTS ? ((r.dir.x < 0.0 ? 1u : 0u) | (r.dir.y < 0.0 ? 2u : 0u) | (r.dir.z < 0.0 ? 4u : 0u)) : 0u) | (org << 3) | (quad << 6));
        }
    }

    SP_MARK(5);                                                      // 5: queueing, stores of the bounce
    // Second sweep: the next sample of the item, or the next item (main.rs:140-152), for every slot that asked.
    // (The barrier also makes the first sweep's pixel sums visible to whichever thread finishes the item here.)
    __syncthreads();
    const uint32_t n_want = n_fresh;
    // One sample per item: every slot on the queue takes a new item, so the segment claims them with ONE atomic instead
    // of one per wave and sweep turn (RT2022_ITEM_BATCH; which slot gets which item changes nothing, §5 of DESIGN.md).
#ifndef RT2022_ITEM_BATCH
#define RT2022_ITEM_BATCH 1
#endif
    __shared__ unsigned long long seg_items;
    __shared__ uint32_t seg_take;        // (ring mode) how many of the n_want items the segment really got
    __shared__ uint32_t seg_more;        // (ring mode) 1: work items remain beyond the ring's limit — the slots left without one ask again
    const bool batch = (RT2022_ITEM_BATCH || RING) && single;
    if (batch) {
        if (tid == 0) {
            if (!RING) {
                seg_items = n_want ? atomicAdd(a.work_counter, (unsigned long long)n_want) : 0ull;
            } else {
                // Never USE an item beyond *claim_limit: sample c + R of a pixel shares its plane with sample c, which the host must
                // have added to the output first (it raises the limit behind the planes it consumes, between passes).
                // One add, like the plain path (a compare-and-swap loop on one word shared by thousands of segments fails most
                // of its tries: measured, a quarter of the frame); what lies beyond the limit is handed back. While a segment's
                // surplus is out, other segments may see the counter too high and take nothing this pass — never too much: an
                // item is only ever used by the segment whose add returned it, and only below the limit.
                const unsigned long long lim = *a.claim_limit;
                const unsigned long long old = n_want ? atomicAdd(a.work_counter, (unsigned long long)n_want) : 0ull;
                const uint32_t take = old < lim ? (uint32_t)((unsigned long long)n_want < lim - old ? (unsigned long long)n_want : lim - old) : 0u;
                const bool bound = lim < a.n_items;                 // the ring, not the end of the work, is what stops claims
                if (bound && take < n_want) atomicAdd(a.work_counter, 0ull - (unsigned long long)(n_want - take));      // (minus: modulo 2^64)
                seg_items = old; seg_take = take;
                seg_more = bound ? 1u : 0u;                          // (at lim == n_items nothing is handed back: beyond it the work IS done)
            }
        }
        __syncthreads();
    }
    for (uint32_t j0 = 0; j0 < n_want; j0 += kBlock) {
        const uint32_t j = j0 + tid;
        const bool on = j < n_want;
        const uint32_t e = on ? (uint32_t)fresh_q[j] : 0u;
        const uint32_t slot = base + (e & 0x7FFFu);                   // (bit 15 is the flag; a segment holds at most 32768 slots)
        bool alive = false;
        bool starved = false;        // (ring mode) wanted a work item, found the ring full: asks again next pass
        Ray r;
        Rng rng;
        uint32_t depth = 0;
        SlotState stt{};
        if (on) {
            // (one-sample items: the item is always finished, nothing of the old state is needed)
            bool have_item = (e & 0x8000u) != 0;
            if (have_item) stt = load_state(pool, slot);
            uint32_t smp = have_item ? stt.smp : 0u, smp_end = have_item ? stt.smp_end : 0u;
            for (int guard = 0; guard < 1 << 20; guard++) {           // loops only through degenerate items (spp or depth 0)
                bool need = !have_item || smp == smp_end;
                if (need && have_item) {                              // section_pixel_color.push(pixel_color), main.rs:152
                    if (!single) {
                        double *o = a.partial + stt.item * 3;
                        const double *ps = pool.pixel_sum + (uint64_t)slot * 4;
                        o[0] = ps[0]; o[1] = ps[1]; o[2] = ps[2];
                    }
                    have_item = false;
                }
                unsigned long long m = wballot(need);
                if (m) {
                    int leader = __ffsll((long long)m) - 1;
                    unsigned long long wbase = 0;
                    if (!batch) {
                        if ((int)lane == leader) wbase = atomicAdd(a.work_counter, (unsigned long long)__popcll(m));
                        wbase = __shfl(wbase, leader);
                    }
                    if (need) {
                        unsigned long long item = batch ? seg_items + j : wbase + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                        if (RING && j >= seg_take) {                    // the ring (or the work) ran out before this slot's turn
                            item = a.n_items;
                            starved = seg_more != 0u;
                        }
                        if (item < a.n_items) {
                            uint64_t pix_slot, yi;
                            uint32_t chunk_id, px;
                            if (RING) {                                 // group-major: item = (group * n_pixels + pixel) * ring_group + sample in the group
                                if (small_job) {
                                    const uint32_t q32 = (uint32_t)item / a.ring_group, np32 = (uint32_t)a.n_pixels;     // group * n_pixels + pixel
                                    const uint32_t g32 = q32 / np32, ps32 = q32 - g32 * np32, y32 = ps32 / a.width;
                                    chunk_id = g32 * a.ring_group + ((uint32_t)item - q32 * a.ring_group);
                                    px = ps32 - y32 * a.width;
                                    pix_slot = ps32; yi = y32;
                                } else {
                                    const uint64_t q = item / a.ring_group, g = q / a.n_pixels;
                                    chunk_id = (uint32_t)(g * a.ring_group + (item - q * a.ring_group));
                                    pix_slot = q - g * a.n_pixels;
                                    yi = pix_slot / a.width;
                                    px = (uint32_t)(pix_slot - yi * a.width);
                                }
                            } else if (small_job) {                     // (32-bit divisions where everything fits: the usual case)
                                const uint32_t ps32 = (uint32_t)item / a.n_chunks, y32 = ps32 / a.width;
                                chunk_id = (uint32_t)item - ps32 * a.n_chunks;
                                px = ps32 - y32 * a.width;
                                pix_slot = ps32; yi = y32;
                            } else {
                                pix_slot = item / a.n_chunks;
                                chunk_id = (uint32_t)(item - pix_slot * a.n_chunks);
                                yi = pix_slot / a.width;
                                px = (uint32_t)(pix_slot - yi * a.width);
                            }
                            uint32_t g = a.row_ids[yi];
                            uint32_t frame = g / a.height;
                            uint32_t py = g - frame * a.height;
                            smp = chunk_id * a.chunk;
                            smp_end = smp + a.chunk < a.spp ? smp + a.chunk : a.spp;
                            stt.item = (uint64_t)(RING ? chunk_id % a.ring : chunk_id) * a.n_pixels + pix_slot;     // (kept as the item's place in the partial sums: ring mode, plane = sample mod R)
                            stt.px = px; stt.py = py; stt.frame = frame;
                            if (!single) {
                                double2 *ps = reinterpret_cast<double2 *>(pool.pixel_sum + (uint64_t)slot * 4);
                                ps[0] = make_double2(0.0, 0.0);
                                ps[1] = make_double2(0.0, 0.0);
                            }
                            have_item = true;
                        }
                    }
                }
                if (!have_item) break;                                // no work left: the slot goes idle
                if (smp == smp_end) continue;                         // empty chunk (spp == 0): store zeros next turn
                uint32_t px = stt.px, py = stt.py, frame = stt.frame;
                uint64_t pixel = (uint64_t)py * a.width + px;
                rng = Rng(rtm::path_key(a.seed, frame, pixel, smp));  // main.rs:144-149
                double rand_u = rng.gen_f64();
                double rand_v = rng.gen_f64();
                double u = ((double)px + rand_u) / (double)(a.width - 1);
                double v = ((double)py + rand_v) / (double)(a.height - 1);
                r = get_ray(a.cam, u, v, rng);
                depth = a.max_depth;
                smp++;
                cnt.path();
                cnt.draws(rng.draws);                                 // words drawn while aiming the camera ray
                rng.draws = 0;
                if (depth == 0) continue;                             // MAX_DEPTH == 0: black at once
                alive = true;
                break;
            }
            stt.smp = smp;
            stt.smp_end = smp_end;
            if (alive) {
                cnt.ray();                                            // world.hit(r, 0.001, f64::MAX), main.rs:243
                pv.store_ray(slot, r, rng.s);
                stt.depth = depth;                                    // (a fresh tape: nothing tainted)
                store_state(pool, slot, stt, true);
                new_kind[slot - base] = (uint8_t)SK_TRACE;            // (a camera ray goes with the short ones: list class 0)
                new_oct[slot - base] = (uint8_t)(RT2022_LIST_OCTANTS ? ((r.dir.x < 0.0 ? 1u : 0u) | (r.dir.y < 0.0 ? 2u : 0u) | (r.dir.z < 0.0 ? 4u : 0u)) : 0u);
                if (RING) { const unsigned long long grp = (smp - 1u) / a.ring_group; my_oldest = grp < my_oldest ? grp : my_oldest; }
            } else if (RING && starved) {
                new_kind[slot - base] = (uint8_t)SK_FRESH;            // (not a ray: listed behind the rays, see below)
            }
        }
    }
    SP_MARK(6);                                                      // 6: second sweep (new paths)
    // Paths handed to the trace pass (the host stops when the whole pool reports none).
    for (uint32_t k = tid; k < kListBins; k += kBlock) bins[k] = 0;
    __shared__ uint32_t list_total, n_starved_out;
    __shared__ unsigned long long seg_oldest;
    if (RING && tid == 0) { n_starved_out = 0; seg_oldest = ~0ull; }
    __syncthreads();
    // The segment's ray list, longest expected traversal first (counting sort, 16 classes): the stragglers of
    // the trace pass then start early instead of keeping a few lanes busy after the list has run dry.
    uint32_t my_key[S / kBlock];
    uint32_t my_starved[RING ? S / kBlock : 1];                        // (ring mode) place among the segment's starved slots, or none
    if (RING) {                                                        // the oldest work item in flight, over the segment
        unsigned long long v = my_oldest;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(v, d); v = o < v ? o : v; }
        if (lane == 0 && v != ~0ull) atomicMin(&seg_oldest, v);
    }
#pragma unroll
    for (int i = 0; i < S / kBlock; i++) {
        uint32_t e = new_kind[i * kBlock + tid];
        uint32_t key = kListBins;                                      // carries no ray
        if (RING) my_starved[i] = e == (uint32_t)SK_FRESH ? atomicAdd(&n_starved_out, 1u) : 0xFFFFFFFFu;
        // (second key: rays that point into the same octant meet the boxes in a similar pattern, and the lanes of a
        // wave draw neighbouring list entries)
        // (new_oct: octant | origin class << 3 | quadrant << 6 — contiguous fields when every key is in use)
        if ((e & 0xFu) == SK_TRACE) {
            const uint32_t o8 = new_oct[i * kBlock + tid];
#ifndef RT2022_LIST_MAJOR
#define RT2022_LIST_MAJOR 1            // 0: expected length first, then origin and octant; 1: origin and octant first, length within; 2: octant, origin, length
#endif
            if (RT2022_LIST_MAJOR == 1) key = ((o8 >> 6) * (8u * kOriginClasses) + (o8 & 63u)) * 16u + (15u - (e >> 4));
            else if (RT2022_LIST_MAJOR == 2) key = (((o8 >> 6) * 8u + (o8 & 7u)) * kOriginClasses + ((o8 >> 3) & 7u)) * 16u + (15u - (e >> 4));
            else key = ((15u - (e >> 4)) * kSpatialClasses + (o8 >> 6)) * (8u * kOriginClasses) + (o8 & 63u);
            atomicAdd(&bins[key], 1u);
        }
        my_key[i] = key;
    }
    __syncthreads();
    // Exclusive prefix sums of the bins, in place: every thread takes kPer consecutive bins; scan over the wave by
    // shuffles, over the four waves through LDS.
    constexpr uint32_t kPer = kListBins > (uint32_t)kBlock ? kListBins / (uint32_t)kBlock : 1u;
    static_assert(kPer * (uint32_t)kBlock >= kListBins, "bins per thread");
    __shared__ uint32_t wave_tot[kBlock / 64];
    {
        uint32_t v[kPer], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) { const uint32_t k = tid * kPer + j; v[j] = k < kListBins ? bins[k] : 0u; sum += v[j]; }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, d); if ((int)lane >= d) inc += t; }
        if (lane == 63) wave_tot[tid >> 6] = inc;
        __syncthreads();
        uint32_t excl = inc - sum;
        for (uint32_t wv = 0; wv < (tid >> 6); wv++) excl += wave_tot[wv];
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) { const uint32_t k = tid * kPer + j; if (k < kListBins) bins[k] = excl; excl += v[j]; }
    }
    if (tid == 0) {
        uint32_t acc = 0;
        for (int wv = 0; wv < kBlock / 64; wv++) acc += wave_tot[wv];
        pool.list_n[blockIdx.x] = acc;
        // Rays handed on by this pass (the host stops a group when a pass reports none). Two counters take
        // turns, so each pass can clear the one the next pass will add to.
        uint32_t going = acc;
        if (RING) {                                                    // (slots waiting for the ring keep the frame going too)
            list_total = acc;
            pool.starved_n[blockIdx.x] = n_starved_out;
            going += n_starved_out;
            if (seg_oldest != ~0ull) atomicMin(&pool.oldest[parity], seg_oldest);
            if (blockIdx.x == 0) pool.oldest[parity ^ 1u] = ~0ull;
        }
        if (going) atomicAdd(&pool.n_active[parity], going);
        if (acc) atomicMax(&pool.max_list[parity], acc);
        if (blockIdx.x == 0) { pool.n_active[parity ^ 1u] = 0; pool.max_list[parity ^ 1u] = 0; *pool.next_chunk = 0; }
    }
    __syncthreads();
    if (RING) {
#pragma unroll
        for (int i = 0; i < S / kBlock; i++)
            if (my_starved[i] != 0xFFFFFFFFu) {
                const uint32_t pos = base + list_total + my_starved[i];
                pool.list[pos] = (uint16_t)((uint32_t)(i * kBlock) + tid);
                pool.kind[pos] = (uint8_t)SK_FRESH;
            }
    }
#pragma unroll
    for (int i = 0; i < S / kBlock; i++)
        if (my_key[i] < kListBins) {
            const uint32_t pos = base + atomicAdd(&bins[my_key[i]], 1u);
            pool.list[pos] = (uint16_t)((uint32_t)(i * kBlock) + tid);
            pool.kind[pos] = (uint8_t)SK_TRACE;                       // until the trace pass has been there
        }
    if (STATS) cnt.flush_wave(a.stats);
    SP_MARK(7);                                                      // 7: kinds written back, ray list built
    SP_FLUSH();
}

// =====================================================================================
// Trace pass: closest hit of every pending ray, in-wave scheduled.
// =====================================================================================
namespace {

struct TLane {
    XRay cur;              // ray inside the enclosing movers
    Vec3 inv;              // 1 / cur.d   (aabb.rs:19, hoisted: same value at every node)
    double a_len;          // cur.d.length_sqr()  (sphere.rs:41, hoisted likewise)
    double tm;
    double closest;
    // ConstantMedium::hit asks its boundary two closest-hit questions of its own
    // (constantmedium.rs:50-51). They run through the same operations as the main query,
    // against (t_lo, sub_closest) instead of (t_min, closest), and never touch the winner.
    double t_lo;           // lower bound in force: a.t_min, or the sub-query's
    double sub_closest;
    double med_t1;
    uint32_t med_ref;      // the medium being evaluated (0 = none: main query)
    Rng rng;
    Chain ctx;
    Chain win_chain;
    uint32_t win_leaf, win_face;
    uint32_t win_mat;      // material word of the winning leaf (index | slot kind << kMatKindShift), taken from the record at hand
    double stash_ix, stash_iz;   // 1/d.x, 1/d.z of the frame a RotateY was entered from (they change only there) ...
    uint32_t stash_level;        // ... and that frame's mover depth (0xFFFFFFFF: nothing stashed)
    // (node table in LDS, RT2022_SIGNED_SLABS) byte addresses, within the table's record 0, of the box coordinate the ray meets
    // first / last on each axis: bmin / bmax by the sign of 1/d — set wherever inv is (t_slabs)
    uint32_t near_at[3], far_at[3];
    // (single-precision slab test, RT2022_F32_SLABS) per axis {(float)(1/d), (float)(-o/d)} — one operand pair of the packed
    // multiply-add that gives the axis' two slab distances — and the ray's share of the test's error bound; set with near_at
    f32x2 p32[3];
    float e_ray;
    uint32_t slot;
    uint32_t entry;        // where on the ray list the slot was found (its kind goes back to the same place)
    uint32_t steps;        // node steps of this ray
    int sp;
    uint32_t top, op;
    // Per-lane flags in ONE register rather than three bools: a bool member lives as a lane mask in a scalar register
    // pair, and every join of the scheduler's control flow then merges each of them with three scalar instructions
    // (seen in the ISA: ~30 per round of the outer loop); a vector register needs no merging.
    //   kPlain     the fast node step applies to this ray (see there)
    //   kHasRay    the lane carries a ray
    //   kSubFound  the medium sub-query in progress has found a boundary hit
    //   kNeed64    the single-precision slab test could not decide the node step at hand: the voted node arm takes it in double precision
    uint32_t flags;
};

template <int STACK, int WG = kBlock>
struct TStack {
    uint32_t *col;
    RT_DEV void push(TLane &L, uint32_t ref) { if (L.sp < STACK) { col[L.sp * WG] = ref; L.sp++; } }
    RT_DEV uint32_t pop(TLane &L) { if (L.sp > 0) { L.sp--; return col[L.sp * WG]; } return REF_EMPTY; }
};

constexpr uint32_t kPlain = 1u, kHasRay = 2u, kSubFound = 4u, kNeed64 = 8u;
RT_DEV void t_flag(TLane &L, uint32_t bit, bool on) { L.flags = on ? (L.flags | bit) : (L.flags & ~bit); asm volatile("" : "+v"(L.flags)); }
RT_DEV bool t_finite(double x) { return (rtm::d2u(x) & 0x7FF0000000000000ull) != 0x7FF0000000000000ull; }
// The fast node step applies (see there): every 1/d finite and non-zero, origin finite, boxes plain.
RT_DEV void t_flags(TLane &L, bool boxes_plain) {
    const bool plain = boxes_plain && t_finite(L.inv.x) && t_finite(L.inv.y) && t_finite(L.inv.z) && L.inv.x != 0.0 && L.inv.y != 0.0 &&
              L.inv.z != 0.0 && t_finite(L.cur.o.x) && t_finite(L.cur.o.y) && t_finite(L.cur.o.z);
    t_flag(L, kPlain, plain);
}
// For a plain ray the slab test's min(t0, t1) / max(t0, t1) per axis IS the choice of bmin or bmax by the sign of 1/d (the
// products are ordered by it: see the fast path) — made here once per direction instead of twice per axis and node step.
RT_DEV void t_slabs(TLane &L, uint32_t table_at) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const bool neg = L.inv[i] < 0.0;
        L.near_at[i] = table_at + (neg ? 24u : 0u) + 8u * (uint32_t)i;          // record: bmin x y z at +0 +8 +16, bmax at +24 +32 +40
        L.far_at[i] = table_at + (neg ? 0u : 24u) + 8u * (uint32_t)i;
    }
}
// The single-precision slab test of the node table in LDS (RT2022_F32_SLABS; see the fast path). A node record there is eleven
// words: per axis {(float)bmin, (float)bmax, (float)bmin} — so that ONE two-word read at `base` or at `base + 4` delivers the
// pair in the order (first met, last met) for either sign of 1/d — then the left child and the push ref.
constexpr uint32_t kNode32Words = 11, kNode32Bytes = 4 * kNode32Words;
// Error bound (u = 2^-24). With b32 = (float)b, i32 = (float)(1/d), n32 = (float)(-o * (1/d)) the kernel computes
// t32 = fma(b32, i32, n32) where the double-precision step computes T = (b - o) * (1/d), rounded twice. Against the real
// number R = b/d - o/d (1/d being the f64 value both use):
//   |b32 i32 - b/d| <= |b/d| (2u + u^2),   |n32 + o/d| <= |o/d| (u + 2^-52),   the fma rounds once: u (1 + u) |t32|,   |T - R| <= 2^-52 |R|,
// and with |b/d| <= |R| + |o/d|, |R| <= |t32| + error, |o/d| <= |n32| (1 + u):
//   |t32 - T| <= 3.000001 u (|t32| + |n32|)
// — an error relative to the VALUE plus a constant of the ray, k = 3.000001 u max |n32|; nothing in it depends on how large the
// scene's other coordinates are. (A bound from the largest box coordinate instead was tried first: with 0.2-unit spheres on
// a 2000-unit ground it left a few per cent of the node steps undecided, and the kernel 10 % slower than the double-precision
// one.) The window's two ends, converted to float, are off by u of their value: the same form. x -> x + c|x| and x -> x - c|x|
// are increasing, so the max / min of such values is off by at most c |max| + k, and the rounded difference of the two by
//   (3.000001 u + u) (|tmx32| + |tmn32|) + 6.000002 u max |n32|   <   4.5 u (|tmx32| + |tmn32|) + e_ray,   e_ray = 6.5 u max |n32| + 2e-8
// (2e-8 for box coordinates below the float normal range, see t_slabs32; the eighths of slack cover the three roundings of the bound's own arithmetic, 3 u each at most).
// It has to be this tight: a ray that leaves a surface tests the boxes that surface lies on the face of, where the verdict hangs
// on t_min = 0.001 against a distance of zero — with coordinates in the hundreds the bound is a few 1e-4 of that.
// Any overflow on the way (1/d beyond f32) makes a value or the bound infinite or NaN: the test then decides nothing and
// the lane takes the double-precision step.
constexpr float kF32RelBound = 4.5f * 0x1p-24f, kF32RayBound = 6.5f * 0x1p-24f;
RT_DEV void t_slabs32(TLane &L, uint32_t table_at) {
    float e = 0.0f;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const bool neg = L.inv[i] < 0.0;
        L.near_at[i] = table_at + 12u * (uint32_t)i + (neg ? 4u : 0u);
        const float i32 = (float)L.inv[i], n32 = (float)(-(L.cur.o[i] * L.inv[i]));
        L.p32[i] = (f32x2){i32, n32};
        e = __builtin_fmaxf(e, __builtin_fabsf(n32));           // (n32 is no NaN for a plain ray — finite origin, finite non-zero 1/d — and only those take the test)
        // The analysis above takes i32 to be 1/d within u, and a float box coordinate within u of the double — or within 1.2e-38
        // of it, for a coordinate below the normal range: 1/d between 1e-30 and 1e30 makes the first true and keeps what the
        // second adds below the 2e-8 of e_ray; directions outside that range leave every step to the double-precision test.
        const float ai = __builtin_fabsf(i32);
        ok = ok && ai >= 1e-30f && ai <= 1e30f;
    }
    L.e_ray = ok ? __builtin_fmaf(e, kF32RayBound, 2e-8f) : __builtin_inff();
}
// The same for the kernels that fetch their single-precision records from HBM (kF32G): no per-lane table addresses, the record's
// {min, max} pairs are ordered after the multiply-adds instead.
RT_DEV void t_slabs32g(TLane &L) {
    float e = 0.0f;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float i32 = (float)L.inv[i], n32 = (float)(-(L.cur.o[i] * L.inv[i]));
        L.p32[i] = (f32x2){i32, n32};
        e = __builtin_fmaxf(e, __builtin_fabsf(n32));
        const float ai = __builtin_fabsf(i32);
        ok = ok && ai >= 1e-30f && ai <= 1e30f;
    }
    L.e_ray = ok ? __builtin_fmaf(e, kF32RayBound, 2e-8f) : __builtin_inff();
}
RT_DEV void t_set_cur(TLane &L, const XRay &c, bool boxes_plain) {
    L.cur = c;
    L.inv = Vec3(1.0 / c.d.x, 1.0 / c.d.y, 1.0 / c.d.z);
    L.a_len = c.d.length_sqr();
    t_flags(L, boxes_plain);
}
RT_DEV double t_hi(const TLane &L) { return L.med_ref ? L.sub_closest : L.closest; }
RT_DEV void t_accept(TLane &L, double t, uint32_t face, uint32_t mat_word) {
    if (L.med_ref) { L.sub_closest = t; t_flag(L, kSubFound, true); return; }
    L.closest = t;
    L.win_leaf = L.top; L.win_face = face; L.win_chain = L.ctx; L.win_mat = mat_word;
}
// The two boundary queries of ConstantMedium::hit on ONE sphere (constantmedium.rs:50-51):
//     sphere_t(center, radius, r, a, -inf, +inf, t1)  and then  sphere_t(center, radius, r, a, t1 + 0.0001, +inf, t2)
// with what they share — oc, half_b, c, the discriminant, its square root and the near root — computed once. Every
// expression and comparison is sphere_t's own (pt_common.hpp, sphere.rs:39-58), so the values are the same bit for bit;
// `first` says whether the first query found a hit (the second is only made, and counted, then).
RT_DEV bool sphere_t_twice(Vec3 center, double radius, const XRay &r, double a, double &t1, double &t2, bool &first) {
    first = false;
    Vec3 oc = r.o - center;
    double half_b = rtm::dot(oc, r.d);
    double c = oc.length_sqr() - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = rtm::sqrt_(discriminant);
    const double near_root = (-half_b - sqrtd) / a;
    double root = near_root;
    if (root < -rtm::INF || rtm::INF < root) {
        root = (-half_b + sqrtd) / a;
        if (root < -rtm::INF || rtm::INF < root) return false;
    }
    t1 = root;
    first = true;
    const double t_min2 = t1 + 0.0001;
    root = near_root;
    if (root < t_min2 || rtm::INF < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min2 || rtm::INF < root) return false;
    }
    t2 = root;
    return true;
}
// Boxes::hit over six sides given by value (boxes.rs:24-66,80-82 + mod.rs:90-100): box_t of pt_common.hpp, fed from
// registers.
RT_DEV bool t_box(double p0x, double p0y, double p0z, double p1x, double p1y, double p1z, const XRay &r, double t_min, double t_max,
                  double &t, uint32_t &face) {
    bool any = false;
    double closest = t_max, tt;
    if (rect_t(RT_RECT_XY, p0x, p1x, p0y, p1y, p1z, r, t_min, closest, tt)) { closest = tt; face = 0; any = true; }
    if (rect_t(RT_RECT_XY, p0x, p1x, p0y, p1y, p0z, r, t_min, closest, tt)) { closest = tt; face = 1; any = true; }
    if (rect_t(RT_RECT_XZ, p0x, p1x, p0z, p1z, p1y, r, t_min, closest, tt)) { closest = tt; face = 2; any = true; }
    if (rect_t(RT_RECT_XZ, p0x, p1x, p0z, p1z, p0y, r, t_min, closest, tt)) { closest = tt; face = 3; any = true; }
    if (rect_t(RT_RECT_YZ, p0y, p1y, p0z, p1z, p1x, r, t_min, closest, tt)) { closest = tt; face = 4; any = true; }
    if (rect_t(RT_RECT_YZ, p0y, p1y, p0z, p1z, p0x, r, t_min, closest, tt)) { closest = tt; face = 5; any = true; }
    t = closest;
    return any;
}
// L.top has just been set: label it. The two cheap steps of ConstantMedium::hit — start the first
// boundary query, turn the first into the second (constantmedium.rs:50-51) — are taken on the spot
// instead of costing the wave a scheduling round each; only the finish (RNG, log) is an operation.
template <int STACK, bool STATS, unsigned FEAT, int WG>
RT_DEV void t_settle(const SceneDev &s, TLane &L, TStack<STACK, WG> &st, double t_min, Counters<STATS> &cnt) {
    if (FEAT & kFeatVolumes) {
        for (int guard = 0; guard < 6; guard++) {
            if (RT_REF_KIND(L.top) == RT_KIND_MEDIUM) {               // a medium leaf: boundary.hit(r, -inf, inf)
                // (a boundary that is one plain sphere — the fog and the subsurface ball of the final scene —
                // is not traversed at all: the medium arm does both queries and the finish in one turn)
                if (s.media_mode == 1u || (s.media_mode == 2u && s.media_dev[RT_REF_INDEX(L.top)].sphere_boundary)) break;
                cnt.prim(RT_KIND_MEDIUM);
                L.med_ref = L.top;
                L.t_lo = -rtm::INF;
                L.sub_closest = rtm::INF; t_flag(L, kSubFound, false);
                st.push(L, REF_MED1);
                L.top = s.media_dev[RT_REF_INDEX(L.top)].boundary;
            } else if (L.top == REF_MED1) {
                if (L.flags & kSubFound) {                            // boundary.hit(r, rec1.t + 0.0001, inf)
                    L.med_t1 = L.sub_closest;
                    L.t_lo = L.med_t1 + 0.0001;
                    L.sub_closest = rtm::INF; t_flag(L, kSubFound, false);
                    st.push(L, REF_MED2);
                    L.top = s.media_dev[RT_REF_INDEX(L.med_ref)].boundary;
                } else {
                    L.med_ref = 0; L.t_lo = t_min;
                    L.top = st.pop(L);
                }
            } else {
                break;
            }
        }
    }
    L.op = classify(L.top);
}
#define T_NEXT() do { L.top = st.pop(L); t_settle<STACK, STATS, FEAT, WG>(s, L, st, t_min, cnt); } while (0)
#define T_SETTLE() t_settle<STACK, STATS, FEAT, WG>(s, L, st, t_min, cnt)

} // namespace

// Census of the single-precision slab test (diagnostic build -DRT2022_F32_CENSUS only): node steps of the fast path that took
// it, how many of them it left to the double-precision test, and how many of its verdicts differed from that test's (the census
// build makes both): read and cleared by f32_slab_census (rt_debug_f32_slabs).
__device__ unsigned long long g_f32_census[3];

// Phase clock of the traversal kernel (diagnostic build -DRT2022_TRACE_PROBE only): every wave adds the shader-clock
// ticks it spent in each phase of the scheduler — [0] node fast path, [1] vote, [2..9] the voted arms by label (node,
// sphere, rect, box, medium, misc, ctx, done), [10] the rest — to pool.dbg[96 + phase]; printed after the render.
#ifdef RT2022_TRACE_PROBE
#define TP_DECL __shared__ unsigned long long tp_lds[WG / 64][12]; unsigned long long tp_t = __builtin_readcyclecounter(); \
    if (lane < 12) tp_lds[tid >> 6][lane] = 0
#define TP_MARK(i) do { const unsigned long long tp_n = __builtin_readcyclecounter(); const unsigned long long tp_m = wballot(true); \
    if ((int)lane == __ffsll((long long)tp_m) - 1) tp_lds[tid >> 6][(i)] += tp_n - tp_t; tp_t = tp_n; } while (0)
#define TP_FLUSH() do { if (lane < 12 && pool.dbg) atomicAdd(&pool.dbg[96 + lane], tp_lds[tid >> 6][lane]); } while (0)
#else
#define TP_DECL do {} while (0)
#define TP_MARK(i) do {} while (0)
#define TP_FLUSH() do {} while (0)
#endif

// Resident traversal workgroups per CU a variant is built and launched for (= waves per SIMD = its VGPR budget):
// the sphere-only kernel needs 82 VGPRs and runs five (C2: +4 % over four; six would spill), the full kernels four.
#ifndef RT2022_LEAN_BLOCKS
#define RT2022_LEAN_BLOCKS 5           // resident workgroups per CU of the sphere-only kernels (FEAT = 0, 256 threads)
#endif
constexpr int trace_blocks_per_cu(int stack, bool stats, unsigned feat) {
    return stack > 32 ? 2 : stats ? 3 : (stack > kStackSmall || (feat & kFeatMisc)) ? 4 : feat == 0 ? RT2022_LEAN_BLOCKS : kTraceBlocksPerCU;
}
// The node-cache variant (WG = kCacheBlock threads, one workgroup per CU, CACHE = kNodeCache records): the BVH's first
// CACHE node records live in LDS — 48 bytes of box and 8 of child refs each — beside the traversal stacks of the
// workgroup's 16 waves. A node step on a cached node is an LDS round trip instead of an L1 / L2 one; the 160 KiB of a
// CU belong to ONE workgroup, so the table exists once per CU rather than once per four waves. Two instances: the
// whole node table of a small scene (PARTIAL = false: no HBM path for nodes at all), and the first kNodeCache records
// of a larger one whose stacks still fit 16 entries (PARTIAL = true) — rt_scene_create numbers the nodes of the
// device copy breadth-first from the root, so the first records are the top levels of the BVHs, the ones every ray
// goes through. Same records, same arithmetic, same results. Measured (tools/scaling_scenes.py, 1200x800x160): 549
// nodes +21 %, 12 213 nodes (1740 of them in LDS) +8.5 %. A third instance — 700 records beside stacks of 30 entries
// for the deep BVHs of 131 K / 1 M / the 1.7 M-node mesh of C5 — measured -2 % / -4.5 % / +-0 and was dropped: the top
// levels of a big BVH are L1-resident anyway, and what the table saves on a small one is the L2 latency of the levels below.
constexpr int trace_waves_per_simd(int stack, bool stats, unsigned feat, int wg) {
    return wg == kBlock ? trace_blocks_per_cu(stack, stats, feat) : wg / 256;
}
// FEAT: which arms the scene can reach (kFeat* bits); the others are compiled out, which is
// worth 20-60 VGPRs — the difference between 3 and 4-5 resident waves per SIMD.
// SPHERES: every primitive of the scene is a sphere (FEAT 0 and no rects) — the scenes whose node boxes are tested in single precision
// (a rect lies in the faces of its box, where that test decides nothing: see kF32 below).
template <int STACK, bool STATS, unsigned FEAT, bool PROBE = false, int WG = kBlock, int CACHE = 0, bool PARTIAL = false, bool PRIMS = false, bool SPHERES = false>
__global__ void __launch_bounds__(WG, trace_waves_per_simd(STACK, STATS, FEAT, WG)) wf_trace(const SceneDev s, const WfPool pool,
                                                   const double t_min, const uint32_t node_quorum_u, const uint32_t parity, StatsDev *stats,
                                                   const uint32_t vote_weights) {
    // (Scene and pool by value: pointer members of kernel arguments are known to be global
    // memory, so node / ray fetches compile to global_load instead of flat_load, and none of
    // them is re-read from a descriptor in memory inside the traversal loop.)
    __shared__ uint32_t stack_lds[STACK * WG];
    // The world ray of every lane's current path, [component][lane] (12 KiB where the scene has movers, 48 bytes
    // otherwise): leaving a mover restarts from it (a ray_at_level of the enclosing frame) without going back to HBM.
    // (The deeper-stack variants have no LDS to spare at four workgroups per CU: they fetch it from the pool again.)
    constexpr bool kStash = (FEAT & kFeatMovers) != 0 && STACK <= kStackSmall && CACHE == 0;
    __shared__ double wray_lds[kStash ? 6 * WG : 6];
    // Node cache (CACHE > 0): boxes as three 16-byte words per node, child refs as one 8-byte word per node — or (kF32: the
    // all-in-LDS instance of sphere-only scenes) the single-precision records of t_slabs32, 44 bytes per node; the double-precision
    // boxes then stay in L2 for the few node steps the single-precision test cannot decide. Why sphere-only scenes: what the
    // float test cannot decide is a ray leaving a surface against a box that surface lies on the face of (the verdict hangs on
    // t_min = 0.001 against a distance of zero); a sphere touches its box in six points, a rect or a box lies in its faces:
    // one node step in 96 000 on the random spheres, one in 194 on the book-2 final scene, one in 27 in the Cornell box
    // (tools/f32_census.py) — measured, the random spheres' traversal kernel 6-7 % faster, the final scene's 1 % and the
    // Cornell box's 6 % slower, whether the undecided lanes fetch the double-precision box on the spot or hand the step to the
    // voted arm (profiles/r3q_ab_f32_slabs.log).
    // (... and the whole-table instance of the sphere-only scenes too large for the all-in-LDS one: 601 to 1 740 nodes.)
    constexpr bool kF32 = RT2022_F32_SLABS == 2 ? (CACHE > 0 && !PARTIAL && !(FEAT & kFeatMisc))
                                                : (RT2022_F32_SLABS == 1 && (PRIMS || (SPHERES && FEAT == 0 && CACHE > 0 && !PARTIAL)));
    // (kF32G) The same test in the plain kernels, for sphere scenes too large for that instance (SPHERES):
    // 32-byte single-precision records {min.x, max.x, min.y, max.y | min.z, max.z, left, push ref} — SceneDev::nodes32 — fetched as
    // two 16-byte loads from L2 / HBM: half the bytes of the double-precision record per node step. (These scenes take the plain
    // kernels even where the partial-table instance would apply: with the first 3 045 of these records in LDS that instance —
    // four waves per SIMD against the plain kernel's five — measured 10 % slower on the 1e4-sphere scene, node_cache_mode.)
    // ... and for the triangle meshes (FEAT with kFeatMisc, no boxes or media): a triangle touches its box in its corners, one node step
    // in 1 348 of wwscene is left undecided (its rings lie in the faces of theirs); C5's traversal kernel -3.2 % — once the ten VGPRs the
    // test needs were found: the RotateY stash is dropped in these instances (two divisions at a RotateY's exit instead; measured alone: no cost).
    constexpr bool kF32G = RT2022_F32_GLOBAL && RT2022_F32_SLABS >= 1 &&
                           (RT2022_F32_GLOBAL == 2 ? !(FEAT & kFeatVolumes) : ((FEAT == 0 && SPHERES) || ((FEAT & kFeatMisc) && !(FEAT & kFeatVolumes)))) &&
                           !PRIMS && !STATS && !PROBE && CACHE == 0 && !kF32;
    constexpr bool kStashInv = RT2022_STASH && !(kF32G && (FEAT & kFeatMovers));
    __shared__ f64x2 nc_box[CACHE > 0 && !kF32 && !kF32G ? 3 * CACHE : 1];
    __shared__ u32x2 nc_ref[CACHE > 0 && !kF32 && !kF32G ? CACHE : 1];
    __shared__ uint32_t nc32[kF32 ? kNode32Words * CACHE : 1];
    // ... and, in every variant (384 bytes), the first records of the two small tables the arms go to most: movers (32 B
    // each) and media (MediumDev, 64 B each) — two of each in the book-2 final scene.
#ifndef RT2022_SMALL_TABLES_EVERYWHERE
#define RT2022_SMALL_TABLES_EVERYWHERE 1
#endif
    constexpr bool kSmall = CACHE > 0 || RT2022_SMALL_TABLES_EVERYWHERE;
    constexpr uint32_t kLdsXforms = kSmall && (FEAT & kFeatMovers) ? 8u : 0u, kLdsMedia = kSmall && (FEAT & kFeatVolumes) ? 2u : 0u;
    __shared__ u32x4 xf_lds[kLdsXforms ? 2 * kLdsXforms : 1];
    __shared__ f64x2 md_lds[kLdsMedia ? 4 * kLdsMedia : 1];
    // PRIMS (sphere-only scenes small enough, C2): the sphere pools too — 32 B of centre and radius + 4 B of material
    // word per Sphere, the 80-byte record per MovingSphere; launched only when both pools fit whole.
    __shared__ f64x2 sp_lds[PRIMS ? 2 * kPrimSpheres : 1];
    __shared__ uint32_t spm_lds[PRIMS ? kPrimSpheres : 1];
    __shared__ f64x2 ms_lds[PRIMS ? 5 * kPrimMoving : 1];
    const PoolView pv{pool};
    const uint32_t tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    Counters<STATS> cnt;
    TStack<STACK, WG> st{stack_lds + tid};
    const uint32_t n_cached = CACHE > 0 ? (s.n_nodes < (uint32_t)CACHE ? s.n_nodes : (uint32_t)CACHE) : 0u;
    if (CACHE > 0) {
        for (uint32_t i = tid; i < n_cached; i += (uint32_t)WG) {
            const f64x2 *np = reinterpret_cast<const f64x2 *>(s.nodes + i);
            f64x2 b0 = np[0], b1 = np[1], b2 = np[2];
            const u32x4 rw = reinterpret_cast<const u32x4 *>(np)[3];          // {left, right, push ref, -}: see rt_scene_create
            const u32x2 rr = {rw.x, rw.z};
            if (kF32) {
                const float lo[3] = {(float)b0.x, (float)b0.y, (float)b1.x}, hi[3] = {(float)b1.y, (float)b2.x, (float)b2.y};
                uint32_t *rec = nc32 + kNode32Words * i;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    rec[3 * a] = __float_as_uint(lo[a]); rec[3 * a + 1] = __float_as_uint(hi[a]); rec[3 * a + 2] = __float_as_uint(lo[a]);
                }
                rec[9] = rr.x; rec[10] = rr.y;
            } else {
                nc_box[3 * i] = b0; nc_box[3 * i + 1] = b1; nc_box[3 * i + 2] = b2;
                nc_ref[i] = rr;
            }
        }
        if (PRIMS) {
            for (uint32_t i = tid; i < s.n_spheres && i < (uint32_t)kPrimSpheres; i += (uint32_t)WG) {
                const f64x2_a8 *qp = reinterpret_cast<const f64x2_a8 *>(s.spheres + i);
                sp_lds[2 * i] = qp[0]; sp_lds[2 * i + 1] = qp[1];
                spm_lds[i] = s.spheres[i].mat;
            }
            for (uint32_t i = tid; i < 5u * s.n_moving_spheres && i < 5u * (uint32_t)kPrimMoving; i += (uint32_t)WG)
                ms_lds[i] = reinterpret_cast<const f64x2 *>(s.moving_spheres)[i];
        }
    }
    if (kLdsXforms || kLdsMedia || CACHE > 0) {
        if (tid < 2 * kLdsXforms && tid < 2 * s.n_xforms) xf_lds[tid] = reinterpret_cast<const u32x4 *>(s.xforms)[tid];
        if (tid >= 64 && tid < 64 + 4 * kLdsMedia && tid < 64 + 4 * s.n_media) md_lds[tid - 64] = reinterpret_cast<const f64x2 *>(s.media_dev)[tid - 64];
        __syncthreads();
    }
    // A mover's record {kind, child | p[0] | p[1], p[2]} from wherever it lives.
    auto xform_words = [&](uint32_t idx, u32x4 &x0, f64x2 &x1) {
        if (kLdsXforms && idx < kLdsXforms) { x0 = xf_lds[2 * idx]; x1 = reinterpret_cast<const f64x2 *>(xf_lds)[2 * idx + 1]; }
        else { const u32x4 *xp = reinterpret_cast<const u32x4 *>(s.xforms + idx); x0 = xp[0]; x1 = reinterpret_cast<const f64x2 *>(xp)[1]; }
    };
    // ray_at_level of pt_common.hpp with the movers' records taken through xform_words.
    auto ray_at = [&](const Chain &ch, uint32_t level, XRay r) {
        for (uint32_t i = 0; i < level && i < RT_MAX_XFORM_DEPTH; i++) {
            const uint32_t ref = ch.at(i);
            u32x4 x0; f64x2 x1;
            xform_words(RT_REF_INDEX(ref), x0, x1);
            r = xform_ray_p(RT_REF_KIND(ref), rtm::u2d(((uint64_t)x0.w << 32) | x0.z), x1.x, x1.y, r);
        }
        return r;
    };
    double *const wray = wray_lds + (kStash ? tid : 0u);
    // (The node-table variant has no LDS left for the world rays and fetches them from the pool again. Keeping them in
    // twelve more registers instead — 128 in all, nothing spilled — measured the same: +0.3 %, A/B.)

    // Work of a pass = the ray lists of all segments (written by the preceding shade pass), cut into chunks of
    // kChunk entries and numbered slice-major: chunk id -> (slice = id / segments, segment = id % segments), so
    // that the counter hands out every segment's longest rays first. Each wave takes chunks from one global
    // counter as it runs dry — the waves, workgroups and CUs of the persistent grid therefore all finish within
    // one chunk of each other however unevenly they advance. (Bound statically to its segments, a workgroup's
    // speed depended on its CU and on its dispatch order within the CU — the arbiter serves the oldest wave
    // first — and a pass waited 10-25 % of its time for the slowest: rt_debug_pass_timing)
    const uint32_t n_seg = pool.n_blocks;
    const uint32_t total_ids = ((pool.max_list[parity] + kChunk - 1u) / kChunk) * n_seg;
    // This wave's chunk — entries [base + taken, base + n) of pool.list — and "the counter has run out": per-wave
    // state {base, n, taken, drained}, kept in LDS rather than in four more live registers. Written by the wave's
    // leader lane and read by whichever lanes publish next, as ONE volatile 16-byte access each way: volatile, so
    // every access is a real ds_read_b128 / ds_write_b128 in program order — one wave's LDS operations complete in
    // the order it issues them, and the compiler may not carry the words in registers from one round to the next.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ u32x4 chunk_lds[WG / 64];
    volatile u32x4 *const cs = &chunk_lds[tid >> 6];
    if (lane == 0) *cs = (u32x4){0u, 0u, 0u, 0u};
    const bool probe = PROBE && pool.dbg != nullptr;                  // (rt_debug_pass_timing: a build of its own, all arms)
    unsigned long long t_start = 0, t_dry = 0;
    bool dry_seen = false;
    if (probe) t_start = wall_clock64();

#ifndef RT2022_REFILL_TOUCH
#define RT2022_REFILL_TOUCH 0
#endif
#if RT2022_REFILL_TOUCH
    uint32_t touch_word = 0;
#endif
    TP_DECL;
#ifdef RT2022_F32_CENSUS
    unsigned f32_steps = 0, f32_undecided = 0, f32_wrong = 0;
#endif
    TLane L;
    L.flags = 0; L.op = OP_SHADE; L.top = REF_EMPTY; L.sp = 0; L.slot = 0; L.entry = 0; L.steps = 0;
    L.closest = rtm::F64_MAX; L.a_len = 0.0; L.tm = 0.0;
    L.t_lo = t_min; L.sub_closest = rtm::INF; L.med_t1 = 0.0; L.med_ref = 0;
    L.ctx.c0 = L.ctx.c1 = L.ctx.c2 = L.ctx.c3 = 0; L.ctx.n = 0;
    L.win_chain = L.ctx; L.win_leaf = REF_EMPTY; L.win_face = 0; L.win_mat = 0;
    L.stash_ix = 0.0; L.stash_iz = 0.0; L.stash_level = 0xFFFFFFFFu;
    const int node_quorum = (int)(node_quorum_u & 0xFFu);
#ifndef RT2022_SPHERE_REPS
#define RT2022_SPHERE_REPS 2
#endif
    constexpr int sphere_reps = RT2022_SPHERE_REPS;                   // (a span-2 leaf pair in one turn; three: -1.6 % on the headline, -3 % on C2 in one A/B call)
    constexpr int tail_factor = 2;
    const bool boxes_plain = (node_quorum_u >> 31) != 0;             // host: every node box finite with min <= max
    unsigned census_rounds[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, census_lanes[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // What the node fast path keeps in registers across its turns (r3). The library is built without MachineLICM (Makefile:
    // hoisted f64 literals were being spilled), so nothing hoists a loop's constants any more — and in THIS loop every
    // instruction counts: rebuilding the classify table, the empty-stack ref and the two LDS table addresses each turn is five
    // more instructions per node step (C2: -5 %). The empty asm makes each an opaque value: it cannot be rebuilt inside.
    unsigned long long ctab = kClassifyTable;
    uint32_t ref_empty = REF_EMPTY;
    asm volatile("" : "+s"(ctab), "+v"(ref_empty));                  // (a select takes one scalar operand, and that is its lane mask)
#ifndef RT2022_SIGNED_SLABS
#define RT2022_SIGNED_SLABS 1          // node table in LDS: the near / far box coordinate of each axis fetched by the sign of 1/d (no min / max per axis)
#endif
    constexpr bool kSlabs = RT2022_SIGNED_SLABS && CACHE > 0 && !PARTIAL && !(FEAT & kFeatMisc) && !kF32;      // (a partial table mixes both sources in one wave; the triangle kernels have no six registers to spare)
    typedef const __attribute__((address_space(3))) f64x2 *LdsBoxPtr;
    typedef const __attribute__((address_space(3))) u32x2 *LdsRefPtr;
    LdsBoxPtr ncb = (LdsBoxPtr)nc_box;
    LdsRefPtr ncr = (LdsRefPtr)nc_ref;
    if (CACHE > 0) asm volatile("" : "+v"(ncb), "+v"(ncr));
    uint32_t table32_at = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)nc32;
    if (kF32) asm volatile("" : "+v"(table32_at));
    const uint32_t table_at = kF32 ? table32_at : (uint32_t)(uintptr_t)ncb;              // (LDS byte address of node record 0's box)
    L.near_at[0] = L.near_at[1] = L.near_at[2] = L.far_at[0] = L.far_at[1] = L.far_at[2] = table_at;
    L.p32[0] = L.p32[1] = L.p32[2] = (f32x2){0.0f, 0.0f}; L.e_ray = __builtin_inff();
    uint32_t refs32_at = table32_at + 36u;                             // (kF32: the child refs of record 0)
    if (kF32) asm volatile("" : "+v"(refs32_at));

    for (;;) {
        // Fast path: keep stepping nodes while enough lanes want to — nn >= the quorum. Below the quorum the vote
        // decides, except where its outcome is known: node steps weigh 1 and everything else 2, so with
        // nn > 2 x (all other pending lanes) the vote would pick the node step anyway (the usual case at the tail of a
        // pass, when the list has run dry and a few long rays are left); staying here saves the vote.
        // Inside the loop a lane can only leave the node state (the others are parked), so the number of pending
        // lanes is fixed on entry and both conditions are ONE threshold on nn: nn >= quorum, or 3 nn > 2 pending.
        // The loop itself is a plain divergent loop over the node lanes — a lane that leaves the node state drops out
        // of it, and all that are left go together when their count falls below the threshold.
        {
            // a node step (OP_NODE is label 0) of a plain ray — and not one the single-precision test has handed on: one compare, one vote
            bool isn = (kF32 || kF32G) ? (L.op | ((L.flags ^ kPlain) & (kPlain | kNeed64))) == 0u : (L.op | (~L.flags & kPlain)) == 0u;
            int nn = __popcll(wballot(isn));
            const int pending = __popcll(wballot(L.op != OP_IDLE));
            const int tail_threshold = tail_factor * pending / (tail_factor + 1) + 1;
            const int threshold = node_quorum < tail_threshold ? node_quorum : tail_threshold;
            // (t_lo and t_hi do not change inside the loop: a lane can only leave it)
            double tlo_c = L.t_lo, thi_c = t_hi(L);
            asm volatile("" : "+v"(tlo_c), "+v"(thi_c));              // (in vector registers, once per entry)
            const bool entered = isn && nn >= threshold;
            // (kF32) the window's ends in single precision and the error bound of this entry: the ray's share plus what the
            // two conversions can be off by (an infinite end converts exactly)
            float tlo32 = 0.0f, thi32 = 0.0f;
            if ((kF32 || kF32G) && entered) {
                tlo32 = (float)tlo_c; thi32 = (float)thi_c;
                asm volatile("" : "+v"(tlo32), "+v"(thi32));
            }
            if (entered) do {
                if (STATS) { const unsigned long long am = wballot(true); if ((int)lane == __ffsll((long long)am) - 1) { census_rounds[8]++; census_lanes[8] += (unsigned)nn; } }
                {
                // BvhNode::hit, bvh/mod.rs:86-101 + AABB::hit, aabb.rs:15-32. The left child is taken
                // at once, the right one waits on the stack and is tested against the then-closest hit.
                //
                // For a `plain` ray (t_set_cur: every 1/d finite and non-zero, origin finite) against
                // finite boxes with min <= max, no t0 / t1 is NaN and the products are ordered by the
                // sign of 1/d, so the swap of aabb.rs:22-24 is min / max of the pair; the interval only
                // shrinks from axis to axis, so the per-axis `t_max <= t_min` exits equal one test at
                // the end. Any other ray takes the literal restatement in the voted arm below.
                //
                // Straight-line on purpose: the node's 64 bytes and the stack entry below the top are
                // requested together, before the arithmetic — no load waits for the outcome of the test.
                const uint32_t nidx = RT_REF_INDEX(L.top);
                const int below_sp = L.sp > 0 ? L.sp - 1 : 0;
                double bmin[3], bmax[3];
                uint32_t left, right, below;
                bool hit, undecided = false;
                if (kF32) {
                    // Single-precision slab test with a double-precision second opinion (r3). The node's box is held as floats
                    // (t_slabs32: one two-word LDS read per axis delivers the coordinates the ray meets first and last), the two
                    // slab distances of an axis are ONE packed multiply-add, max3 / min3 fold the axes: ten vector instructions
                    // where the double-precision test needs nineteen. Its verdict is taken only where it cannot differ from the
                    // double-precision one: |tmx - tmn| above the error bound of t_slabs32; a lane it leaves undecided (one node step
                    // in 200 on the book-2 final scene, one in 70 000 on the random spheres: tools/f32_census.py) hands the step to the
                    // voted node arm, which fetches the double-precision record. Same decisions, bit for bit.
                    uint32_t a0, a1, a2, ar;
                    asm("v_mad_u32_u24 %0, %1, 44, %2" : "=v"(a0) : "v"(L.top), "v"(L.near_at[0]));
                    asm("v_mad_u32_u24 %0, %1, 44, %2" : "=v"(a1) : "v"(L.top), "v"(L.near_at[1]));
                    asm("v_mad_u32_u24 %0, %1, 44, %2" : "=v"(a2) : "v"(L.top), "v"(L.near_at[2]));
                    asm("v_mad_u32_u24 %0, %1, 44, %2" : "=v"(ar) : "v"(L.top), "v"(refs32_at));
                    static_assert(kNode32Bytes == 44, "the multiply-adds above carry the record size");
                    const uint32_t below_at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(st.col + below_sp * WG);
                    f32x2 bx, by, bz, tx, ty, tz;
                    u32x2 cr;
                    asm volatile("ds_read2_b32 %0, %5 offset1:1\n\tds_read2_b32 %1, %6 offset1:1\n\tds_read2_b32 %2, %7 offset1:1\n\t"
                                 "ds_read2_b32 %3, %8 offset1:1\n\tds_read_b32 %4, %9\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(bx), "=&v"(by), "=&v"(bz), "=&v"(cr), "=&v"(below) : "v"(a0), "v"(a1), "v"(a2), "v"(ar), "v"(below_at) : "memory");
                    // {t first, t last} = {b first, b last} * (1/d) + (-o/d): low halves of both results take the pair's low word, the addend its high word
                    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tx) : "v"(bx), "v"(L.p32[0]));
                    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(ty) : "v"(by), "v"(L.p32[1]));
                    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tz) : "v"(bz), "v"(L.p32[2]));
                    float tmn32, tmx32;
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmn32) : "v"(tx.x), "v"(ty.x), "v"(tz.x));
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmx32) : "v"(tx.y), "v"(ty.y), "v"(tz.y));
                    asm("v_max_f32 %0, %1, %2" : "=v"(tmn32) : "v"(tmn32), "v"(tlo32));
                    asm("v_min_f32 %0, %1, %2" : "=v"(tmx32) : "v"(tmx32), "v"(thi32));
                    const float gap = tmx32 - tmn32;
                    hit = gap > 0.0f;
                    const float e_tot = __builtin_fmaf(__builtin_fabsf(tmx32) + __builtin_fabsf(tmn32), kF32RelBound, L.e_ray);
                    undecided = !(__builtin_fabsf(gap) > e_tot);                   // (a NaN anywhere lands here too)
#ifdef RT2022_F32_CENSUS
                    f32_steps++; if (undecided) f32_undecided++;
                    if (!undecided) {                                      // (the census build checks every verdict it takes against the double-precision test)
                        const f64x2 *np = reinterpret_cast<const f64x2 *>(s.nodes + nidx);
                        const f64x2 n0 = np[0], n1 = np[1], n2 = np[2];
                        const double lo3[3] = {n0.x, n0.y, n1.x}, hi3[3] = {n1.y, n2.x, n2.y};
                        double tmn = tlo_c, tmx = thi_c;
                        for (int i = 0; i < 3; i++) {
                            const double t0 = (lo3[i] - L.cur.o[i]) * L.inv[i], t1 = (hi3[i] - L.cur.o[i]) * L.inv[i];
                            tmn = __builtin_fmax(tmn, __builtin_fmin(t0, t1));
                            tmx = __builtin_fmin(tmx, __builtin_fmax(t0, t1));
                        }
                        if (hit != !(tmx <= tmn)) f32_wrong++;
                    }
#endif
                    left = cr.x; right = cr.y;
                } else if (kF32G) {
                    // The single-precision test on the 32-byte record (see kF32G above): {min, max} of an axis are one register pair,
                    // one packed multiply-add gives the axis' two slab distances, ordered afterwards (a min and a max per axis —
                    // no per-lane addresses here: one base address serves both loads).
                    const u32x4 *np = reinterpret_cast<const u32x4 *>(s.nodes32) + 2u * (uint64_t)nidx;
                    u32x4 q0 = np[0], q1 = np[1];
                    below = st.col[below_sp * WG];
                    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(below));              // (both halves and the stack entry asked for together)
                    const f32x2 bx = {__uint_as_float(q0.x), __uint_as_float(q0.y)}, by = {__uint_as_float(q0.z), __uint_as_float(q0.w)},
                                bz = {__uint_as_float(q1.x), __uint_as_float(q1.y)};
                    f32x2 tx, ty, tz;
                    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tx) : "v"(bx), "v"(L.p32[0]));
                    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(ty) : "v"(by), "v"(L.p32[1]));
                    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tz) : "v"(bz), "v"(L.p32[2]));
                    float nx, ny, nz, fx, fy, fz, tmn32, tmx32;
                    asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(tx.x), "v"(tx.y)); asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(tx.x), "v"(tx.y));
                    asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ty.x), "v"(ty.y)); asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ty.x), "v"(ty.y));
                    asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(tz.x), "v"(tz.y)); asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(tz.x), "v"(tz.y));
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmn32) : "v"(nx), "v"(ny), "v"(nz));
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmx32) : "v"(fx), "v"(fy), "v"(fz));
                    asm("v_max_f32 %0, %1, %2" : "=v"(tmn32) : "v"(tmn32), "v"(tlo32));
                    asm("v_min_f32 %0, %1, %2" : "=v"(tmx32) : "v"(tmx32), "v"(thi32));
                    const float gap = tmx32 - tmn32;
                    hit = gap > 0.0f;
                    const float e_tot = __builtin_fmaf(__builtin_fabsf(tmx32) + __builtin_fabsf(tmn32), kF32RelBound, L.e_ray);
                    undecided = !(__builtin_fabsf(gap) > e_tot);                   // (a NaN anywhere lands here too)
#ifdef RT2022_F32_CENSUS
                    f32_steps++; if (undecided) f32_undecided++;
                    if (!undecided) {
                        const f64x2 *np = reinterpret_cast<const f64x2 *>(s.nodes + nidx);
                        const f64x2 n0 = np[0], n1 = np[1], n2 = np[2];
                        const double lo3[3] = {n0.x, n0.y, n1.x}, hi3[3] = {n1.y, n2.x, n2.y};
                        double tmn = tlo_c, tmx = thi_c;
                        for (int i = 0; i < 3; i++) {
                            const double t0 = (lo3[i] - L.cur.o[i]) * L.inv[i], t1 = (hi3[i] - L.cur.o[i]) * L.inv[i];
                            tmn = __builtin_fmax(tmn, __builtin_fmin(t0, t1));
                            tmx = __builtin_fmin(tmx, __builtin_fmax(t0, t1));
                        }
                        if (hit != !(tmx <= tmn)) f32_wrong++;
                    }
#endif
                    left = q1.z; right = q1.w;
                } else {
                if (CACHE > 0 && (!PARTIAL || nidx < n_cached)) {     // (PARTIAL: the table holds the first n_cached nodes — the top of the BVHs, rt_scene_create numbers them breadth-first)
                    // (LDS addresses are 32 bits and a table index is far below 2^24: one v_mad_u32_u24 instead of a 64-bit multiply-add)
                    // (the 24-bit multiply-add takes the low 24 bits of the ref: its index, un-masked)
                    const uint32_t box_at = (uint32_t)(uintptr_t)ncb + __umul24(L.top, 48u);
                    uint32_t ref_at;
                    asm("v_mad_u32_u24 %0, %1, 8, %2" : "=v"(ref_at) : "v"(L.top), "v"((uint32_t)(uintptr_t)ncr));     // (one instruction; left alone the compiler masks, shifts and adds)
                    const uint32_t below_at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(st.col + below_sp * WG);
                    // The five LDS reads of a node step, issued back to back and waited for ONCE — written out, because the compiler's
                    // own placement of the waits split them (seen in the ISA: the child refs were waited for before the box was even
                    // asked for: two LDS round trips per node step instead of one).
                    u32x2 cr;
                    if (kSlabs) {
                        // bmin[] / bmax[] here are the coordinates the ray meets FIRST / LAST on each axis (bmin or bmax by the sign of
                        // 1/d: t_slabs): for a plain ray min(t0, t1) is the product with the first, max(t0, t1) with the last.
                        const uint32_t off = __umul24(L.top, 48u);
                        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                                     "ds_read_b64 %4, %12\n\tds_read_b64 %5, %13\n\tds_read_b64 %6, %14\n\tds_read_b32 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(bmin[0]), "=&v"(bmin[1]), "=&v"(bmin[2]), "=&v"(bmax[0]), "=&v"(bmax[1]), "=&v"(bmax[2]), "=&v"(cr), "=&v"(below)
                                     : "v"(L.near_at[0] + off), "v"(L.near_at[1] + off), "v"(L.near_at[2] + off), "v"(L.far_at[0] + off), "v"(L.far_at[1] + off),
                                       "v"(L.far_at[2] + off), "v"(ref_at), "v"(below_at) : "memory");
                    } else {
                        f64x2 c0, c1, c2;
                        asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\t"
                                     "ds_read_b64 %3, %6\n\tds_read_b32 %4, %7\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(cr), "=&v"(below) : "v"(box_at), "v"(ref_at), "v"(below_at) : "memory");
                        bmin[0] = c0.x; bmin[1] = c0.y; bmin[2] = c1.x; bmax[0] = c1.y; bmax[1] = c2.x; bmax[2] = c2.y;
                    }
                    left = cr.x; right = cr.y;
                } else {
                    const uint4 *np = reinterpret_cast<const uint4 *>(s.nodes + nidx);
                    uint4 q0 = np[0], q1 = np[1], q2 = np[2];
                    u32x4 q3 = reinterpret_cast<const u32x4 *>(np)[3];
                    below = st.col[below_sp * WG];
                    asm volatile("" : "+v"(q3), "+v"(below));         // (q3 as ONE 16-byte load — left and the push ref are not neighbours in it — and the stack read beside the fetches)
                    bmin[0] = rtm::u2d(((uint64_t)q0.y << 32) | q0.x); bmin[1] = rtm::u2d(((uint64_t)q0.w << 32) | q0.z); bmin[2] = rtm::u2d(((uint64_t)q1.y << 32) | q1.x);
                    bmax[0] = rtm::u2d(((uint64_t)q1.w << 32) | q1.z); bmax[1] = rtm::u2d(((uint64_t)q2.y << 32) | q2.x); bmax[2] = rtm::u2d(((uint64_t)q2.w << 32) | q2.z);
                    left = q3.x; right = q3.z;                    // (the push ref: `right`, or "nothing" for a span-1 twin — rt_scene_create)
                }
                double tmn, tmx;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    double t0 = (bmin[i] - L.cur.o[i]) * L.inv[i];
                    double t1 = (bmax[i] - L.cur.o[i]) * L.inv[i];
                    constexpr bool sorted = kSlabs;                               // (t0 <= t1 already: the coordinates came in that order)
                    const double lo = sorted ? t0 : __builtin_fmin(t0, t1), hi = sorted ? t1 : __builtin_fmax(t0, t1);
                    // fmax / fmin of a value that is not an arithmetic result of the same block first "canonicalises" it (a
                    // v_max_f64 x, x) — per node step, for the window's two ends, which never change in here. Written as the
                    // instruction fmax / fmin compile to; no operand is a NaN on this path (see above), so it is the same value.
                    asm("v_max_f64 %0, %1, %2" : "=v"(tmn) : "v"(i == 0 ? tlo_c : tmn), "v"(lo));
                    asm("v_min_f64 %0, %1, %2" : "=v"(tmx) : "v"(i == 0 ? thi_c : tmx), "v"(hi));
                }
                hit = !(tmx <= tmn);
                }
                // A span-1 node holds the same object twice (bvh/mod.rs:44-47). Testing a plain
                // primitive a second time against t_max = its own t finds the same hit again, so
                // only the count of tests is kept; anything that can draw from the RNG or carry
                // movers (media, movers, nodes, lists) is really visited twice. rt_scene_create has
                // worked that out per node: `right` here is the node's PUSH REF — its right child, or
                // REF_EMPTY where the twin needs no second visit (r3: one compare instead of five).
                if ((kF32 || kF32G) && undecided) {
                    // Mostly a ray leaving a surface against a box that surface lies on the face of: the verdict hangs on t_min = 0.001
                    // against a distance of zero, which floats of the scene's size cannot tell apart. The lane keeps its node and leaves
                    // the loop; the voted node arm takes the step on the double-precision record (no lane here waits for that fetch).
                    L.flags |= kNeed64;
                    isn = false;
                } else {
                cnt.node();
                L.steps++;
                const bool twin = right == ref_empty;
                const bool push = hit && !twin && L.sp < STACK;
                if (push) st.col[L.sp * WG] = right;
                if (STATS && hit && twin) cnt.prim(RT_REF_KIND(left));
                const uint32_t next = hit ? left : (L.sp > 0 ? below : ref_empty);
                L.sp = hit ? L.sp + (push ? 1 : 0) : below_sp;
                L.top = next;
                // (a node ref is kind 0 without the FlipFace bit — rt_scene_create refuses a flipped node — so "another node step"
                // is one compare; the label of whatever else came up is looked up once, when the lane leaves the loop)
                isn = next < (1u << RT_REF_KIND_SHIFT);
                }
                }
                nn = __popcll(wballot(isn));
            } while (isn && nn >= threshold);
            if (entered) L.op = classify(L.top, ctab);                // (media met in there start in their own arm)
        }
        TP_MARK(0);
        // Vote: the label most lanes are waiting on (ties -> lowest id).
        int best = -1, best_n = 0;
#pragma unroll
        for (int o = 0; o < (int)OP_COUNT; o++) {
            if (!(FEAT & kFeatMisc) && o == (int)OP_MISC) continue;
            if (!(FEAT & kFeatMovers) && o == (int)OP_CTX) continue;
            if (!(FEAT & kFeatVolumes) && (o == (int)OP_BOX || o == (int)OP_MEDIUM)) continue;
            int n = __popcll(wballot(L.op == (uint32_t)o));
            // Weights, four bits per label (rt_debug_set_tuning; default kWfVoteWeights): a node step outside the fast path and
            // the refill yield to everything else — node 2, refill 2, the rest 4 (refill at 4: -3 % on the headline, A/B).
            int score = n * (int)((vote_weights >> (4 * o)) & 0xFu);
            if (score > best_n) { best_n = score; best = o; }
        }
        if (best < 0) break;                                          // every lane idle
        TP_MARK(1);
        if (STATS) {
            unsigned served = (unsigned)__popcll(wballot(L.op == (uint32_t)best));
            if (lane == 0) { census_rounds[best]++; census_lanes[best] += served; }
        }
        if (L.op != (uint32_t)best) {
            // parked: this lane's operation did not win the vote
        } else if (best == OP_NODE) {
            // (a node step below the fast path's quorum — a handful of lanes: those whose next entry is a node again take it in the
            // same turn, like the leaf arms do with their pairs)
#ifndef RT2022_NODE_REPS
#define RT2022_NODE_REPS 1
#endif
#pragma unroll 1
            for (int rep = 0; rep < RT2022_NODE_REPS && L.op == OP_NODE; rep++) {
            cnt.node();
            L.steps++;
            const uint32_t nidx = RT_REF_INDEX(L.top);
            f64x2 n0, n1, n2;
            u32x4 n3;
            bool decided = false, miss = false;
            if (kF32) {
                // (the single-precision test of the fast path for the plain rays that come through here — a node step below the
                // quorum; everything else, and what it leaves undecided, takes the literal test on the double-precision record)
                const uint32_t *rec = nc32 + kNode32Words * nidx;
                n3 = (u32x4){rec[9], 0u, rec[10], 0u};
                if ((L.flags & (kPlain | kNeed64)) == kPlain) {
                    const float tlo32 = (float)L.t_lo, thi32 = (float)t_hi(L);
                    float tmn32 = tlo32, tmx32 = thi32;
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const bool neg = L.inv[i] < 0.0;
                        const float lo = __uint_as_float(rec[3 * i]), hi = __uint_as_float(rec[3 * i + 1]);
                        const float t0 = __builtin_fmaf(neg ? hi : lo, L.p32[i].x, L.p32[i].y), t1 = __builtin_fmaf(neg ? lo : hi, L.p32[i].x, L.p32[i].y);
                        tmn32 = __builtin_fmaxf(tmn32, t0); tmx32 = __builtin_fminf(tmx32, t1);
                    }
                    const float gap = tmx32 - tmn32;
                    const float e_tot = __builtin_fmaf(__builtin_fabsf(tmx32) + __builtin_fabsf(tmn32), kF32RelBound, L.e_ray);
                    decided = __builtin_fabsf(gap) > e_tot;
                    miss = decided && !(gap > 0.0f);
                }
            }
            if (kF32 && decided) {
                n0 = n1 = n2 = (f64x2){0.0, 0.0};
            } else if (CACHE > 0 && !kF32 && (!PARTIAL || nidx < n_cached)) {     // (PARTIAL: the table holds the first n_cached nodes — the top of the BVHs, rt_scene_create numbers them breadth-first)
                n0 = nc_box[3 * nidx]; n1 = nc_box[3 * nidx + 1]; n2 = nc_box[3 * nidx + 2];
                const u32x2 cr = nc_ref[nidx];
                n3 = (u32x4){cr.x, 0u, cr.y, 0u};
            } else {
                const f64x2 *np = reinterpret_cast<const f64x2 *>(s.nodes + nidx);
                n0 = np[0]; n1 = np[1]; n2 = np[2];
                n3 = reinterpret_cast<const u32x4 *>(np)[3];
            }
            t_pin(n0); t_pin(n1); t_pin(n2); t_pin(n3);
            const double bmin[3] = {n0.x, n0.y, n1.x}, bmax[3] = {n1.y, n2.x, n2.y};
            double tmn = L.t_lo, tmx = t_hi(L);
            if (!(kF32 && decided)) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                double inv_d = L.inv[i];
                double t0 = (bmin[i] - L.cur.o[i]) * inv_d;
                double t1 = (bmax[i] - L.cur.o[i]) * inv_d;
                if (inv_d < 0.0) { double tmp = t0; t0 = t1; t1 = tmp; }
                tmn = t0 > tmn ? t0 : tmn;
                tmx = t1 < tmx ? t1 : tmx;
                miss = miss || (tmx <= tmn);
            }
            }
            if (kF32 || kF32G) L.flags &= ~kNeed64;
            if (!miss) {
                const uint32_t left = n3.x, push_ref = n3.z;           // (push ref: see the fast path)
                if (push_ref == REF_EMPTY) cnt.prim(RT_REF_KIND(left));
                else st.push(L, push_ref);
                L.top = left;
                L.op = classify(left);
            } else {
                L.top = st.pop(L);
                L.op = classify(L.top);
            }
            }
        } else if (best == OP_SPHERE) {                               // Sphere / MovingSphere::hit
            // BVH leaves come in pairs (span-2 nodes): a lane whose next entry is a sphere again takes
            // it here and now rather than waiting for another round.
#pragma unroll 1
            for (int rep = 0; rep < sphere_reps && L.op == OP_SPHERE; rep++) {
                uint32_t kind = RT_REF_KIND(L.top), idx = RT_REF_INDEX(L.top);
                cnt.prim(kind);
                Vec3 center;
                double radius;
                uint32_t mat_word;
                if (kind == RT_KIND_SPHERE) {                         // rt_sphere, 40 B: center, radius, mat
                    f64x2 q0, q1;
                    if (PRIMS) { q0 = sp_lds[2 * idx]; q1 = sp_lds[2 * idx + 1]; mat_word = spm_lds[idx]; }
                    else {
                        const f64x2_a8 *qp = reinterpret_cast<const f64x2_a8 *>(s.spheres + idx);
                        q0 = qp[0]; q1 = qp[1];
                        mat_word = s.spheres[idx].mat;
                    }
                    t_pin(q0); t_pin(q1); t_pin(mat_word);
#ifdef RT2022_WHATIF_SPHERE_FETCH
                    // (diagnostic build: the record fetched a second time, behind the first — one more memory round trip per turn
                    // of this arm, same values: how much of the kernel's time is this arm's fetch latency? profiles/r3e_whatif_arms.log)
                    if (!PRIMS) {
                        uint32_t z;
                        asm volatile("v_and_b32 %0, 0, %1" : "=v"(z) : "v"((uint32_t)rtm::d2u(q0.x)));     // 0 — once q0 has arrived
                        const f64x2_a8 *qp2 = reinterpret_cast<const f64x2_a8 *>(s.spheres + (idx + z));
                        q0 = qp2[0]; q1 = qp2[1];
                        t_pin(q0); t_pin(q1);
                    }
#endif
                    center = Vec3(q0.x, q0.y, q1.x); radius = q1.y;
                } else {                                              // rt_moving_sphere, 80 B: center0, center1, time0, time1, radius, mat
                    const f64x2 *qp = PRIMS ? ms_lds + 5 * idx : reinterpret_cast<const f64x2 *>(s.moving_spheres + idx);
                    f64x2 q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3], q4 = qp[4];
                    t_pin(q0); t_pin(q1); t_pin(q2); t_pin(q3); t_pin(q4);
                    const Vec3 c0(q0.x, q0.y, q1.x), c1(q1.y, q2.x, q2.y);
                    center = c0 + (c1 - c0) * ((L.tm - q3.x) / (q3.y - q3.x));   // MovingSphere::center, sphere.rs:124-127
                    radius = q4.x;
                    mat_word = (uint32_t)rtm::d2u(q4.y);
                }
                double t;
                bool h = sphere_t(center, radius, L.cur, L.a_len, L.t_lo, t_hi(L), t);
#ifdef RT2022_WHATIF_SPHERE
                // (diagnostic build: the arm's arithmetic N times over, same result — how much of the kernel's time is this arm's
                // arithmetic? profiles/r3d_whatif_arms.log)
                for (int k = 1; k < RT2022_WHATIF_SPHERE; k++) {
                    Vec3 c2 = center; double r2 = radius, t2;
                    asm volatile("" : "+v"(c2.x), "+v"(c2.y), "+v"(c2.z), "+v"(r2));
                    const bool h2 = sphere_t(c2, r2, L.cur, L.a_len, L.t_lo, t_hi(L), t2);
                    h = h && h2; t = h ? t2 : t;
                }
#endif
                if (h) t_accept(L, t, 0, mat_word);
                T_NEXT();
            }
        } else if (best == OP_RECT) {
            cnt.prim(RT_KIND_RECT);
            const f64x2 *qp = reinterpret_cast<const f64x2 *>(s.rects + RT_REF_INDEX(L.top));      // rt_rect, 48 B: a0 a1 | b0 b1 | k, axis+mat
            f64x2 q0 = qp[0], q1 = qp[1], q2 = qp[2];
            t_pin(q0); t_pin(q1); t_pin(q2);
            const uint64_t am = rtm::d2u(q2.y);
            double t;
            if (rect_t((uint32_t)am, q0.x, q0.y, q1.x, q1.y, q2.x, L.cur, L.t_lo, t_hi(L), t)) t_accept(L, t, 0, (uint32_t)(am >> 32));
            T_NEXT();
        } else if ((FEAT & kFeatVolumes) && best == OP_BOX) {
            // (like the spheres: the leaves of a box BVH come in pairs, a lane whose next entry is a box again takes it in the same turn)
#ifndef RT2022_BOX_REPS
#define RT2022_BOX_REPS 2
#endif
#pragma unroll 1
            for (int rep = 0; rep < RT2022_BOX_REPS && L.op == OP_BOX; rep++) {
            cnt.prim(RT_KIND_BOX);
            const uint32_t bidx = RT_REF_INDEX(L.top);
            const f64x2_a8 *bp = reinterpret_cast<const f64x2_a8 *>(s.boxes + bidx);                // rt_box, 56 B: p0, p1, mat
            f64x2 b0 = bp[0], b1 = bp[1], b2 = bp[2];
            uint32_t mat_word = s.boxes[bidx].mat;
            t_pin(b0); t_pin(b1); t_pin(b2); t_pin(mat_word);
#ifdef RT2022_WHATIF_BOX_FETCH
            {
                uint32_t z;
                asm volatile("v_and_b32 %0, 0, %1" : "=v"(z) : "v"((uint32_t)rtm::d2u(b0.x)));
                const f64x2_a8 *bp2 = reinterpret_cast<const f64x2_a8 *>(s.boxes + (bidx + z));
                b0 = bp2[0]; b1 = bp2[1]; b2 = bp2[2];
                t_pin(b0); t_pin(b1); t_pin(b2);
            }
#endif
            double t;
            uint32_t face = 0;
            bool h = t_box(b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, L.cur, L.t_lo, t_hi(L), t, face);
#ifdef RT2022_WHATIF_BOX
            for (int k = 1; k < RT2022_WHATIF_BOX; k++) {
                f64x2 e0 = b0, e1 = b1, e2 = b2; double t2; uint32_t f2 = 0;
                asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2));
                const bool h2 = t_box(e0.x, e0.y, e1.x, e1.y, e2.x, e2.y, L.cur, L.t_lo, t_hi(L), t2, f2);
                h = h && h2; t = h ? t2 : t; face = h ? f2 : face;
            }
#endif
            if (h) t_accept(L, t, face, mat_word);
            T_NEXT();
            }
        } else if ((FEAT & kFeatVolumes) && best == OP_MEDIUM) {      // ConstantMedium::hit, constantmedium.rs:49-83
            // (the medium's record — boundary sphere inline — in one fetch: MediumDev, pt_device.h)
            f64x2 m0{0.0, 0.0}, m1{0.0, 0.0}, m2{0.0, 0.0};
            u32x4 m3{0u, 0u, 0u, 0u};
            const bool is_leaf = RT_REF_KIND(L.top) == RT_KIND_MEDIUM;
            if (is_leaf) {
                const uint32_t midx = RT_REF_INDEX(L.top);
                if (kLdsMedia && midx < kLdsMedia) {
                    m0 = md_lds[4 * midx]; m1 = md_lds[4 * midx + 1]; m2 = md_lds[4 * midx + 2];
                    m3 = reinterpret_cast<const u32x4 *>(md_lds)[4 * midx + 3];
                } else {
                    const f64x2 *mp = reinterpret_cast<const f64x2 *>(s.media_dev + midx);
                    m0 = mp[0]; m1 = mp[1]; m2 = mp[2];
                    m3 = reinterpret_cast<const u32x4 *>(mp)[3];
                }
            }
            t_pin(m0); t_pin(m1); t_pin(m2); t_pin(m3);
            if (is_leaf && m3.x != 0u) {
                // ConstantMedium::hit with a Sphere boundary, constantmedium.rs:49-83 in one go: the two
                // boundary queries are Sphere::hit (sphere.rs:39-58) on the same sphere with different t_min.
                struct { double neg_inv_density; } m{m2.x};
                struct { double radius; } q{m1.y};
                const Vec3 center(m0.x, m0.y, m1.x);
                const uint32_t mat_word = (uint32_t)(rtm::d2u(m2.y) >> 32);
                cnt.prim(RT_KIND_MEDIUM);
                cnt.prim(RT_KIND_SPHERE);
                double t1 = 0.0, t2 = 0.0;
                bool first;
                const bool both = sphere_t_twice(center, q.radius, L.cur, L.a_len, t1, t2, first);
                if (first) cnt.prim(RT_KIND_SPHERE);
                if (both) {
                    t1 = rtm::fmax_(t1, t_min);
                    t2 = rtm::fmin_(t2, L.closest);
                    if (!(t1 >= t2)) {
                        t1 = rtm::fmax_(t1, 0.0);
                        double ray_length = L.cur.d.length();
                        double distance_inside_boundary = (t2 - t1) * ray_length;
                        double rnd = L.rng.gen_f64();
                        double hit_distance = m.neg_inv_density * (rtm::log_(rnd) / rtm::log_(rtm::E_));
                        if (!(hit_distance > distance_inside_boundary)) t_accept(L, t1 + hit_distance / ray_length, 0, mat_word);   // (L.top is the medium)
                    }
                }
                T_NEXT();
            } else if (L.top != REF_MED2) {
                T_SETTLE();                                           // a medium leaf or a finished first query: same steps as inline
            } else if (L.top == REF_MED2) {
                uint32_t mref = L.med_ref;
                bool both = (L.flags & kSubFound) != 0;
                double t2 = L.sub_closest;
                L.med_ref = 0; L.t_lo = t_min;                        // back in the main query
                if (both) {
                    const MediumDev &m = s.media_dev[RT_REF_INDEX(mref)];
                    double t1 = rtm::fmax_(L.med_t1, t_min);
                    t2 = rtm::fmin_(t2, L.closest);
                    if (!(t1 >= t2)) {
                        t1 = rtm::fmax_(t1, 0.0);
                        double ray_length = L.cur.d.length();
                        double distance_inside_boundary = (t2 - t1) * ray_length;
                        double rnd = L.rng.gen_f64();
                        double hit_distance = m.neg_inv_density * (rtm::log_(rnd) / rtm::log_(rtm::E_));
                        if (!(hit_distance > distance_inside_boundary)) {
                            L.top = mref;                             // the medium itself is the winning leaf
                            t_accept(L, t1 + hit_distance / ray_length, 0, m.mat);
                        }
                    }
                }
                T_NEXT();
            }
        } else if ((FEAT & kFeatMisc) && best == OP_MISC) {                                 // Triangle, Ring
#ifndef RT2022_MISC_REPS
#define RT2022_MISC_REPS 2
#endif
#pragma unroll 1
            for (int rep = 0; rep < RT2022_MISC_REPS && L.op == OP_MISC; rep++) {
            uint32_t kind = RT_REF_KIND(L.top), idx = RT_REF_INDEX(L.top);
            cnt.prim(kind);
            double t;
            bool h;
            uint32_t mat_word;
            if (kind == RT_KIND_TRIANGLE) {                           // rt_triangle, 80 B: a, b, c, mat
                const f64x2 *qp = reinterpret_cast<const f64x2 *>(s.triangles + idx);
                f64x2 q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3], q4 = qp[4];
                t_pin(q0); t_pin(q1); t_pin(q2); t_pin(q3); t_pin(q4);
#ifdef RT2022_WHATIF_TRI_FETCH
                {   // (diagnostic build: the record fetched a second time behind the first — what is one memory round trip of this arm worth?)
                    uint32_t z;
                    asm volatile("v_and_b32 %0, 0, %1" : "=v"(z) : "v"((uint32_t)rtm::d2u(q0.x)));
                    const f64x2 *qp2 = reinterpret_cast<const f64x2 *>(s.triangles + (idx + z));
                    q0 = qp2[0]; q1 = qp2[1]; q2 = qp2[2]; q3 = qp2[3]; q4 = qp2[4];
                    t_pin(q0); t_pin(q1); t_pin(q2); t_pin(q3); t_pin(q4);
                }
#endif
                rt_triangle tr;
                tr.a[0] = q0.x; tr.a[1] = q0.y; tr.a[2] = q1.x; tr.b[0] = q1.y; tr.b[1] = q2.x; tr.b[2] = q2.y;
                tr.c[0] = q3.x; tr.c[1] = q3.y; tr.c[2] = q4.x;
                mat_word = (uint32_t)rtm::d2u(q4.y);
                h = triangle_t(tr, L.cur, L.t_lo, t_hi(L), t);
            } else {
                mat_word = s.rings[idx].mat;
                h = ring_t(s.rings[idx], L.cur, L.t_lo, t_hi(L), t);
            }
            if (h) t_accept(L, t, 0, mat_word);
            T_NEXT();
            }
        } else if ((FEAT & kFeatMovers) && best == OP_CTX) {                                  // movers in / out, HittableList expansion
            // 1/d of the ray changes only where d does: RotateY (x and z). Translate and Zoom leave the direction alone
            // (hittable/mod.rs:165-167,321-323), so entering or leaving them keeps inv and a_len — the same values the
            // three divisions would give again.
#ifndef RT2022_CTX_CHAINS
#define RT2022_CTX_CHAINS 1            // movers nested directly in one another (Translate(RotateY(Zoom(..))), scene.rs:320-322,403-412) are entered, and left, in ONE turn
#endif
            if (L.top == REF_POPCTX) {
                // Leaving a mover. When the next stack entry is the exit of the enclosing mover too — the movers were nested directly,
                // nothing else waits in the frames between — all of them are left in this turn: only the outermost frame's ray is ever
                // used again (r3b: three turns, three world-ray fetches and three re-derivations became one for the meshes of wwscene).
                bool rotated = false;
                uint32_t levels = 0;
                do {
                    L.ctx.n--;
                    rotated = rotated || RT_REF_KIND(L.ctx.at(L.ctx.n)) == RT_KIND_ROTATE_Y;           // (a mover being left)
                    L.top = st.pop(L);
                } while (RT2022_CTX_CHAINS && L.top == REF_POPCTX && L.ctx.n > 0u && ++levels < RT_MAX_XFORM_DEPTH);
                // the world ray from this lane's LDS column (written at refill), then back down to the enclosing frame
                XRay world;
                if (kStash) {
                    world = XRay{Vec3(wray[0 * WG], wray[1 * WG], wray[2 * WG]), Vec3(wray[3 * WG], wray[4 * WG], wray[5 * WG])};
                } else {
                    Ray wr = pv.load_ray(L.slot);
                    world = XRay{wr.orig, wr.dir};
                }
                L.cur = ray_at(L.ctx, L.ctx.n, world);
                if (rotated) {
                    // 1/d.x, 1/d.z of the frame arrived in: the stash holds them for the frame its RotateY was entered from — this one, or
                    // one whose direction is this one's (only a RotateY changes d); otherwise the two divisions again (same values).
                    bool stash_ok = kStashInv && L.stash_level != 0xFFFFFFFFu && L.stash_level >= L.ctx.n;
                    for (uint32_t j = L.ctx.n; stash_ok && j < L.stash_level && j < RT_MAX_XFORM_DEPTH; j++)
                        stash_ok = RT_REF_KIND(L.ctx.at(j)) != RT_KIND_ROTATE_Y;
                    if (stash_ok) { L.inv.x = L.stash_ix; L.inv.z = L.stash_iz; }
                    else { L.inv.x = 1.0 / L.cur.d.x; L.inv.z = 1.0 / L.cur.d.z; }
                    L.stash_level = 0xFFFFFFFFu;
                    L.a_len = L.cur.d.length_sqr();
                }
                t_flags(L, boxes_plain);
                if (kSlabs) t_slabs(L, table_at); if (kF32) t_slabs32(L, table_at); if (kF32G) t_slabs32g(L);
                T_SETTLE();
            } else {
                uint32_t kind = RT_REF_KIND(L.top), idx = RT_REF_INDEX(L.top);
                if (kind == RT_KIND_LIST) {
                    cnt.prim(kind);
                    const rt_list &l = s.lists[idx];
                    for (uint32_t i = l.count; i > 0; i--) st.push(L, s.list_items[l.first + i - 1]);
                    T_NEXT();
                } else if (L.ctx.n < RT_MAX_XFORM_DEPTH) {
                    // Entering a mover — and, in the same turn, the movers its child is wrapped in directly.
#pragma unroll 1
                    for (uint32_t rep = 0; rep < RT_MAX_XFORM_DEPTH; rep++) {
                        cnt.prim(kind);
                        u32x4 x0;                                     // rt_xform, 32 B: kind, child, p[3]
                        f64x2 x1;
                        xform_words(idx, x0, x1);
                        t_pin(x0); t_pin(x1);
                        const double p0 = rtm::u2d(((uint64_t)x0.w << 32) | x0.z), p1 = x1.x, p2 = x1.y;
                        if (kind == RT_KIND_TRANSLATE) {              // Translate::hit, mod.rs:165-167
                            L.cur.o = L.cur.o - Vec3(p0, p1, p2);
                        } else if (kind == RT_KIND_ROTATE_Y) {        // RotateY::hit, mod.rs:235-247 (p0 = sin, p1 = cos)
                            const double ox = p1 * L.cur.o.x - p0 * L.cur.o.z, oz = p0 * L.cur.o.x + p1 * L.cur.o.z;
                            const double dx = p1 * L.cur.d.x - p0 * L.cur.d.z, dz = p0 * L.cur.d.x + p1 * L.cur.d.z;
                            L.cur.o.x = ox; L.cur.o.z = oz; L.cur.d.x = dx; L.cur.d.z = dz;
                            if (kStashInv) { L.stash_ix = L.inv.x; L.stash_iz = L.inv.z; L.stash_level = L.ctx.n; }
                            L.inv.x = 1.0 / dx; L.inv.z = 1.0 / dz;
                            L.a_len = L.cur.d.length_sqr();
                        } else {                                      // Zoom::hit, mod.rs:321-323: the origin only
                            L.cur.o = L.cur.o / p0;
                        }
                        L.ctx.push(L.top);
                        st.push(L, REF_POPCTX);
                        L.top = x0.y;
                        kind = RT_REF_KIND(L.top); idx = RT_REF_INDEX(L.top);
                        if (!(RT2022_CTX_CHAINS && kind >= RT_KIND_TRANSLATE && kind <= RT_KIND_ZOOM && L.ctx.n < RT_MAX_XFORM_DEPTH)) break;
                    }
                    t_flags(L, boxes_plain);
                    if (kSlabs) t_slabs(L, table_at); if (kF32) t_slabs32(L, table_at); if (kF32G) t_slabs32g(L);
                    T_SETTLE();
                } else {
                    cnt.prim(kind);
                    T_NEXT();
                }
            }
        } else if (best == OP_SHADE) {
            // OP_SHADE here = "this lane's traversal is finished (or it has no ray yet)":
            // publish the winner, then pull the next ray from the block's list.
            if (L.flags & kHasRay) {
                uint32_t slot = L.slot;
                bool found = L.win_leaf != REF_EMPTY;
                uint32_t kind = SK_MISS;
                const uint32_t steps16 = (L.steps > 0xFFFFu ? 0xFFFFu : L.steps) << 16;
                if (found) {
                    pv.store_hit(slot, L.closest, L.win_leaf, L.win_face | (L.win_chain.n << 4) | steps16, L.win_chain, L.win_mat);
                    kind = L.win_mat >> kMatKindShift;                // (the word came with the winning primitive's record)
                }
                pool.kind[L.entry] = (uint8_t)kind;                   // (by list position: see wf_shade)
                if (L.rng.draws) { pv.store_rng(slot, L.rng.s); cnt.draws(L.rng.draws); }
                t_flag(L, kHasRay, false);
            }
            const unsigned long long m = wballot(true);
            const int leader = __ffsll((long long)m) - 1;
            uint32_t need = (uint32_t)__popcll(m);
            uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
#if RT2022_REFILL_TOUCH
            const uint32_t rank0 = rank;
#endif
            uint32_t entry_idx = 0xFFFFFFFFu;                         // index into pool.list of the entry this lane takes
            const u32x4 cs_now = *cs;
            uint32_t ch_base = cs_now.x, ch_n = cs_now.y, ch_taken = cs_now.z;
            bool drained = cs_now.w != 0;
            for (;;) {
                const uint32_t avail = ch_n - ch_taken;
                if (entry_idx == 0xFFFFFFFFu) {
                    if (rank < avail) entry_idx = ch_base + ch_taken + rank;
                    else rank -= avail;
                }
                const uint32_t take = need < avail ? need : avail;
                ch_taken += take;
                need -= take;
                if (need == 0 || drained) break;
                uint32_t id = 0;                                      // next chunk
                if ((int)lane == leader) id = atomicAdd(pool.next_chunk, 1u);
                id = (uint32_t)__shfl((int)id, leader);
                if (id >= total_ids) { drained = true; break; }
                const uint32_t slice = id / n_seg, seg = id - slice * n_seg, first = slice * kChunk;
                const uint32_t n = pool.list_n[seg];
                ch_n = n > first ? (n - first < kChunk ? n - first : kChunk) : 0u;
                ch_base = seg * (uint32_t)S + first;
                ch_taken = 0;
            }
            if ((int)lane == leader) *cs = (u32x4){ch_base, ch_n, ch_taken, drained ? 1u : 0u};
            if (probe && !dry_seen && drained) { dry_seen = true; t_dry = wall_clock64(); }
#if RT2022_REFILL_TOUCH
            // (Off by default: measured +0.3 % on the headline and +2 % on C2, for 35 % more HBM reads by the counters.)
            // Touch-ahead: the entries this wave's NEXT refill round will hand out follow the ones handed out now; each
            // refilling lane asks for one word of the ray record of the entry `its rank` places further on, beside its own
            // fetches (same two dependent round trips, issued in parallel) — the next round's records then come from L2 or
            // the Infinity Cache instead of HBM. The word is never used: the empty asm at the head of the next refill
            // gives the load a consumer.
            asm volatile("" :: "v"(touch_word));
            const uint32_t touch_entry = ch_base + ch_taken + rank0;
            const bool touch = touch_entry < ch_base + ch_n;
            uint32_t touch_local = 0;
            if (touch) touch_local = pool.list[touch_entry];
#endif
            if (entry_idx != 0xFFFFFFFFu) {
                const uint32_t sbase = entry_idx & ~((uint32_t)S - 1u);
                L.entry = entry_idx;
                L.slot = sbase + pool.list[entry_idx];
                uint64_t rs;
                Ray wr = pv.load_ray(L.slot, rs);
#if RT2022_REFILL_TOUCH
                asm volatile("" ::: "memory");
                if (touch) touch_word = *reinterpret_cast<const uint32_t *>(pool.ray + (uint64_t)((touch_entry & ~((uint32_t)S - 1u)) + touch_local) * kRecDoubles);
                asm volatile("" ::: "memory");
#endif
                L.tm = wr.tm;
                L.rng = Rng(rs);
                t_set_cur(L, XRay{wr.orig, wr.dir}, boxes_plain);
                if (kSlabs) t_slabs(L, table_at); if (kF32) t_slabs32(L, table_at); if (kF32G) t_slabs32g(L);
                if (FEAT & kFeatMovers) L.stash_level = 0xFFFFFFFFu;
                if (kStash) {                                         // what leaving a mover goes back to (OP_CTX)
                    wray[0 * WG] = wr.orig.x; wray[1 * WG] = wr.orig.y; wray[2 * WG] = wr.orig.z;
                    wray[3 * WG] = wr.dir.x; wray[4 * WG] = wr.dir.y; wray[5 * WG] = wr.dir.z;
                }
                L.closest = rtm::F64_MAX;
                L.steps = 0;
                L.t_lo = t_min; L.med_ref = 0;
                L.win_leaf = REF_EMPTY; L.win_face = 0;
                L.ctx.n = 0;
                L.sp = 0;
                L.top = s.root;
                T_SETTLE();
                t_flag(L, kHasRay, true);
            } else {
                L.op = OP_IDLE;
            }
        }
        TP_MARK(2 + best);
    }
    TP_FLUSH();
#ifdef RT2022_F32_CENSUS
    if (f32_steps) atomicAdd(&g_f32_census[0], (unsigned long long)f32_steps);
    if (f32_undecided) atomicAdd(&g_f32_census[1], (unsigned long long)f32_undecided);
    if (f32_wrong) atomicAdd(&g_f32_census[2], (unsigned long long)f32_wrong);
#endif
    if (probe && lane == 0) {
        unsigned long long t_end = wall_clock64();
        atomicMin(&pool.dbg[0], t_start);
        atomicMax(&pool.dbg[1], t_end);
        atomicAdd(&pool.dbg[2], t_end - t_start);
        atomicAdd(&pool.dbg[3], t_end - (dry_seen ? t_dry : t_end));
        atomicAdd(&pool.dbg[4], 1ull);
        if (tid == 0) { pool.dbg[8 + 2 * blockIdx.x] = t_start; pool.dbg[9 + 2 * blockIdx.x] = t_end; }
    }
    if (STATS) {
        cnt.flush_wave(stats);
        if (stats)                                                // (each lane adds what it counted as a round's first lane)
            for (int o = 0; o < 9; o++) {
                if (census_rounds[o]) atomicAdd(&stats->op_rounds[o], (unsigned long long)census_rounds[o]);
                if (census_lanes[o]) atomicAdd(&stats->op_lanes[o], (unsigned long long)census_lanes[o]);
            }
    }
}

// Marks the first `used` slots of every workgroup FRESH and the rest IDLE: a small job is spread
// over all workgroups (a few slots each) instead of filling a few workgroups to the brim.
__global__ void __launch_bounds__(256) wf_init(uint8_t *kind, uint16_t *list, uint32_t *list_n, uint32_t n_slots, uint32_t used) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots) {                                            // the first "ray list" of a segment: its slots in use, all FRESH
        const uint32_t local = i % (uint32_t)S;
        kind[i] = local < used ? (uint8_t)SK_FRESH : (uint8_t)SK_IDLE;
        list[i] = (uint16_t)local;
        if (local == 0) list_n[i / (uint32_t)S] = used;
    }
}

// ---- host side of the engine -------------------------------------------------------------
struct WfLaunch {
    SceneDev scene;
    WfPool pool;                // this group's view of the pool
    const RenderArgs *d_args;
    double t_min;
    uint32_t node_quorum;
    uint32_t vote_weights;
    StatsDev *stats;
    uint32_t blocks;            // segments of the group
    hipStream_t stream;
    bool ring = false;          // RenderArgs::ring in use: the shade pass's ring build
};
template <bool STATS>
static void launch_shade(const WfLaunch &w, uint32_t parity) {
    if (w.ring) hipLaunchKernelGGL((wf_shade<STATS, true>), dim3(w.blocks), dim3(kBlock), 0, w.stream, w.scene, w.d_args, w.pool, parity);
    else hipLaunchKernelGGL((wf_shade<STATS, false>), dim3(w.blocks), dim3(kBlock), 0, w.stream, w.scene, w.d_args, w.pool, parity);
}
// Ring mode: out[i] = (first plane of the frame ? 0 : out[i]) + partial[first mod R][i] + ... in sample order — pixel_color += ...,
// main.rs:150, continued where the last call of this kernel left off (chunk_sum_kernel's sum, taken a few planes at a time).
__global__ void __launch_bounds__(256) ring_accumulate_kernel(const double *partial, double *out, uint64_t n_values, uint32_t first, uint32_t count, uint32_t ring) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_values; i += stride) {
        double acc = first == 0 ? 0.0 : out[i];
        for (uint32_t c = first; c < first + count; c++) acc += partial[(uint64_t)(c % ring) * n_values + i];
        out[i] = acc;
    }
}
__global__ void ring_set_limit_kernel(unsigned long long *limit, unsigned long long value) { *limit = value; }
template <int STACK, bool STATS, unsigned FEAT, bool PROBE = false>
static void launch_trace(const WfLaunch &w, uint32_t parity) {
    // A persistent grid: as many workgroups as the kernel's launch bounds keep resident, never more than the work
    // (a segment holds at most 4096 / kChunk chunks for the 4 waves of a workgroup).
    constexpr uint32_t per_cu = trace_blocks_per_cu(STACK, STATS, FEAT);
    uint32_t grid = per_cu * (w.pool.n_cus ? w.pool.n_cus : 1u);
    const uint32_t most = w.blocks * ((uint32_t)S / kChunk / 4u);
    if (grid > most) grid = most;
    if constexpr (FEAT == 0 && !STATS && !PROBE) {
        if (w.scene.n_rects == 0) {                                   // a sphere-only scene: the instance that tests node boxes in single precision (kF32G)
            hipLaunchKernelGGL((wf_trace<STACK, STATS, FEAT, PROBE, kBlock, 0, false, false, true>), dim3(grid), dim3(kBlock), 0, w.stream, w.scene, w.pool, w.t_min,
                               w.node_quorum, parity, w.stats, w.vote_weights);
            return;
        }
    }
    hipLaunchKernelGGL((wf_trace<STACK, STATS, FEAT, PROBE>), dim3(grid), dim3(kBlock), 0, w.stream, w.scene, w.pool, w.t_min,
                       w.node_quorum, parity, w.stats, w.vote_weights);
}
// The node-cache variant: one workgroup of kCacheBlock threads per CU (see wf_trace).
template <unsigned FEAT, int STACK, int CACHE, bool PARTIAL>
static void launch_trace_cached(const WfLaunch &w, uint32_t parity) {
    uint32_t grid = w.pool.n_cus ? w.pool.n_cus : 1u;
    const uint32_t most = std::max(1u, w.blocks * ((uint32_t)S / kChunk) / (uint32_t)(kCacheBlock / 64));
    if (grid > most) grid = most;
    if constexpr (FEAT == 0 && !PARTIAL) {
        if (w.scene.n_rects == 0) {                                   // (a sphere-only scene: the single-precision records of t_slabs32 in the table)
            hipLaunchKernelGGL((wf_trace<STACK, false, FEAT, false, kCacheBlock, CACHE, PARTIAL, false, true>), dim3(grid), dim3(kCacheBlock), 0, w.stream,
                               w.scene, w.pool, w.t_min, w.node_quorum, parity, w.stats, w.vote_weights);
            return;
        }
    }
    hipLaunchKernelGGL((wf_trace<STACK, false, FEAT, false, kCacheBlock, CACHE, PARTIAL>), dim3(grid), dim3(kCacheBlock), 0, w.stream,
                       w.scene, w.pool, w.t_min, w.node_quorum, parity, w.stats, w.vote_weights);
}
// The all-in-LDS instance for sphere-only scenes (FEAT = 0): node table of kPrimNodes records and both sphere pools.
static void launch_trace_prims(const WfLaunch &w, uint32_t parity) {
    uint32_t grid = w.pool.n_cus ? w.pool.n_cus : 1u;
    const uint32_t most = std::max(1u, w.blocks * ((uint32_t)S / kChunk) / (uint32_t)(kCacheBlock / 64));
    if (grid > most) grid = most;
    hipLaunchKernelGGL((wf_trace<kStackTiny, false, 0, false, kCacheBlock, kPrimNodes, false, true>), dim3(grid), dim3(kCacheBlock), 0, w.stream,
                       w.scene, w.pool, w.t_min, w.node_quorum, parity, w.stats, w.vote_weights);
}
template <int STACK, int CACHE, bool PARTIAL>
static void launch_trace_cached_feat(unsigned feat, const WfLaunch &w, uint32_t parity) {
    switch (feat & 7u) {
        case 0: launch_trace_cached<0, STACK, CACHE, PARTIAL>(w, parity); break;
        case 1: launch_trace_cached<1, STACK, CACHE, PARTIAL>(w, parity); break;
        case 2: launch_trace_cached<2, STACK, CACHE, PARTIAL>(w, parity); break;
        case 3: launch_trace_cached<3, STACK, CACHE, PARTIAL>(w, parity); break;
        case 4: launch_trace_cached<4, STACK, CACHE, PARTIAL>(w, parity); break;
        case 5: launch_trace_cached<5, STACK, CACHE, PARTIAL>(w, parity); break;
        case 6: launch_trace_cached<6, STACK, CACHE, PARTIAL>(w, parity); break;
        default: launch_trace_cached<7, STACK, CACHE, PARTIAL>(w, parity); break;
    }
}
// Whether a scene takes the node-cache variant: its stacks fit the variant's, and its node table fits the cache whole.
// (Bit 28 of the tuning word — rt_debug_set_tuning — or RT2022_NODE_CACHE=0 in the environment keeps the plain kernels:
// A/B runs, and the test that the two give the same bits.)
// 0: the plain kernels; 1: the whole table (stacks of 16); 2: its first kNodeCache records (stacks of 16); 3: a
// sphere-only scene whose node table and sphere pools all fit (RT2022_PRIM_TABLES=0 in the environment: mode 1 instead).
// The scenes whose plain kernels test node boxes in single precision on SceneDev::nodes32 (wf_trace: kF32G).
static bool f32_from_hbm(const SceneDev &scene, unsigned features) {
    return RT2022_F32_SLABS >= 1 && (RT2022_F32_GLOBAL == 2 ? !(features & kFeatVolumes)
           : (RT2022_F32_GLOBAL == 1 && ((features == 0 && scene.n_rects == 0) || ((features & kFeatMisc) && !(features & kFeatVolumes)))));
}
static int node_cache_mode(const SceneDev &scene, uint32_t stack_need, uint32_t tuning, unsigned features) {
    static const bool enabled = [] { const char *e = getenv("RT2022_NODE_CACHE"); return !(e && e[0] == '0'); }();
    static const bool prims = [] { const char *e = getenv("RT2022_PRIM_TABLES"); return !(e && e[0] == '0'); }();
    if (!enabled || (tuning & (1u << 28)) || stack_need > (uint32_t)kStackTiny) return 0;
    if (prims && features == 0 && scene.n_rects == 0 && scene.n_nodes <= (uint32_t)kPrimNodes && scene.n_spheres <= (uint32_t)kPrimSpheres &&
        scene.n_moving_spheres <= (uint32_t)kPrimMoving) return 3;
    if (scene.n_nodes <= (uint32_t)kNodeCache) return 1;
    // A scene whose nodes are tested in single precision from 32-byte records (sphere-only, or a triangle mesh: wf_trace, kF32G) takes
    // the plain kernels when its table does not fit whole: five waves per SIMD there against four here, and half the bytes per node
    // step either way — the partial table measured 10 % slower (1e4 spheres: 1 854 against 2 039 Mrays/s, profiles/r3zp_partial_vs_plain.log).
    if (f32_from_hbm(scene, features)) return 0;
    return 2;
}
template <int STACK, bool PROBE = false>
static void launch_trace_feat(unsigned feat, const WfLaunch &w, uint32_t parity) {
    switch (feat & 7u) {
        case 0: launch_trace<STACK, false, 0, PROBE>(w, parity); break;
        case 1: launch_trace<STACK, false, 1, PROBE>(w, parity); break;
        case 2: launch_trace<STACK, false, 2, PROBE>(w, parity); break;
        case 3: launch_trace<STACK, false, 3, PROBE>(w, parity); break;
        case 4: launch_trace<STACK, false, 4, PROBE>(w, parity); break;
        case 5: launch_trace<STACK, false, 5, PROBE>(w, parity); break;
        case 6: launch_trace<STACK, false, 6, PROBE>(w, parity); break;
        default: launch_trace<STACK, false, 7, PROBE>(w, parity); break;
    }
}
static void launch_pass(const WfLaunch &w, uint32_t parity, uint32_t stack_need, unsigned features, bool counters, bool probe,
                        hipEvent_t between = nullptr) {
    if (counters) launch_shade<true>(w, parity);
    else launch_shade<false>(w, parity);
    if (between) (void)hipEventRecord(between, w.stream);
    const int table = (counters || probe) ? 0 : node_cache_mode(w.scene, stack_need, w.node_quorum, features);
    if (table == 3) {
        launch_trace_prims(w, parity);
    } else if (table == 1) {
        launch_trace_cached_feat<kStackTiny, kNodeCache, false>(features, w, parity);
    } else if (table == 2) {
        launch_trace_cached_feat<kStackTiny, kNodeCache, true>(features, w, parity);
    } else if (stack_need <= (uint32_t)kStackSmall) {
        if (counters) launch_trace<kStackSmall, true, 7>(w, parity);
        else if (probe) launch_trace_feat<kStackSmall, true>(features, w, parity);   // (the probe exists per feature set for the small stack only)
        else launch_trace_feat<kStackSmall>(features, w, parity);
    } else if (stack_need <= (uint32_t)kStackMid) {
        if (counters) launch_trace<kStackMid, true, 7>(w, parity);
        else if (probe) launch_trace<kStackMid, false, 7, true>(w, parity);
        else launch_trace_feat<kStackMid>(features, w, parity);
    } else {
        if (counters) launch_trace<kStackLarge, true, 7>(w, parity);
        else if (probe) launch_trace<kStackLarge, false, 7, true>(w, parity);
        else launch_trace_feat<kStackLarge>(features, w, parity);
    }
}

// The pool is cut into groups of segments, each on a stream of its own and alternating shade / trace passes
// at its own pace: nothing couples the groups but the work counter, so while the last long rays of one
// group's trace pass keep a few waves busy, the other groups' kernels fill the rest of the chip. (Measured
// with one group: the mean wave lives 0.41 of a trace pass — rt_debug_pass_timing.)
static hipError_t render_passes(const SceneDev &scene, const RenderArgs &args, const RenderArgs *d_args,
                                const WfPool &pool, uint32_t stack_need, unsigned features, bool counters,
                                const WfStreams &gs, hipStream_t stream, uint32_t *out_iterations, double *timing, KernelTimes *kt,
                                const Progress *progress, const RingCtl *ring) {
    if (stack_need > (uint32_t)kStackLarge) return hipErrorInvalidValue;
    const bool report = progress && progress->cb && gs.h_work;
    unsigned long long reported = 0;
    // Ring of partial-sum planes (RenderArgs::ring): planes consumed so far, and the claim limit that follows them.
    const bool ringed = ring && ring->planes > 0 && args.ring == ring->planes && gs.h_work && gs.h_oldest && args.ring_group > 0 &&
                        args.ring % args.ring_group == 0 && args.n_chunks % args.ring_group == 0;
    if (ring && ring->planes > 0 && !ringed) return hipErrorInvalidValue;
    uint32_t consumed = 0;                      // planes added to the output so far (a multiple of the sample group)
    const uint64_t n_values = args.n_pixels * 3;
    const uint64_t per_group = ringed ? args.n_pixels * args.ring_group : 1;         // work items of one sample group
    auto ring_limit = [&](uint32_t done) {      // the groups whose planes are free: those consumed, and R planes' worth beyond them
        const unsigned long long lim = ((unsigned long long)done + ring->planes) / args.ring_group * per_group;
        return lim < args.n_items ? lim : (unsigned long long)args.n_items;
    };
    auto ring_consume = [&](uint32_t upto, hipStream_t st) -> hipError_t {          // planes [consumed, upto) are complete
        if (upto <= consumed) return hipSuccess;
        const uint64_t want = (n_values + 255) / 256;
        hipLaunchKernelGGL(ring_accumulate_kernel, dim3((unsigned)(want > 2048 ? 2048 : (want ? want : 1))), dim3(256), 0, st,
                           args.partial, ring->out, n_values, consumed, upto - consumed, ring->planes);
        consumed = upto;
        hipLaunchKernelGGL(ring_set_limit_kernel, dim3(1), dim3(1), 0, st, ring->d_limit, ring_limit(consumed));
        return hipGetLastError();
    };
    const uint32_t blocks = pool.n_blocks;
    hipError_t e;
    // Slots in use start FRESH (at most one work item per slot is ever needed at a time).
    {
        uint64_t per_block = (args.n_items + blocks - 1) / blocks;
        uint32_t used = (uint32_t)(per_block > (uint64_t)S ? (uint64_t)S : (per_block + 63) / 64 * 64);
        if (used < 64) used = 64;
        uint32_t n = blocks * (uint32_t)S;
        hipLaunchKernelGGL(wf_init, dim3((n + 255) / 256), dim3(256), 0, stream, pool.kind, pool.list, pool.list_n, n, used);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipMemsetAsync(pool.n_active, 0, 2 * kMaxGroups * sizeof(uint32_t), stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(pool.max_list, 0, 2 * kMaxGroups * sizeof(uint32_t), stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(pool.fault, 0, sizeof(uint32_t), stream)) != hipSuccess) return e;
        if (ringed) {
            if ((e = hipMemsetAsync(pool.oldest, 0xFF, 2 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
            if ((e = hipMemsetAsync(pool.starved_n, 0, blocks * sizeof(uint32_t), stream)) != hipSuccess) return e;
            hipLaunchKernelGGL(ring_set_limit_kernel, dim3(1), dim3(1), 0, stream, ring->d_limit, ring_limit(0));
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
    }
    const uint32_t trace_blocks = blocks / pool.segs;
    int G = (timing || kt || ringed) ? 1 : gs.n;       // (per-kernel times: one group, so that a launch's duration is its own; with several, launches of different groups overlap)
    if (G < 1) G = 1;
    if ((uint32_t)G > trace_blocks) G = (int)trace_blocks;
    WfLaunch w[kMaxGroups];
    uint32_t iter[kMaxGroups] = {};
    uint32_t batches[kMaxGroups] = {};          // batches enqueued
    bool drained[kMaxGroups] = {};
    for (int g = 0; g < G; g++) {
        const uint32_t tb0 = (uint32_t)((uint64_t)trace_blocks * g / G), tb1 = (uint32_t)((uint64_t)trace_blocks * (g + 1) / G);
        const uint32_t seg_begin = tb0 * pool.segs, n_segs = (tb1 - tb0) * pool.segs;
        const uint64_t off = (uint64_t)seg_begin * S;
        WfPool v = pool;
        v.n_blocks = n_segs; v.n_slots = n_segs * (uint32_t)S;
        v.kind += off; v.ray += off * kRecDoubles; v.hit += off * kRecWords; v.state += off * kRecWords; v.pixel_sum += off * 4;
        v.tape += off * pool.tape_cap * 4; v.list += off; v.list_n += seg_begin;
        v.n_active = pool.n_active + 2 * g;
        v.next_chunk = pool.next_chunk + g;
        v.max_list = pool.max_list + 2 * g;
        w[g] = WfLaunch{scene, v, d_args, args.t_min, args.node_quorum, args.vote_weights, args.stats, n_segs, G == 1 ? stream : gs.stream[g], ringed};
    }
    if (G > 1) {                                // the groups start after what the caller's stream holds so far
        if ((e = hipEventRecord(gs.ev[0][0], stream)) != hipSuccess) return e;
        for (int g = 0; g < G; g++)
            if ((e = hipStreamWaitEvent(gs.stream[g], gs.ev[0][0], 0)) != hipSuccess) return e;
        if ((e = hipEventSynchronize(gs.ev[0][0])) != hipSuccess) return e;     // (the event is reused below)
    }
    const uint32_t poll_every = timing ? 1 : 4;
    uint32_t iterations = 0;
    // One batch = poll_every passes + a read-back of the group's "rays handed on" counter. Two batches per group
    // are kept in flight so that a group's stream never runs empty while the host looks at the previous answer.
    auto enqueue_batch = [&](int g) -> hipError_t {
        for (uint32_t k = 0; k < poll_every; k++) {
            if (timing) {
                if ((e = hipMemsetAsync(pool.dbg, 0xFF, sizeof(unsigned long long), w[g].stream)) != hipSuccess) return e;
                if ((e = hipMemsetAsync(pool.dbg + 1, 0, 4 * sizeof(unsigned long long), w[g].stream)) != hipSuccess) return e;
            }
            // (kernel times: three events per pass pair k, on the stream of its group — 3k before its shade pass, 3k+1 between its shade
            // and trace pass, 3k+2 after: with several groups the pairs of different groups overlap, each launch's own duration is what
            // is summed)
            hipEvent_t mid = nullptr;
            if (kt) {
                while (kt->ev.size() < 3 * (size_t)iterations + 3) {
                    hipEvent_t ne = nullptr;
                    if ((e = hipEventCreate(&ne)) != hipSuccess) return e;
                    kt->ev.push_back(ne);
                }
                if ((e = hipEventRecord(kt->ev[3 * iterations], w[g].stream)) != hipSuccess) return e;
                mid = kt->ev[3 * iterations + 1];
            }
            launch_pass(w[g], iter[g] & 1u, stack_need, features, counters, timing != nullptr, mid);
            if (kt && (e = hipEventRecord(kt->ev[3 * iterations + 2], w[g].stream)) != hipSuccess) return e;
            iter[g]++;
            iterations++;
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        const uint32_t b = batches[g]++ & 1u;   // ring of two: batches of a group complete in order
        if ((e = hipMemcpyAsync(gs.h_active + 2 * g + b, w[g].pool.n_active + ((iter[g] - 1) & 1u), sizeof(uint32_t),
                                hipMemcpyDeviceToHost, w[g].stream)) != hipSuccess) return e;
        if ((report || ringed) && (e = hipMemcpyAsync(gs.h_work + 2 * g + b, args.work_counter, sizeof(unsigned long long),
                                                      hipMemcpyDeviceToHost, w[g].stream)) != hipSuccess) return e;
        if (ringed && (e = hipMemcpyAsync(gs.h_oldest + 2 * g + b, w[g].pool.oldest + ((iter[g] - 1) & 1u), sizeof(unsigned long long),
                                          hipMemcpyDeviceToHost, w[g].stream)) != hipSuccess) return e;
        return hipEventRecord(gs.ev[g][b], w[g].stream);
    };
    uint32_t waited[kMaxGroups] = {};           // batches whose answer has been read
    for (int g = 0; g < G; g++) {
        if ((e = enqueue_batch(g)) != hipSuccess) return e;
        if (!timing && (e = enqueue_batch(g)) != hipSuccess) return e;
    }
    int live = G;
    while (live > 0) {
        for (int g = 0; g < G; g++) {
            if (drained[g]) continue;
            const uint32_t b = waited[g] & 1u;
            if ((e = hipEventSynchronize(gs.ev[g][b])) != hipSuccess) return e;
            waited[g]++;
            if (report) {                       // work items handed out so far -> camera paths started (main.rs:154-155: the bar's inc)
                unsigned long long items = gs.h_work[2 * g + b];
                if (items > args.n_items) items = args.n_items;       // (the counter overshoots at the end of the work)
                unsigned long long paths = items * progress->per_item;
                if (paths > progress->total) paths = progress->total;
                if (paths > reported && paths < progress->total) { reported = paths; progress->cb(progress->user, 0u, paths, progress->total); }
            }
            if (timing) {
                unsigned long long h[5];
                if ((e = hipMemcpy(h, pool.dbg, sizeof h, hipMemcpyDeviceToHost)) != hipSuccess) return e;
                if (h[4]) {
                    timing[0] += (double)(h[1] - h[0]);                // span of the pass
                    timing[1] += (double)h[2] / (double)h[4];          // mean wave lifetime
                    timing[2] += (double)h[3] / (double)h[4];          // mean wave time after its list ran dry
                    timing[3] += 1.0;
                    timing[4] += (double)h[4];
                    if (getenv("RT2022_PASS_LOG") && (iter[g] == 50 || iter[g] == 51 || iter[g] == 80)) {      // wave 0 of every workgroup
                        const uint32_t nb = (uint32_t)(h[4] / 4);                 // (workgroups of the persistent grid that ran)
                        std::vector<unsigned long long> bt(2 * nb);
                        if (hipMemcpy(bt.data(), pool.dbg + 8, bt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                            std::vector<double> st_, en_;
                            for (uint32_t b = 0; b < nb; b++) {
                                st_.push_back((double)(bt[2 * b] - h[0]) / 100.0); en_.push_back((double)(bt[2 * b + 1] - h[0]) / 100.0);
                            }
                            if (const char *dump = getenv("RT2022_BLOCK_DUMP")) {
                                char name[512];
                                snprintf(name, sizeof name, "%s.%u", dump, iter[g]);
                                if (FILE *f = fopen(name, "w")) { for (uint32_t b = 0; b < nb; b++) fprintf(f, "%u %.1f %.1f\n", b, st_[b], en_[b]); fclose(f); }
                            }
                            std::sort(st_.begin(), st_.end()); std::sort(en_.begin(), en_.end());
                            fprintf(stderr, "pass %u workgroups %u: start us p0 %.1f p50 %.1f p100 %.1f | end us p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f\n", iter[g], nb,
                                    st_[0], st_[nb / 2], st_[nb - 1], en_[0], en_[nb / 10], en_[nb / 2], en_[nb * 9 / 10], en_[nb - 1]);
                        }
                    }
                    if (getenv("RT2022_PASS_LOG"))
                        fprintf(stderr, "pass %u span_us %.1f life/span %.3f dry/life %.3f waves %llu rays %u\n", iter[g], (double)(h[1] - h[0]) / 100.0,
                                (double)h[2] / (double)h[4] / (double)(h[1] - h[0]), (double)h[3] / (double)(h[2] ? h[2] : 1), h[4], gs.h_active[2 * g + b]);
                }
            }
            if (ringed) {
                // Everything below the oldest item in flight (and below the counter: items not handed out yet are not in flight
                // either) is finished: whole planes under that frontier go to the output, and the limit follows them. (The words
                // were copied behind the batch's last shade pass; the kernels launched here run behind the batches already queued,
                // whose claims still obey the old limit.)
                unsigned long long groups_done = (gs.h_work[2 * g + b] < args.n_items ? gs.h_work[2 * g + b] : (unsigned long long)args.n_items) / per_group;
                if (gs.h_oldest[2 * g + b] < groups_done) groups_done = gs.h_oldest[2 * g + b];
                if ((e = ring_consume((uint32_t)groups_done * args.ring_group, w[g].stream)) != hipSuccess) return e;
                if (ring->max_passes && iterations > ring->max_passes) return hipErrorUnknown;     // (a frame cannot take this long: never spin)
            }
            if (gs.h_active[2 * g + b] == 0) {  // the batch's last shade pass handed no ray on: the group has drained
                drained[g] = true;
                live--;
                continue;
            }
            if (iter[g] > (1u << 26)) return hipErrorUnknown;
            if ((e = enqueue_batch(g)) != hipSuccess) return e;
        }
    }
    if (ringed && (e = ring_consume(args.n_chunks, w[0].stream)) != hipSuccess) return e;      // (nothing is in flight any more)
    for (int g = 0; g < G; g++)                 // (a drained group may still have an idle batch queued)
        if ((e = hipStreamSynchronize(w[g].stream)) != hipSuccess) return e;
    if (kt) {                                   // device time of the shade passes and of the trace passes
        kt->shade_ms = kt->trace_ms = 0.0;
        for (uint32_t k = 0; k < iterations; k++) {
            float a = 0.f, b = 0.f;
            if ((e = hipEventElapsedTime(&a, kt->ev[3 * k], kt->ev[3 * k + 1])) != hipSuccess) return e;
            if ((e = hipEventElapsedTime(&b, kt->ev[3 * k + 1], kt->ev[3 * k + 2])) != hipSuccess) return e;
            kt->shade_ms += (double)a;
            kt->trace_ms += (double)b;
        }
    }
    if (out_iterations) *out_iterations = iterations;
    return hipSuccess;
}

hipError_t f32_slab_census(unsigned long long out[5]) {
    unsigned long long c[3] = {0, 0, 0};
    hipError_t e = hipMemcpyFromSymbol(c, HIP_SYMBOL(g_f32_census), sizeof(c));
    if (e != hipSuccess) return e;
    const unsigned long long zero[3] = {0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_f32_census), zero, sizeof(zero));
    out[0] = c[0]; out[1] = c[1]; out[4] = c[2];
#ifdef RT2022_F32_CENSUS
    out[2] = 1;
#else
    out[2] = 0;
#endif
    out[3] = RT2022_F32_SLABS;
    return e;
}

void trace_variant(const SceneDev &scene, uint32_t stack_need, uint32_t tuning, unsigned features, uint32_t out[4]) {
    out[3] = 0;
    const int table = node_cache_mode(scene, stack_need, tuning, features);
    // Which instances test node boxes in single precision (wf_trace: kF32, kF32G) — bit 1 of out[3]
    const bool f32 = table == 3 ? RT2022_F32_SLABS == 1 || RT2022_F32_SLABS == 2
                   : table == 1 ? (RT2022_F32_SLABS == 2 && !(features & kFeatMisc)) || (RT2022_F32_SLABS == 1 && features == 0 && scene.n_rects == 0)
                   : table == 0 ? f32_from_hbm(scene, features) : false;
    if (table) {
        out[0] = (uint32_t)kCacheBlock; out[1] = (uint32_t)kStackTiny; out[2] = scene.n_nodes < (uint32_t)kNodeCache ? scene.n_nodes : (uint32_t)kNodeCache;
        out[3] = (table == 3 ? 1u : 0u) | (f32 ? 2u : 0u);
        return;
    }
    out[0] = (uint32_t)kBlock;
    out[1] = stack_need <= (uint32_t)kStackSmall ? (uint32_t)kStackSmall : stack_need <= (uint32_t)kStackMid ? (uint32_t)kStackMid : (uint32_t)kStackLarge;
    out[2] = 0;
    out[3] = f32 ? 2u : 0u;
}

hipError_t launch_render_wavefront(const SceneDev &scene, const RenderArgs &args, const RenderArgs *d_args,
                                   const WfPool &pool, uint32_t stack_need, unsigned features, bool counters,
                                   const WfStreams &gs, hipStream_t stream, uint32_t *out_iterations, double *timing,
                                   uint32_t *out_fault, KernelTimes *kt, const Progress *progress, const RingCtl *ring) {
    hipError_t e = render_passes(scene, args, d_args, pool, stack_need, features, counters, gs, stream, out_iterations, timing, kt, progress, ring);
    if (e != hipSuccess) {
        // Passes may still be queued or running against the pool on the group streams: let them finish (best
        // effort) before the caller sees the error and possibly frees or reuses the pool.
        for (int g = 0; g < kMaxGroups; g++)
            if (gs.stream[g]) (void)hipStreamSynchronize(gs.stream[g]);
        (void)hipStreamSynchronize(stream);
        return e;
    }
#ifdef RT2022_TRACE_PROBE
    if (pool.dbg) {
        unsigned long long h[12];
        if (hipMemcpy(h, pool.dbg + 96, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            static const char *const names[12] = {"node_fast", "vote", "node", "sphere", "rect", "box", "medium", "misc", "ctx", "done", "rest", "-"};
            double tot = 0; for (int i = 0; i < 11; i++) tot += (double)h[i];
            fprintf(stderr, "trace probe (shader-clock ticks of all waves and passes; share):");
            for (int i = 0; i < 11; i++) fprintf(stderr, " %s %.3f", names[i], tot > 0 ? (double)h[i] / tot : 0.0);
            fprintf(stderr, "  total %.3e ticks\n", tot);
        }
        (void)hipMemset(pool.dbg + 96, 0, 12 * sizeof(unsigned long long));
    }
#endif
#ifdef RT2022_SHADE_PROBE
    if (pool.dbg) {
        unsigned long long h[10];
        if (hipMemcpy(h, pool.dbg + 64, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            double tot = 0; for (int i = 0; i < 8; i++) tot += (double)h[i];
            fprintf(stderr, "shade probe (ticks of wave 0, all workgroups and passes; share):");
            for (int i = 0; i < 8; i++) fprintf(stderr, " [%d] %.3f", i, tot > 0 ? (double)h[i] / tot : 0.0);
            fprintf(stderr, "  total %.3e ticks\n", tot);
        }
        (void)hipMemset(pool.dbg + 64, 0, 10 * sizeof(unsigned long long));
    }
#endif
    uint32_t fault = 0;
    if ((e = hipMemcpy(&fault, pool.fault, sizeof fault, hipMemcpyDeviceToHost)) != hipSuccess) return e;
    if (out_fault) *out_fault = fault;
    return hipSuccess;
}

} // namespace rt2022

// pt_kernel.hip — the path-tracing megakernel for gfx950 (MI355X, CDNA4).
//
// One persistent launch renders a whole batch of rows. Every lane of every 64-wide
// wavefront owns one path at a time and runs the reference's recursion
// (raytracer/src/main.rs:233-278) as an iterative bounce loop; when a path ends the
// lane moves to the next sample of its work item (pixel x sample-chunk), and when the
// item ends the wave refills the idle lanes from a global work counter with one atomic
// per wave (__ballot / __popcll / __shfl). The wave schedules itself: lanes are labelled
// with their next operation and the most common one runs (see "The megakernel" below).
// BVH traversal uses an explicit per-lane stack in LDS ([depth][lane] -> bank = lane,
// conflict-free), visits nodes in exactly the reference's order (left subtree, then right
// with t_max = closest so far, hittable/bvh/mod.rs:86-101) because ConstantMedium::hit
// draws from the RNG during traversal (constantmedium.rs:60), and defers the hit record
// (normal, uv) to the single winning candidate. All arithmetic is f64 through rt_math.h,
// so every path is bit-identical to the CPU oracle's.
//
// No MFMA: this is branchy scalar f64, not a contraction.
#include "pt_common.hpp"

namespace rt2022 {

namespace {

// =====================================================================================
// The megakernel: an in-wave scheduled state machine.
//
// Every lane carries one path and a label `op` naming the next thing it has to do:
// test the BVH node on top of its stack, test a sphere / rect / box / medium leaf,
// enter or leave a mover, or shade. Lanes of a wave are at different points of
// different paths, so a plain "each lane runs its own switch" loop executes every
// arm serially with a handful of lanes each (measured: 8 % lane utilisation). Here the
// wave votes instead: it counts the lanes per label (__ballot + __popcll), runs the
// most popular arm once with all the lanes that wait for it, and lets the others
// stay parked — a lane's own sequence of operations (and therefore its RNG stream
// and hit order) never changes, only when it gets its turn.
// =====================================================================================
struct Lane {
    // work item: pixel slot x sample chunk
    uint64_t slot;
    uint32_t chunk_id, smp, smp_end, px, py, frame;
    Vec3 pixel_sum;
    bool have_item, alive;
    // path
    Ray r;                 // the ray being traced (world frame)
    int depth;             // remaining depth (ray_color's `depth`)
    uint32_t nb;           // bounce records on the tape
    Rng rng;
    // traversal
    XRay cur;              // r inside the enclosing movers
    Vec3 inv;              // 1 / cur.d (aabb.rs:19, hoisted: same value at every node)
    double a_len;          // cur.d.length_sqr() (sphere.rs:41, hoisted likewise)
    double closest;
    bool found;
    Winner win;
    Chain ctx;
    int sp;
    uint32_t top;          // entry being processed (popped from the stack)
    uint32_t op;
};

template <int STACK>
struct Stack {
    uint32_t *col;         // this lane's column: entry d at col[d * kBlock]
    RT_DEV void push(Lane &L, uint32_t ref) { if (L.sp < STACK) { col[L.sp * kBlock] = ref; L.sp++; } }
    RT_DEV uint32_t pop(Lane &L) { if (L.sp > 0) { L.sp--; return col[L.sp * kBlock]; } return REF_EMPTY; }
};

RT_DEV void set_cur(Lane &L, const XRay &c) {
    L.cur = c;
    L.inv = Vec3(1.0 / c.d.x, 1.0 / c.d.y, 1.0 / c.d.z);
    L.a_len = c.d.length_sqr();
}
RT_DEV void accept(Lane &L, double t, uint32_t face) {
    L.closest = t;
    L.found = true;
    L.win.t = t; L.win.leaf = L.top; L.win.face = face; L.win.chain = L.ctx;
}
template <int STACK>
RT_DEV void next_entry(Lane &L, Stack<STACK> &st) {
    L.top = st.pop(L);
    L.op = classify(L.top);
}

// BvhNode::hit, bvh/mod.rs:86-101 + AABB::hit, aabb.rs:15-32. Left child is taken at once,
// the right one waits on the stack and is tested against the then-closest hit.
template <int STACK, bool STATS>
RT_DEV void op_node(const SceneDev &s, Lane &L, Stack<STACK> &st, double t_min, Counters<STATS> &cnt) {
    cnt.node();
    const rt_bvh_node &n = s.nodes[RT_REF_INDEX(L.top)];
    double tmn = t_min, tmx = L.closest;
    bool miss = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double inv_d = L.inv[i];
        double t0 = (n.bmin[i] - L.cur.o[i]) * inv_d;
        double t1 = (n.bmax[i] - L.cur.o[i]) * inv_d;
        if (inv_d < 0.0) { double tmp = t0; t0 = t1; t1 = tmp; }
        tmn = t0 > tmn ? t0 : tmn;
        tmx = t1 < tmx ? t1 : tmx;
        miss = miss || (tmx <= tmn);
    }
    if (!miss) {
        st.push(L, n.right);
        L.top = n.left;
        L.op = classify(L.top);
    } else {
        next_entry(L, st);
    }
}

template <int STACK, bool STATS>
RT_DEV void op_sphere(const SceneDev &s, Lane &L, Stack<STACK> &st, double t_min, Counters<STATS> &cnt) {
    uint32_t kind = RT_REF_KIND(L.top), idx = RT_REF_INDEX(L.top);
    cnt.prim(kind);
    Vec3 center;
    double radius;
    if (kind == RT_KIND_SPHERE) { const rt_sphere &q = s.spheres[idx]; center = ld3(q.center); radius = q.radius; }
    else { const rt_moving_sphere &q = s.moving_spheres[idx]; center = moving_center(q, L.r.tm); radius = q.radius; }
    double t;
    if (sphere_t(center, radius, L.cur, L.a_len, t_min, L.closest, t)) accept(L, t, 0);
    next_entry(L, st);
}
template <int STACK, bool STATS>
RT_DEV void op_rect(const SceneDev &s, Lane &L, Stack<STACK> &st, double t_min, Counters<STATS> &cnt) {
    cnt.prim(RT_KIND_RECT);
    const rt_rect &q = s.rects[RT_REF_INDEX(L.top)];
    double t;
    if (rect_t(q.axis, q.a0, q.a1, q.b0, q.b1, q.k, L.cur, t_min, L.closest, t)) accept(L, t, 0);
    next_entry(L, st);
}
template <int STACK, bool STATS>
RT_DEV void op_box(const SceneDev &s, Lane &L, Stack<STACK> &st, double t_min, Counters<STATS> &cnt) {
    cnt.prim(RT_KIND_BOX);
    double t;
    uint32_t face = 0;
    if (box_t(s.boxes[RT_REF_INDEX(L.top)], L.cur, t_min, L.closest, t, face)) accept(L, t, face);
    next_entry(L, st);
}
template <int STACK, bool STATS>
RT_DEV void op_misc(const SceneDev &s, Lane &L, Stack<STACK> &st, double t_min, Counters<STATS> &cnt) {
    uint32_t kind = RT_REF_KIND(L.top), idx = RT_REF_INDEX(L.top);
    cnt.prim(kind);
    double t;
    bool h = kind == RT_KIND_TRIANGLE ? triangle_t(s.triangles[idx], L.cur, t_min, L.closest, t)
                                      : ring_t(s.rings[idx], L.cur, t_min, L.closest, t);
    if (h) accept(L, t, 0);
    next_entry(L, st);
}
// ConstantMedium::hit, constantmedium.rs:49-83.
template <int STACK, bool STATS>
RT_DEV void op_medium(const SceneDev &s, Lane &L, Stack<STACK> &st, double t_min, Counters<STATS> &cnt) {
    cnt.prim(RT_KIND_MEDIUM);
    const rt_medium &m = s.media[RT_REF_INDEX(L.top)];
    double t1, t2;
    if (boundary_t<STATS>(s, m.boundary, L.cur, L.r.tm, -rtm::INF, rtm::INF, t1, cnt) &&
        boundary_t<STATS>(s, m.boundary, L.cur, L.r.tm, t1 + 0.0001, rtm::INF, t2, cnt)) {
        t1 = rtm::fmax_(t1, t_min);
        t2 = rtm::fmin_(t2, L.closest);
        if (!(t1 >= t2)) {
            t1 = rtm::fmax_(t1, 0.0);
            double ray_length = L.cur.d.length();
            double distance_inside_boundary = (t2 - t1) * ray_length;
            double rnd = L.rng.gen_f64();
            double hit_distance = m.neg_inv_density * (rtm::log_(rnd) / rtm::log_(rtm::E_));
            if (!(hit_distance > distance_inside_boundary)) accept(L, t1 + hit_distance / ray_length, 0);
        }
    }
    next_entry(L, st);
}
// Translate / RotateY / Zoom entry and exit; HittableList expansion (mod.rs:90-100).
template <int STACK, bool STATS>
RT_DEV void op_ctx(const SceneDev &s, Lane &L, Stack<STACK> &st, Counters<STATS> &cnt) {
    if (L.top == REF_POPCTX) {
        L.ctx.n--;
        set_cur(L, ray_at_level(s, L.ctx, L.ctx.n, XRay{L.r.orig, L.r.dir}));
        next_entry(L, st);
        return;
    }
    uint32_t kind = RT_REF_KIND(L.top), idx = RT_REF_INDEX(L.top);
    cnt.prim(kind);
    if (kind == RT_KIND_LIST) {
        const rt_list &l = s.lists[idx];
        for (uint32_t i = l.count; i > 0; i--) st.push(L, s.list_items[l.first + i - 1]);
        next_entry(L, st);
        return;
    }
    if (L.ctx.n < RT_MAX_XFORM_DEPTH) {
        L.ctx.push(L.top);
        set_cur(L, xform_ray(s, L.top, L.cur));
        st.push(L, REF_POPCTX);
        L.top = s.xforms[idx].child;
        L.op = classify(L.top);
    } else {
        next_entry(L, st);
    }
}

// One bounce record of the reference's recursion (main.rs:246-271):
//   specular:  L = w * L_next                         stored with p = 1 ((w*L)/1 == w*L)
//   diffuse:   L = emitted + ((att*spdf) * L_next) / pdf_val,   emitted = 0 for every
//              material that scatters (only DiffuseLight emits, and it ends the path)
// kept on a per-lane tape in HBM and unwound innermost-first when the path ends, so the
// f64 result is the recursion's, operation for operation (including its NaN / inf cases).
struct Tape {
    double *base;          // [record k][field f][lane] -> base[(k*4+f)*n_lanes + lane]
    uint64_t n_lanes, lane;
    RT_DEV void put(uint32_t k, Vec3 w, double p) {
        double *q = base + ((uint64_t)k * 4) * n_lanes + lane;
        q[0] = w.x; q[n_lanes] = w.y; q[2 * n_lanes] = w.z; q[3 * n_lanes] = p;
    }
    RT_DEV Vec3 unwind(uint32_t nb, Vec3 Lr) {
        for (uint32_t k = nb; k > 0; k--) {
            const double *q = base + ((uint64_t)(k - 1) * 4) * n_lanes + lane;
            Vec3 w(q[0], q[n_lanes], q[2 * n_lanes]);
            double p = q[3 * n_lanes];
            Lr = Vec3(0.0, 0.0, 0.0) + (w * Lr) / p;
        }
        return Lr;
    }
};

// Shade the finished traversal (main.rs:243-277), then — if the path ended — add it to the
// pixel, move on to the next sample / item (main.rs:144-152) and aim the next camera ray.
template <int STACK, bool STATS>
RT_DEV void op_shade(const SceneDev &s, const RenderArgs &a, Lane &L, Stack<STACK> &st, Tape &tape, unsigned lane,
                     Counters<STATS> &cnt) {
    const Vec3 background = ld3(a.background);
    if (L.alive) {
        bool end_path = false;
        Vec3 Lterm(0.0, 0.0, 0.0);
        if (!L.found) {
            Lterm = background;
            end_path = true;
        } else {
            HitRec rec;
            winner_record(s, L.r, L.win, rec);
            const rt_material &mat = s.materials[rec.mat & kMatIndexMask];
            uint32_t mk = mat.kind;
            if (mk == RT_MAT_DIFFUSE_LIGHT) {                     // emitted; scatter = None (mod.rs:16-18,174-180)
                Lterm = rec.front_face ? texture_value(s, mat.tex, rec.u, rec.v, rec.p) : Vec3(0.0, 0.0, 0.0);
                end_path = true;
            } else {
                Vec3 w;
                double p = 1.0;
                Vec3 dir;
                double tm = L.r.tm;
                if (mk == RT_MAT_LAMBERTIAN) {
                    Vec3 att = texture_value(s, mat.tex, rec.u, rec.v, rec.p);
                    rtm::Onb uvw = rtm::onb_from_w(rec.normal);
                    double cosv;
                    if (s.n_lights == 0) {                        // cosine-only mode
                        dir = uvw.local_vec(random_cosine_direction(L.rng));
                        cosv = rtm::dot(rtm::to_unit(dir), uvw.w);
                        p = cosv <= 0.0 ? 0.0 : cosv / rtm::PI;
                    } else {                                      // MixturePdf(lights, cos), pdf.rs:94-104
                        if (L.rng.gen_range(0.0, 1.0) < 0.5) dir = lights_random(s, rec.p, L.rng);
                        else dir = uvw.local_vec(random_cosine_direction(L.rng));
                        double lp = lights_pdf_value<STATS>(s, rec.p, dir, cnt);
                        cosv = rtm::dot(rtm::to_unit(dir), uvw.w);
                        double cp = cosv <= 0.0 ? 0.0 : cosv / rtm::PI;
                        p = 0.5 * lp + 0.5 * cp;
                    }
                    double cosine = rtm::dot(rec.normal, rtm::to_unit(dir));
                    double spdf = cosine < 0.0 ? 0.0 : cosine / rtm::PI;
                    w = att * spdf;
                } else if (mk == RT_MAT_METAL) {                  // mod.rs:85-96
                    Vec3 reflected = rtm::reflect(rtm::to_unit(L.r.dir), rec.normal);
                    dir = reflected + random_in_unit_sphere(L.rng) * mat.param;
                    w = ld3(mat.albedo);
                    tm = 0.0;                                     // time = 0., mod.rs:91
                } else if (mk == RT_MAT_DIELECTRIC) {             // mod.rs:120-147
                    double refraction_ratio = rec.front_face ? 1.0 / mat.param : mat.param;
                    Vec3 unit_direction = rtm::to_unit(L.r.dir);
                    double cos_theta = rtm::fmin_(rtm::dot(-unit_direction, rec.normal), 1.0);
                    double sin_theta = rtm::sqrt_(1.0 - cos_theta * cos_theta);
                    bool cannot_refract = refraction_ratio * sin_theta > 1.0;
                    double random_double = L.rng.gen_range(0.0, 1.0);
                    dir = (cannot_refract || reflectance(cos_theta, refraction_ratio) > random_double)
                              ? rtm::reflect(unit_direction, rec.normal)
                              : rtm::refract(unit_direction, rec.normal, refraction_ratio);
                    w = Vec3(1.0, 1.0, 1.0);
                } else {                                          // Isotropic, mod.rs:207-213
                    w = texture_value(s, mat.tex, rec.u, rec.v, rec.p);
                    dir = random_in_unit_sphere(L.rng);
                }
                tape.put(L.nb, w, p);
                L.nb++;
                L.r = Ray(rec.p, dir, tm);
                L.depth--;
                if (L.depth <= 0) end_path = true;                // the next ray_color returns (0,0,0), main.rs:240-242
            }
        }
        if (end_path) {
            L.pixel_sum = L.pixel_sum + tape.unwind(L.nb, Lterm);
            L.alive = false;
            cnt.draws(L.rng.draws);
        }
    }

    if (!L.alive) {
        // Next sample of the item, or the next item (one atomic per wave for all idle lanes).
        bool need = !L.have_item || L.smp == L.smp_end;
        if (need && L.have_item) {
            double *o = a.partial + ((uint64_t)L.chunk_id * a.n_pixels + L.slot) * 3;
            o[0] = L.pixel_sum.x; o[1] = L.pixel_sum.y; o[2] = L.pixel_sum.z;
            L.have_item = false;
            if (STATS) cnt.flush(a.stats);
        }
        unsigned long long m = __ballot(need);
        if (m) {
            int leader = __ffsll((long long)m) - 1;
            unsigned long long base = 0;
            if ((int)lane == leader) base = atomicAdd(a.work_counter, (unsigned long long)__popcll(m));
            base = __shfl(base, leader);
            if (need) {
                unsigned long long item = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                if (item < a.n_items) {
                    L.slot = item / a.n_chunks;
                    L.chunk_id = (uint32_t)(item - L.slot * a.n_chunks);
                    uint64_t yi = L.slot / a.width;
                    L.px = (uint32_t)(L.slot - yi * a.width);
                    uint32_t g = a.row_ids[yi];
                    L.frame = g / a.height;
                    L.py = g - L.frame * a.height;
                    L.smp = L.chunk_id * a.chunk;
                    L.smp_end = L.smp + a.chunk < a.spp ? L.smp + a.chunk : a.spp;
                    L.pixel_sum = Vec3(0.0, 0.0, 0.0);
                    L.have_item = true;
                }
            }
        }
        if (!L.have_item) { L.op = OP_IDLE; return; }
        if (L.smp == L.smp_end) return;                           // empty chunk (spp == 0): stored on the next turn
        uint64_t pixel = (uint64_t)L.py * a.width + L.px;
        L.rng = Rng(rtm::path_key(a.seed, L.frame, pixel, L.smp));
        double rand_u = L.rng.gen_f64();
        double rand_v = L.rng.gen_f64();
        double u = ((double)L.px + rand_u) / (double)(a.width - 1);
        double v = ((double)L.py + rand_v) / (double)(a.height - 1);
        L.r = get_ray(a.cam, u, v, L.rng);
        L.depth = (int)a.max_depth;
        L.nb = 0;
        L.smp++;
        cnt.path();
        if (L.depth <= 0) {                                       // MAX_DEPTH == 0: ray_color returns black at once
            cnt.draws(L.rng.draws);
            return;                                               // alive stays false: next sample on the next turn
        }
        L.alive = true;
    }

    // world.hit(r, 0.001, f64::MAX), main.rs:243
    cnt.ray();
    set_cur(L, XRay{L.r.orig, L.r.dir});
    L.closest = rtm::F64_MAX;
    L.found = false;
    L.ctx.n = 0;
    L.sp = 0;
    L.top = s.root;
    L.op = classify(L.top);
}

} // namespace

template <int STACK, bool STATS>
__global__ void __launch_bounds__(kBlock) pt_megakernel(const SceneDev s, const RenderArgs a) {
    __shared__ uint32_t stack_lds[STACK * kBlock];
    Stack<STACK> st{stack_lds + threadIdx.x};
    const unsigned lane = threadIdx.x & 63u;
    Counters<STATS> cnt;
    Tape tape{a.tape, (uint64_t)gridDim.x * kBlock, (uint64_t)blockIdx.x * kBlock + threadIdx.x};

    Lane L;
    L.have_item = false; L.alive = false; L.found = false;
    L.slot = 0; L.chunk_id = 0; L.smp = 0; L.smp_end = 0; L.px = 0; L.py = 0; L.frame = 0;
    L.depth = 0; L.nb = 0; L.sp = 0; L.top = REF_EMPTY; L.op = OP_SHADE;
    L.closest = rtm::F64_MAX; L.a_len = 0.0;
    L.ctx.c0 = L.ctx.c1 = L.ctx.c2 = L.ctx.c3 = 0; L.ctx.n = 0;
    L.win.t = 0.0; L.win.leaf = 0; L.win.face = 0; L.win.chain = L.ctx;
    const double t_min = a.t_min;
    const int node_quorum = (int)(a.node_quorum & 0xFFu);          // (the upper bits are wavefront-engine tuning)

    for (;;) {
        // Fast path: keep stepping nodes while enough lanes want to.
        for (;;) {
            bool isn = L.op == OP_NODE;
            int nn = __popcll(__ballot(isn));
            if (nn < node_quorum) break;
            if (isn) op_node<STACK, STATS>(s, L, st, t_min, cnt);
        }
        // Vote: the label most lanes are waiting on (ties -> lowest id).
        int best = -1, best_n = 0;
#pragma unroll
        for (int o = 0; o < (int)OP_COUNT; o++) {
            int n = __popcll(__ballot(L.op == (uint32_t)o));
            if (n > best_n) { best_n = n; best = o; }
        }
        if (best < 0) break;                                      // every lane idle
        if (L.op == (uint32_t)best) {
            switch (best) {
                case OP_NODE: op_node<STACK, STATS>(s, L, st, t_min, cnt); break;
                case OP_SPHERE: op_sphere<STACK, STATS>(s, L, st, t_min, cnt); break;
                case OP_RECT: op_rect<STACK, STATS>(s, L, st, t_min, cnt); break;
                case OP_BOX: op_box<STACK, STATS>(s, L, st, t_min, cnt); break;
                case OP_MEDIUM: op_medium<STACK, STATS>(s, L, st, t_min, cnt); break;
                case OP_MISC: op_misc<STACK, STATS>(s, L, st, t_min, cnt); break;
                case OP_CTX: op_ctx<STACK, STATS>(s, L, st, cnt); break;
                default: op_shade<STACK, STATS>(s, a, L, st, tape, lane, cnt); break;
            }
        }
    }
    if (STATS) cnt.flush(a.stats);
}

// out[i] = sum over chunks, in chunk order, of partial[c][i] (i over n_values doubles).
__global__ void __launch_bounds__(256) chunk_sum_kernel(const double *partial, double *out, uint64_t n_values, uint32_t n_chunks) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_values; i += stride) {
        double acc = 0.0;
        for (uint32_t c = 0; c < n_chunks; c++) acc += partial[(uint64_t)c * n_values + i];
        out[i] = acc;
    }
}

// write_color, main.rs:280-299.
__global__ void __launch_bounds__(256) tonemap_kernel(const double *rgb_sum, uint64_t n, int32_t spp, uint8_t *rgb8) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        double c = rgb_sum[i];
        if (c != c) c = 0.0;
        double v = rtm::floor_(rtm::clamp_(rtm::sqrt_(c / (double)spp), 0.0, 0.999) * 255.999);
        rgb8[i] = (uint8_t)v;
    }
}

// rt_math.h on the device, element-wise (parity probe for tests).
__global__ void __launch_bounds__(256) math_probe_kernel(int op, const double *a, const double *b, double *out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b ? b[i] : 0.0, r;
    switch (op) {
        case 0: r = rtm::sin_(x); break;
        case 1: r = rtm::cos_(x); break;
        case 2: r = rtm::acos_(x); break;
        case 3: r = rtm::atan2_(x, y); break;
        case 4: r = rtm::log_(x); break;
        case 5: r = rtm::sqrt_(x); break;
        case 6: r = x / y; break;
        default: r = 0.0;
    }
    out[i] = r;
}
// RNG stream on the device: mode 0 = next_u64, 1 = gen_f64 bits, 2 = gen_range bits, 3 = gen_index.
__global__ void rng_probe_kernel(uint64_t state, int mode, double lo, double hi, uint64_t bound, uint64_t *out, uint64_t n) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng r(state);
    for (uint64_t i = 0; i < n; i++) {
        if (mode == 0) out[i] = r.next_u64();
        else if (mode == 1) out[i] = rtm::d2u(r.gen_f64());
        else if (mode == 2) out[i] = rtm::d2u(r.gen_range(lo, hi));
        else out[i] = r.gen_index(bound);
    }
}

// ---- launchers ------------------------------------------------------------------------
template <int STACK, bool STATS>
static int blocks_for() {
    int per_cu = 0;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_megakernel<STACK, STATS>, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    return per_cu * prop.multiProcessorCount;
}

int render_grid_blocks(uint32_t stack_need, bool counters) {
    if (stack_need <= (uint32_t)kStackSmall) return counters ? blocks_for<kStackSmall, true>() : blocks_for<kStackSmall, false>();
    return counters ? blocks_for<kStackLarge, true>() : blocks_for<kStackLarge, false>();
}

hipError_t launch_render(const SceneDev &scene, const RenderArgs &args, uint32_t stack_need, bool counters,
                         int n_blocks_hint, hipStream_t stream) {
    if (stack_need > (uint32_t)kStackLarge) return hipErrorInvalidValue;
    uint64_t want = (args.n_items + kBlock - 1) / kBlock;
    int blocks = n_blocks_hint > 0 ? n_blocks_hint : render_grid_blocks(stack_need, counters);
    if ((uint64_t)blocks > want) blocks = (int)(want ? want : 1);
    dim3 grid((unsigned)blocks), block(kBlock);
    if (stack_need <= (uint32_t)kStackSmall) {
        if (counters) hipLaunchKernelGGL((pt_megakernel<kStackSmall, true>), grid, block, 0, stream, scene, args);
        else hipLaunchKernelGGL((pt_megakernel<kStackSmall, false>), grid, block, 0, stream, scene, args);
    } else {
        if (counters) hipLaunchKernelGGL((pt_megakernel<kStackLarge, true>), grid, block, 0, stream, scene, args);
        else hipLaunchKernelGGL((pt_megakernel<kStackLarge, false>), grid, block, 0, stream, scene, args);
    }
    return hipGetLastError();
}

hipError_t launch_chunk_sum(const double *partial, double *out, uint64_t n_values, uint32_t n_chunks, hipStream_t stream) {
    uint64_t want = (n_values + 255) / 256;
    unsigned blocks = (unsigned)(want > 2048 ? 2048 : (want ? want : 1));
    hipLaunchKernelGGL(chunk_sum_kernel, dim3(blocks), dim3(256), 0, stream, partial, out, n_values, n_chunks);
    return hipGetLastError();
}
hipError_t launch_tonemap(const double *rgb_sum, uint64_t n_pixels, int32_t spp, uint8_t *rgb8, hipStream_t stream) {
    uint64_t n = n_pixels * 3;
    uint64_t want = (n + 255) / 256;
    unsigned blocks = (unsigned)(want > 2048 ? 2048 : (want ? want : 1));
    hipLaunchKernelGGL(tonemap_kernel, dim3(blocks), dim3(256), 0, stream, rgb_sum, n, spp, rgb8);
    return hipGetLastError();
}
hipError_t launch_math_probe(int op, const double *a, const double *b, double *out, uint64_t n, hipStream_t stream) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(math_probe_kernel, dim3(blocks), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError();
}
hipError_t launch_rng_probe(uint64_t state, int mode, double lo, double hi, uint64_t bound, uint64_t *out, uint64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(rng_probe_kernel, dim3(1), dim3(64), 0, stream, state, mode, lo, hi, bound, out, n);
    return hipGetLastError();
}

} // namespace rt2022

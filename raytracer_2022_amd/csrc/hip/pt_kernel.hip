// pt_kernel.hip — the path-tracing megakernel for gfx950 (MI355X, CDNA4).
//
// One persistent launch renders a whole batch of rows. Every lane of every
// 64-wide wavefront owns one path at a time and runs the reference's recursion
// (raytracer/src/main.rs:233-278) as an iterative bounce loop; when a path ends the
// lane moves to the next sample of its work item (pixel x sample-chunk), and when
// the item ends the wave refills the idle lanes from a global work counter with one
// atomic per wave (__ballot / __popcll / __shfl) — so lanes never sit dead while
// work remains. BVH traversal uses an explicit per-lane stack in LDS
// ([depth][lane] -> bank = lane, conflict-free), visits nodes in exactly the
// reference's order (left subtree, then right with t_max = closest so far,
// hittable/bvh/mod.rs:86-101) because ConstantMedium::hit draws from the RNG
// during traversal (constantmedium.rs:60), and defers the hit record (normal, uv)
// to the single winning candidate. All arithmetic is f64 through rt_math.h, so the
// paths are bit-identical to the CPU oracle's.
//
// No MFMA: this is branchy scalar f64, not a contraction.
#include <hip/hip_runtime.h>

#include "../rt_math.h"
#include "pt_device.h"

namespace rt2022 {

using rtm::Ray;
using rtm::Rng;
using rtm::Vec3;

#define RT_DEV __device__ __forceinline__

namespace {

constexpr uint32_t REF_POPCTX = 0xFFFFFFFFu;   // stack sentinel: leave the innermost mover

RT_DEV Vec3 ld3(const double *p) { return Vec3(p[0], p[1], p[2]); }

struct XRay { Vec3 o, d; };                    // ray without its time (movers never change tm)

template <bool STATS>
struct Counters;
template <>
struct Counters<false> {
    RT_DEV void path() {}
    RT_DEV void ray() {}
    RT_DEV void node() {}
    RT_DEV void prim(uint32_t) {}
    RT_DEV void light_pdf() {}
    RT_DEV void draws(uint32_t) {}
    RT_DEV void flush(StatsDev *) {}
};
template <>
struct Counters<true> {
    uint32_t paths = 0, rays = 0, nodes = 0, lpdf = 0, ndraws = 0;
    uint32_t prims[RT_KIND_COUNT] = {};
    RT_DEV void path() { paths++; }
    RT_DEV void ray() { rays++; }
    RT_DEV void node() { nodes++; }
    RT_DEV void prim(uint32_t k) {
#pragma unroll
        for (int i = 0; i < RT_KIND_COUNT; i++) prims[i] += (k == (uint32_t)i) ? 1u : 0u;
    }
    RT_DEV void light_pdf() { lpdf++; }
    RT_DEV void draws(uint32_t n) { ndraws += n; }
    RT_DEV void flush(StatsDev *st) {
        if (!st) return;
        if (paths) atomicAdd(&st->paths, (unsigned long long)paths);
        if (rays) atomicAdd(&st->rays, (unsigned long long)rays);
        if (nodes) atomicAdd(&st->node_visits, (unsigned long long)nodes);
        if (lpdf) atomicAdd(&st->light_pdf_tests, (unsigned long long)lpdf);
        if (ndraws) atomicAdd(&st->rng_draws, (unsigned long long)ndraws);
#pragma unroll
        for (int i = 0; i < RT_KIND_COUNT; i++)
            if (prims[i]) atomicAdd(&st->prim_tests[i], (unsigned long long)prims[i]);
        paths = rays = nodes = lpdf = ndraws = 0;
#pragma unroll
        for (int i = 0; i < RT_KIND_COUNT; i++) prims[i] = 0;
    }
};

// ---- movers: Translate / RotateY / Zoom (hittable/mod.rs:165-174,235-264,321-330) ----
RT_DEV XRay xform_ray(const SceneDev &s, uint32_t ref, XRay r) {
    const rt_xform &x = s.xforms[RT_REF_INDEX(ref)];
    uint32_t kind = RT_REF_KIND(ref);
    if (kind == RT_KIND_TRANSLATE) {
        r.o = r.o - ld3(x.p);
    } else if (kind == RT_KIND_ROTATE_Y) {
        double sin_theta = x.p[0], cos_theta = x.p[1];
        double ox = cos_theta * r.o.x - sin_theta * r.o.z;
        double oz = sin_theta * r.o.x + cos_theta * r.o.z;
        double dx = cos_theta * r.d.x - sin_theta * r.d.z;
        double dz = sin_theta * r.d.x + cos_theta * r.d.z;
        r.o.x = ox; r.o.z = oz; r.d.x = dx; r.d.z = dz;
    } else {
        r.o = r.o / x.p[0];
    }
    return r;
}

struct HitRec {                                // HitRecord, hittable/mod.rs:18-26
    Vec3 p, normal;
    double t, u, v;
    bool front_face;
    uint32_t mat;
    RT_DEV void set_face_normal(Vec3 dir, Vec3 outward_normal) {      // mod.rs:49-56
        front_face = rtm::dot(dir, outward_normal) < 0.0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

// The record coming back up through one mover (`moved` = the ray inside it).
RT_DEV void xform_record(const SceneDev &s, uint32_t ref, const XRay &moved, HitRec &rec) {
    const rt_xform &x = s.xforms[RT_REF_INDEX(ref)];
    uint32_t kind = RT_REF_KIND(ref);
    if (kind == RT_KIND_TRANSLATE) {
        rec.p = rec.p + ld3(x.p);
        rec.set_face_normal(moved.d, rec.normal);
    } else if (kind == RT_KIND_ROTATE_Y) {
        double sin_theta = x.p[0], cos_theta = x.p[1];
        Vec3 p = rec.p, n = rec.normal;
        p.x = cos_theta * rec.p.x + sin_theta * rec.p.z;
        p.z = -sin_theta * rec.p.x + cos_theta * rec.p.z;
        n.x = cos_theta * rec.normal.x + sin_theta * rec.normal.z;
        n.z = -sin_theta * rec.normal.x + cos_theta * rec.normal.z;
        rec.p = p;
        rec.set_face_normal(moved.d, n);
    } else {
        rec.p = rec.p * x.p[0];
        rec.set_face_normal(moved.d, rec.normal);
    }
    if (ref & RT_REF_FLIP) rec.front_face = !rec.front_face;
}

// Chain of enclosing movers, outermost first.
struct Chain {
    uint32_t c0, c1, c2, c3;
    uint32_t n;
    RT_DEV uint32_t at(uint32_t i) const { return i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : c3; }
    RT_DEV void push(uint32_t ref) {
        if (n == 0) c0 = ref; else if (n == 1) c1 = ref; else if (n == 2) c2 = ref; else c3 = ref;
        n++;
    }
};
RT_DEV XRay ray_at_level(const SceneDev &s, const Chain &ch, uint32_t level, XRay world) {
    XRay r = world;
    for (uint32_t i = 0; i < level && i < RT_MAX_XFORM_DEPTH; i++) r = xform_ray(s, ch.at(i), r);
    return r;
}

// ---- primitive tests (t only; the record is rebuilt for the winner) -----------------
// Sphere::hit roots, sphere.rs:39-58.
RT_DEV bool sphere_t(Vec3 center, double radius, const XRay &r, double a, double t_min, double t_max, double &t) {
    Vec3 oc = r.o - center;
    double half_b = rtm::dot(oc, r.d);
    double c = oc.length_sqr() - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = rtm::sqrt_(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    t = root;
    return true;
}
RT_DEV Vec3 moving_center(const rt_moving_sphere &q, double time) {   // sphere.rs:124-127
    Vec3 c0 = ld3(q.center0), c1 = ld3(q.center1);
    return c0 + (c1 - c0) * ((time - q.time0) / (q.time1 - q.time0));
}
// X?Rect::hit, aarect.rs:46-56 (and the XZ / YZ twins).
RT_DEV bool rect_t(uint32_t axis, double a0, double a1, double b0, double b1, double k, const XRay &r,
                   double t_min, double t_max, double &t) {
    double ok, dk, oa, da, ob, db;
    if (axis == RT_RECT_XY) { ok = r.o.z; dk = r.d.z; oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; }
    else if (axis == RT_RECT_XZ) { ok = r.o.y; dk = r.d.y; oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; }
    else { ok = r.o.x; dk = r.d.x; oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; }
    double tt = (k - ok) / dk;
    if (tt < t_min || tt > t_max) return false;
    double a = oa + tt * da;
    double b = ob + tt * db;
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    t = tt;
    return true;
}
struct RectP { uint32_t axis; double a0, a1, b0, b1, k; };
RT_DEV RectP box_side(const rt_box &bx, int i) {                       // boxes.rs:24-66
    const double *p0 = bx.p0, *p1 = bx.p1;
    RectP q;
    if (i < 2) { q.axis = RT_RECT_XY; q.a0 = p0[0]; q.a1 = p1[0]; q.b0 = p0[1]; q.b1 = p1[1]; q.k = (i == 0) ? p1[2] : p0[2]; }
    else if (i < 4) { q.axis = RT_RECT_XZ; q.a0 = p0[0]; q.a1 = p1[0]; q.b0 = p0[2]; q.b1 = p1[2]; q.k = (i == 2) ? p1[1] : p0[1]; }
    else { q.axis = RT_RECT_YZ; q.a0 = p0[1]; q.a1 = p1[1]; q.b0 = p0[2]; q.b1 = p1[2]; q.k = (i == 4) ? p1[0] : p0[0]; }
    return q;
}
// Boxes::hit = HittableList::hit over the six sides (boxes.rs:80-82, mod.rs:90-100).
RT_DEV bool box_t(const rt_box &bx, const XRay &r, double t_min, double t_max, double &t, uint32_t &face) {
    bool any = false;
    double closest = t_max;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        RectP q = box_side(bx, i);
        double tt;
        if (rect_t(q.axis, q.a0, q.a1, q.b0, q.b1, q.k, r, t_min, closest, tt)) { closest = tt; face = (uint32_t)i; any = true; }
    }
    t = closest;
    return any;
}
// Triangle::hit, triangle.rs:51-63.
RT_DEV bool triangle_t(const rt_triangle &tr, const XRay &r, double t_min, double t_max, double &t) {
    Vec3 a = ld3(tr.a), b = ld3(tr.b), c = ld3(tr.c);
    Vec3 n = rtm::to_unit(rtm::cross(b - a, c - a));
    double tt = rtm::dot(r.d, n);
    tt = rtm::dot(a - r.o, n) / tt;
    if (tt != tt || tt < t_min || tt > t_max) return false;
    Vec3 p = r.o + r.d * tt;
    bool inside = rtm::dot(rtm::cross(c - a, p - a), rtm::cross(c - a, b - a)) >= 0.0 &&
                  rtm::dot(rtm::cross(a - b, p - b), rtm::cross(a - b, c - b)) >= 0.0 &&
                  rtm::dot(rtm::cross(b - c, p - c), rtm::cross(b - c, a - c)) >= 0.0;
    if (!inside) return false;
    t = tt;
    return true;
}
// Ring::hit, ring.rs:36-47.
RT_DEV bool ring_t(const rt_ring &g, const XRay &r, double t_min, double t_max, double &t) {
    double tt = -r.o.y / r.d.y;
    if (tt != tt || tt < t_min || tt > t_max) return false;
    Vec3 p = r.o + r.d * tt;
    double d = p.x * p.x + p.z * p.z;
    if (d < g.dis_min || d > g.dis_max) return false;
    t = tt;
    return true;
}

// dyn Hittable::hit for the leaf kinds SPHERE..RING.
RT_DEV bool prim_t(const SceneDev &s, uint32_t kind, uint32_t idx, const XRay &r, double tm,
                   double t_min, double t_max, double &t, uint32_t &face) {
    switch (kind) {
        case RT_KIND_SPHERE: {
            const rt_sphere &q = s.spheres[idx];
            return sphere_t(ld3(q.center), q.radius, r, r.d.length_sqr(), t_min, t_max, t);
        }
        case RT_KIND_MOVING_SPHERE: {
            const rt_moving_sphere &q = s.moving_spheres[idx];
            return sphere_t(moving_center(q, tm), q.radius, r, r.d.length_sqr(), t_min, t_max, t);
        }
        case RT_KIND_RECT: {
            const rt_rect &q = s.rects[idx];
            return rect_t(q.axis, q.a0, q.a1, q.b0, q.b1, q.k, r, t_min, t_max, t);
        }
        case RT_KIND_BOX: return box_t(s.boxes[idx], r, t_min, t_max, t, face);
        case RT_KIND_TRIANGLE: return triangle_t(s.triangles[idx], r, t_min, t_max, t);
        case RT_KIND_RING: return ring_t(s.rings[idx], r, t_min, t_max, t);
        default: return false;
    }
}

// boundary.hit(r, t_min, t_max).t for a medium boundary: movers around one primitive.
template <bool STATS>
RT_DEV bool boundary_t(const SceneDev &s, uint32_t ref, XRay r, double tm, double t_min, double t_max, double &t,
                       Counters<STATS> &cnt) {
    for (int lvl = 0; lvl <= RT_MAX_XFORM_DEPTH; lvl++) {
        uint32_t kind = RT_REF_KIND(ref);
        cnt.prim(kind);
        if (kind >= RT_KIND_TRANSLATE && kind <= RT_KIND_ZOOM) {
            r = xform_ray(s, ref, r);
            ref = s.xforms[RT_REF_INDEX(ref)].child;
            continue;
        }
        uint32_t face;
        return prim_t(s, kind, RT_REF_INDEX(ref), r, tm, t_min, t_max, t, face);
    }
    return false;
}

// ---- closest hit over the world: BvhNode::hit & friends as an explicit-stack loop ---
struct Winner {
    double t;
    uint32_t leaf;      // ref of the winning leaf (with its flip bit)
    uint32_t face;      // box side
    Chain chain;        // movers enclosing it
};

template <int STACK, bool STATS>
RT_DEV bool trace(const SceneDev &s, uint32_t *stk, const Ray &wr, double t_min, Rng &rng, Winner &win,
                  Counters<STATS> &cnt) {
    const XRay world{wr.orig, wr.dir};
    XRay cur = world;
    Vec3 inv(1.0 / cur.d.x, 1.0 / cur.d.y, 1.0 / cur.d.z);     // aabb.rs:19, hoisted (same value at every node)
    double a_len = cur.d.length_sqr();                          // sphere.rs:41, hoisted likewise
    double closest = rtm::F64_MAX;                              // main.rs:243
    bool found = false;
    Chain ctx;
    ctx.c0 = ctx.c1 = ctx.c2 = ctx.c3 = 0; ctx.n = 0;
    int sp = 0;
    stk[0] = s.root; sp = 1;
    while (sp > 0) {
        sp--;
        uint32_t ref = stk[sp * kBlock];
        if (ref == REF_POPCTX) {
            ctx.n--;
            cur = ray_at_level(s, ctx, ctx.n, world);
            inv = Vec3(1.0 / cur.d.x, 1.0 / cur.d.y, 1.0 / cur.d.z);
            a_len = cur.d.length_sqr();
            continue;
        }
        uint32_t kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
        if (kind == RT_KIND_NODE) {
            cnt.node();
            const rt_bvh_node &n = s.nodes[idx];
            // AABB::hit, aabb.rs:15-32 (all three slabs; the early return only skips work).
            double tmn = t_min, tmx = closest;
            bool miss = false;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                double inv_d = inv[i];
                double t0 = (n.bmin[i] - cur.o[i]) * inv_d;
                double t1 = (n.bmax[i] - cur.o[i]) * inv_d;
                if (inv_d < 0.0) { double tmp = t0; t0 = t1; t1 = tmp; }
                tmn = t0 > tmn ? t0 : tmn;
                tmx = t1 < tmx ? t1 : tmx;
                miss = miss || (tmx <= tmn);
            }
            if (!miss && sp + 2 <= STACK) {
                stk[sp * kBlock] = n.right; sp++;
                stk[sp * kBlock] = n.left; sp++;
            }
            continue;
        }
        cnt.prim(kind);
        if (kind >= RT_KIND_TRANSLATE && kind <= RT_KIND_ZOOM) {
            if (ctx.n < RT_MAX_XFORM_DEPTH && sp + 2 <= STACK) {
                ctx.push(ref);
                cur = xform_ray(s, ref, cur);
                inv = Vec3(1.0 / cur.d.x, 1.0 / cur.d.y, 1.0 / cur.d.z);
                a_len = cur.d.length_sqr();
                stk[sp * kBlock] = REF_POPCTX; sp++;
                stk[sp * kBlock] = s.xforms[idx].child; sp++;
            }
            continue;
        }
        if (kind == RT_KIND_LIST) {                             // HittableList::hit, mod.rs:90-100
            const rt_list &l = s.lists[idx];
            if (sp + (int)l.count <= STACK)
                for (uint32_t i = l.count; i > 0; i--) { stk[sp * kBlock] = s.list_items[l.first + i - 1]; sp++; }
            continue;
        }
        double t;
        uint32_t face = 0;
        bool h;
        if (kind == RT_KIND_MEDIUM) {                           // ConstantMedium::hit, constantmedium.rs:49-83
            const rt_medium &m = s.media[idx];
            double t1, t2;
            h = false;
            if (boundary_t<STATS>(s, m.boundary, cur, wr.tm, -rtm::INF, rtm::INF, t1, cnt) &&
                boundary_t<STATS>(s, m.boundary, cur, wr.tm, t1 + 0.0001, rtm::INF, t2, cnt)) {
                t1 = rtm::fmax_(t1, t_min);
                t2 = rtm::fmin_(t2, closest);
                if (!(t1 >= t2)) {
                    t1 = rtm::fmax_(t1, 0.0);
                    double ray_length = cur.d.length();
                    double distance_inside_boundary = (t2 - t1) * ray_length;
                    double rnd = rng.gen_f64();
                    double hit_distance = m.neg_inv_density * (rtm::log_(rnd) / rtm::log_(rtm::E_));
                    if (!(hit_distance > distance_inside_boundary)) { t = t1 + hit_distance / ray_length; h = true; }
                }
            }
        } else if (kind == RT_KIND_SPHERE) {
            const rt_sphere &q = s.spheres[idx];
            h = sphere_t(ld3(q.center), q.radius, cur, a_len, t_min, closest, t);
        } else if (kind == RT_KIND_MOVING_SPHERE) {
            const rt_moving_sphere &q = s.moving_spheres[idx];
            h = sphere_t(moving_center(q, wr.tm), q.radius, cur, a_len, t_min, closest, t);
        } else {
            h = prim_t(s, kind, idx, cur, wr.tm, t_min, closest, t, face);
        }
        if (h) {
            closest = t;
            found = true;
            win.t = t; win.leaf = ref; win.face = face; win.chain = ctx;
        }
    }
    return found;
}

// get_sphere_uv, sphere.rs:30-34.
RT_DEV void sphere_uv(Vec3 p, double &u, double &v) {
    double theta = rtm::acos_(-p.y);
    double phi = rtm::atan2_(-p.z, p.x) + rtm::PI;
    u = phi / (2.0 * rtm::PI);
    v = theta / rtm::PI;
}

RT_DEV void rect_record(const RectP &q, uint32_t mat, const XRay &r, double t, HitRec &rec) {   // aarect.rs:51-71
    double oa, da, ob, db;
    Vec3 outward_normal;
    if (q.axis == RT_RECT_XY) { oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; outward_normal = Vec3(0.0, 0.0, 1.0); }
    else if (q.axis == RT_RECT_XZ) { oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; outward_normal = Vec3(0.0, 1.0, 0.0); }
    else { oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; outward_normal = Vec3(1.0, 0.0, 0.0); }
    double a = oa + t * da;
    double b = ob + t * db;
    rec.p = r.o + r.d * t;
    rec.t = t;
    rec.u = (a - q.a0) / (q.a1 - q.a0);
    rec.v = (b - q.b0) / (q.b1 - q.b0);
    rec.mat = mat;
    rec.set_face_normal(r.d, outward_normal);
}

// The HitRecord of the winning candidate, rebuilt from (leaf, t) in the leaf's own
// frame and then carried out through its movers.
RT_DEV void winner_record(const SceneDev &s, const Ray &wr, const Winner &w, HitRec &rec) {
    const XRay world{wr.orig, wr.dir};
    XRay r = ray_at_level(s, w.chain, w.chain.n, world);
    uint32_t kind = RT_REF_KIND(w.leaf), idx = RT_REF_INDEX(w.leaf);
    double t = w.t;
    switch (kind) {
        case RT_KIND_SPHERE: case RT_KIND_MOVING_SPHERE: {       // sphere.rs:59-65,158-164
            Vec3 center; double radius; uint32_t mat;
            if (kind == RT_KIND_SPHERE) { const rt_sphere &q = s.spheres[idx]; center = ld3(q.center); radius = q.radius; mat = q.mat; }
            else { const rt_moving_sphere &q = s.moving_spheres[idx]; center = moving_center(q, wr.tm); radius = q.radius; mat = q.mat; }
            Vec3 at = r.o + r.d * t;
            Vec3 outward_normal = (at - center) / radius;
            sphere_uv(outward_normal, rec.u, rec.v);
            rec.p = at; rec.t = t; rec.mat = mat;
            rec.set_face_normal(r.d, outward_normal);
            break;
        }
        case RT_KIND_RECT: {
            const rt_rect &q = s.rects[idx];
            RectP rp{q.axis, q.a0, q.a1, q.b0, q.b1, q.k};
            rect_record(rp, q.mat, r, t, rec);
            break;
        }
        case RT_KIND_BOX: {
            const rt_box &bx = s.boxes[idx];
            rect_record(box_side(bx, (int)w.face), bx.mat, r, t, rec);
            break;
        }
        case RT_KIND_TRIANGLE: {                                 // triangle.rs:54-76
            const rt_triangle &tr = s.triangles[idx];
            Vec3 a = ld3(tr.a), b = ld3(tr.b), c = ld3(tr.c);
            Vec3 n = rtm::to_unit(rtm::cross(b - a, c - a));
            Vec3 p = r.o + r.d * t;
            double a1 = a.x - b.x, b1 = a.x - c.x, c1 = a.x - p.x;
            double a2 = a.y - b.y, b2 = a.y - c.y, c2 = a.y - p.y;
            rec.u = (c1 * b2 - b1 * c2) / (a1 * b2 - b1 * a2);
            rec.v = (a1 * c2 - a2 * c1) / (a1 * b2 - b1 * a2);
            rec.p = p; rec.t = t; rec.mat = tr.mat;
            rec.set_face_normal(r.d, n);
            break;
        }
        case RT_KIND_RING: {                                     // ring.rs:49-52
            rec.p = r.o + r.d * t; rec.t = t; rec.u = 0.0; rec.v = 0.0; rec.mat = s.rings[idx].mat;
            rec.set_face_normal(r.d, Vec3(0.0, 1.0, 0.0));
            break;
        }
        default: {                                               // medium, constantmedium.rs:66-74
            rec.p = r.o + r.d * t; rec.normal = Vec3(1.0, 0.0, 0.0); rec.t = t; rec.u = 0.0; rec.v = 0.0;
            rec.front_face = true; rec.mat = s.media[idx].mat;
            break;
        }
    }
    if (w.leaf & RT_REF_FLIP) rec.front_face = !rec.front_face; // FlipFace::hit, mod.rs:281-288
    for (uint32_t lvl = w.chain.n; lvl > 0; lvl--) {
        XRay moved = ray_at_level(s, w.chain, lvl, world);
        xform_record(s, w.chain.at(lvl - 1), moved, rec);
    }
}

// ---- textures, texture/mod.rs:25-139, texture/perlin.rs:52-112 ----------------------
RT_DEV double perlin_noise(const rt_perlin &pl, Vec3 p) {
    double fx = rtm::floor_(p.x), fy = rtm::floor_(p.y), fz = rtm::floor_(p.z);
    double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    u = u * u * (3.0 - 2.0 * u);
    v = v * v * (3.0 - 2.0 * v);
    w = w * w * (3.0 - 2.0 * w);
    int32_t i = rtm::f64_as_i32(fx), j = rtm::f64_as_i32(fy), k = rtm::f64_as_i32(fz);
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int d = 0; d < 2; d++) {
                int32_t ii = (int32_t)((uint32_t)i + (uint32_t)a), jj = (int32_t)((uint32_t)j + (uint32_t)b), kk = (int32_t)((uint32_t)k + (uint32_t)d);
                int32_t id = pl.perm_x[ii & 255] ^ pl.perm_y[jj & 255] ^ pl.perm_z[kk & 255];
                Vec3 c = ld3(pl.randvec[id]);
                Vec3 weight_v(u - (double)a, v - (double)b, w - (double)d);
                accum += rtm::dot(c, weight_v)
                       * ((double)a * uu + (double)(1 - a) * (1.0 - uu))
                       * ((double)b * vv + (double)(1 - b) * (1.0 - vv))
                       * ((double)d * ww + (double)(1 - d) * (1.0 - ww));
            }
    return accum;
}
RT_DEV double perlin_turb(const rt_perlin &pl, Vec3 p, int depth) {
    double accum = 0.0;
    Vec3 tmp_p = p;
    double weight = 1.0;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pl, tmp_p);
        weight *= 0.5;
        tmp_p = tmp_p * 2.0;
    }
    return rtm::fabs_(accum);
}
RT_DEV Vec3 texture_value(const SceneDev &s, uint32_t tex, double u, double v, Vec3 p) {
    // CheckerTexture only selects one of its two children: resolve iteratively.
    for (int lvl = 0; lvl < 8; lvl++) {
        const rt_texture &t = s.textures[tex];
        if (t.kind != RT_TEX_CHECKER) break;
        double sines = rtm::sin_(p.x * 10.0) * rtm::sin_(p.y * 10.0) * rtm::sin_(p.z * 10.0);
        tex = sines < 0.0 ? t.a : t.b;
    }
    const rt_texture &t = s.textures[tex];
    if (t.kind == RT_TEX_NOISE) {
        double k = 1.0 + rtm::sin_(t.scale * p.z + 10.0 * perlin_turb(s.perlins[t.a], p, 7));
        return Vec3(1.0, 1.0, 1.0) * 0.5 * k;
    }
    if (t.kind == RT_TEX_IMAGE) {
        const rt_image &im = s.images[t.a];
        if ((uint64_t)im.width * im.height == 0) return Vec3(0.0, 1.0, 1.0);
        double uc = rtm::clamp_(u, 0.0, 1.0), vc = rtm::clamp_(v, 0.0, 1.0);
        uint64_t i = rtm::f64_as_usize(uc * (double)im.width);
        uint64_t j = rtm::f64_as_usize(vc * (double)im.height);
        if (i >= im.width) i = im.width - 1;
        if (j >= im.height) j = im.height - 1;
        double color_scale = 1.0 / 255.999;
        const uint8_t *px = s.image_data + im.offset + 3 * (j * im.width + i);
        return Vec3((double)px[0] * color_scale, (double)px[1] * color_scale, (double)px[2] * color_scale);
    }
    return ld3(t.color);   // SolidColor (a Checker nested deeper than 8 falls back to its colour field)
}

// ---- samplers, vec.rs:69-117, pdf.rs:12-21 ------------------------------------------
RT_DEV Vec3 random_in_unit_sphere(Rng &rng) {
    Vec3 p;
    for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
        double x = rng.gen_range(-1.0, 1.0), y = rng.gen_range(-1.0, 1.0), z = rng.gen_range(-1.0, 1.0);
        p = Vec3(x, y, z);
        if (p.length() < 1.0) break;
    }
    return p;
}
RT_DEV Vec3 random_in_unit_disk(Rng &rng) {
    Vec3 p;
    for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
        double x = rng.gen_range(-1.0, 1.0), y = rng.gen_range(-1.0, 1.0);
        p = Vec3(x, y, 0.0);
        if (p.length() < 1.0) break;
    }
    return p;
}
RT_DEV Vec3 random_to_sphere(Rng &rng, double radius, double dis_sqr) {
    double r1 = rng.gen_f64();
    double r2 = rng.gen_f64();
    double z = 1.0 + r2 * (rtm::sqrt_(1.0 - radius * radius / dis_sqr) - 1.0);
    double phi = 2.0 * rtm::PI * r1;
    double sp, cp;
    rtm::sincos_(phi, sp, cp);
    double x = cp * rtm::sqrt_(1.0 - z * z);
    double y = sp * rtm::sqrt_(1.0 - z * z);
    return Vec3(x, y, z);
}
RT_DEV Vec3 random_cosine_direction(Rng &rng) {
    double r1 = rng.gen_f64();
    double r2 = rng.gen_f64();
    double z = rtm::sqrt_(1.0 - r2);
    double phi = 2.0 * rtm::PI * r1;
    double sp, cp;
    rtm::sincos_(phi, sp, cp);
    double x = cp * rtm::sqrt_(r2);
    double y = sp * rtm::sqrt_(r2);
    return Vec3(x, y, z);
}

// ---- light list: HittableList::pdf_value / random over Sphere and rect lights -------
// (sphere.rs:75-90, aarect.rs:74-93,157-176,240-259, mod.rs:62-67,121-132)
template <bool STATS>
RT_DEV double lights_pdf_value(const SceneDev &s, Vec3 o, Vec3 v, Counters<STATS> &cnt) {
    double sum = 0.0;
    for (uint32_t li = 0; li < s.n_lights; li++) {
        uint32_t ref = s.lights[li];
        uint32_t kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
        double val = 0.0;
        XRay r{o, v};
        if (!(ref & RT_REF_FLIP)) {
            if (kind == RT_KIND_SPHERE) {
                const rt_sphere &q = s.spheres[idx];
                cnt.light_pdf();
                double t;
                if (sphere_t(ld3(q.center), q.radius, r, v.length_sqr(), 0.001, rtm::INF, t)) {
                    double cos_max = rtm::sqrt_(1.0 - q.radius * q.radius / (ld3(q.center) - o).length_sqr());
                    double solid_angle = 2.0 * rtm::PI * (1.0 - cos_max);
                    val = 1.0 / solid_angle;
                }
            } else if (kind == RT_KIND_RECT) {
                const rt_rect &q = s.rects[idx];
                cnt.light_pdf();
                double t;
                if (rect_t(q.axis, q.a0, q.a1, q.b0, q.b1, q.k, r, 0.001, rtm::INF, t)) {
                    HitRec rec;
                    RectP rp{q.axis, q.a0, q.a1, q.b0, q.b1, q.k};
                    rect_record(rp, q.mat, r, t, rec);
                    double area = (q.a1 - q.a0) * (q.b1 - q.b0);
                    double dis_sqr = rec.t * rec.t * v.length_sqr();
                    double cosv = rtm::fabs_(rtm::dot(v, rec.normal) / v.length());
                    val = dis_sqr / (cosv * area);
                }
            }
        }
        sum += val;
    }
    return sum / (double)s.n_lights;
}
RT_DEV Vec3 lights_random(const SceneDev &s, Vec3 o, Rng &rng) {
    uint64_t target = rng.gen_index(s.n_lights);
    uint32_t ref = s.lights[target];
    uint32_t kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
    if (!(ref & RT_REF_FLIP)) {
        if (kind == RT_KIND_SPHERE) {
            const rt_sphere &q = s.spheres[idx];
            Vec3 direction = ld3(q.center) - o;
            double dis_sqr = direction.length_sqr();
            rtm::Onb uvw = rtm::onb_from_w(direction);
            return uvw.local_vec(random_to_sphere(rng, q.radius, dis_sqr));
        }
        if (kind == RT_KIND_RECT) {
            const rt_rect &q = s.rects[idx];
            double a = rng.gen_range(q.a0, q.a1);
            double b = rng.gen_range(q.b0, q.b1);
            Vec3 random_point = q.axis == RT_RECT_XY ? Vec3(a, b, q.k) : q.axis == RT_RECT_XZ ? Vec3(a, q.k, b) : Vec3(q.k, a, b);
            return random_point - o;
        }
    }
    return Vec3(1.0, 0.0, 0.0);
}

RT_DEV double reflectance(double cosv, double ref_idx) {            // material/mod.rs:112-116
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    double x = 1.0 - cosv;
    double x2 = x * x;
    return r0 + (1.0 - r0) * (x2 * x2 * x);
}

// Camera::get_ray, camera.rs:64-73.
RT_DEV Ray get_ray(const rt_camera &cam, double sx, double ty, Rng &rng) {
    Vec3 rd = random_in_unit_disk(rng) * cam.lens_radius;
    Vec3 offset = ld3(cam.u) * rd.x + ld3(cam.v) * rd.y;
    Vec3 origin = ld3(cam.origin);
    Vec3 orig = origin + offset;
    Vec3 dir = ld3(cam.lower_left_corner) + ld3(cam.horizontal) * sx + ld3(cam.vertical) * ty - origin - offset;
    double tm = rng.gen_range(cam.time0, cam.time1);
    return Ray(orig, dir, tm);
}

} // namespace

// =====================================================================================
// The megakernel.
// =====================================================================================
template <int STACK, bool STATS>
__global__ void __launch_bounds__(kBlock) pt_megakernel(const SceneDev s, const RenderArgs a) {
    __shared__ uint32_t stack_lds[STACK * kBlock];
    uint32_t *stk = stack_lds + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    Counters<STATS> cnt;

    // Work item (pixel slot x sample chunk) owned by this lane.
    bool have_item = false, done = false, alive = false;
    uint64_t slot = 0;
    uint32_t chunk_id = 0, smp = 0, smp_end = 0, px = 0, py = 0, frame = 0;
    Vec3 pixel_sum;
    // Path state.
    Ray r;
    Vec3 T, A;
    int depth = 0;
    Rng rng;
    const Vec3 background = ld3(a.background);

    for (;;) {
        // ---- item bookkeeping + wave-level refill ------------------------------------
        bool need = !done && !alive && (!have_item || smp == smp_end);
        if (need && have_item) {
            double *o = a.partial + ((uint64_t)chunk_id * a.n_pixels + slot) * 3;
            o[0] = pixel_sum.x; o[1] = pixel_sum.y; o[2] = pixel_sum.z;
            have_item = false;
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
                if (item >= a.n_items) {
                    done = true;
                } else {
                    slot = item / a.n_chunks;
                    chunk_id = (uint32_t)(item - slot * a.n_chunks);
                    uint64_t yi = slot / a.width;
                    px = (uint32_t)(slot - yi * a.width);
                    uint32_t g = a.row_ids[yi];
                    frame = g / a.height;
                    py = g - frame * a.height;
                    smp = chunk_id * a.chunk;
                    smp_end = smp + a.chunk < a.spp ? smp + a.chunk : a.spp;
                    pixel_sum = Vec3(0.0, 0.0, 0.0);
                    have_item = true;
                }
            }
        }
        if (__ballot(!done) == 0ull) break;

        // ---- start the next sample of the item (main.rs:144-149) ---------------------
        if (!done && !alive && smp < smp_end) {
            uint64_t pixel = (uint64_t)py * a.width + px;
            rng = Rng(rtm::path_key(a.seed, frame, pixel, smp));
            double rand_u = rng.gen_f64();
            double rand_v = rng.gen_f64();
            double u = ((double)px + rand_u) / (double)(a.width - 1);
            double v = ((double)py + rand_v) / (double)(a.height - 1);
            r = get_ray(a.cam, u, v, rng);
            T = Vec3(1.0, 1.0, 1.0);
            A = Vec3(0.0, 0.0, 0.0);
            depth = (int)a.max_depth;
            alive = true;
            smp++;
            cnt.path();
        }

        // ---- one level of ray_color (main.rs:233-278) -----------------------------------
        if (alive) {
            bool end_path = false;
            if (depth <= 0) {
                end_path = true;                                  // returns (0,0,0)
            } else {
                cnt.ray();
                Winner w;
                if (!trace<STACK, STATS>(s, stk, r, a.t_min, rng, w, cnt)) {
                    A = A + T * background;
                    end_path = true;
                } else {
                    HitRec rec;
                    winner_record(s, r, w, rec);
                    const rt_material &mat = s.materials[rec.mat];
                    uint32_t mk = mat.kind;
                    if (mk == RT_MAT_DIFFUSE_LIGHT) {             // emitted, scatter = None
                        Vec3 emit = rec.front_face ? texture_value(s, mat.tex, rec.u, rec.v, rec.p) : Vec3(0.0, 0.0, 0.0);
                        A = A + T * emit;
                        end_path = true;
                    } else if (mk == RT_MAT_LAMBERTIAN) {
                        Vec3 att = texture_value(s, mat.tex, rec.u, rec.v, rec.p);
                        rtm::Onb uvw = rtm::onb_from_w(rec.normal);
                        Vec3 dir;
                        double pdf_val;
                        if (s.n_lights == 0) {
                            dir = uvw.local_vec(random_cosine_direction(rng));
                            double cosv = rtm::dot(rtm::to_unit(dir), uvw.w);
                            pdf_val = cosv <= 0.0 ? 0.0 : cosv / rtm::PI;
                        } else {
                            if (rng.gen_range(0.0, 1.0) < 0.5) dir = lights_random(s, rec.p, rng);
                            else dir = uvw.local_vec(random_cosine_direction(rng));
                            double lp = lights_pdf_value<STATS>(s, rec.p, dir, cnt);
                            double cosv = rtm::dot(rtm::to_unit(dir), uvw.w);
                            double cp = cosv <= 0.0 ? 0.0 : cosv / rtm::PI;
                            pdf_val = 0.5 * lp + 0.5 * cp;
                        }
                        double cosine = rtm::dot(rec.normal, rtm::to_unit(dir));
                        double spdf = cosine < 0.0 ? 0.0 : cosine / rtm::PI;
                        T = (T * (att * spdf)) / pdf_val;
                        r = Ray(rec.p, dir, r.tm);
                        depth--;
                    } else if (mk == RT_MAT_METAL) {
                        Vec3 reflected = rtm::reflect(rtm::to_unit(r.dir), rec.normal);
                        Vec3 d = reflected + random_in_unit_sphere(rng) * mat.param;
                        T = T * ld3(mat.albedo);
                        r = Ray(rec.p, d, 0.0);                   // time = 0., material/mod.rs:91
                        depth--;
                    } else if (mk == RT_MAT_DIELECTRIC) {
                        double refraction_ratio = rec.front_face ? 1.0 / mat.param : mat.param;
                        Vec3 unit_direction = rtm::to_unit(r.dir);
                        double cos_theta = rtm::fmin_(rtm::dot(-unit_direction, rec.normal), 1.0);
                        double sin_theta = rtm::sqrt_(1.0 - cos_theta * cos_theta);
                        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
                        double random_double = rng.gen_range(0.0, 1.0);
                        Vec3 d = (cannot_refract || reflectance(cos_theta, refraction_ratio) > random_double)
                                     ? rtm::reflect(unit_direction, rec.normal)
                                     : rtm::refract(unit_direction, rec.normal, refraction_ratio);
                        T = T * Vec3(1.0, 1.0, 1.0);
                        r = Ray(rec.p, d, r.tm);
                        depth--;
                    } else {                                      // Isotropic
                        Vec3 att = texture_value(s, mat.tex, rec.u, rec.v, rec.p);
                        Vec3 d = random_in_unit_sphere(rng);
                        T = T * att;
                        r = Ray(rec.p, d, r.tm);
                        depth--;
                    }
                }
            }
            if (end_path) {
                pixel_sum = pixel_sum + A;
                alive = false;
                cnt.draws(rng.draws);
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

// pt_common.hpp — device functions shared by the megakernel (pt_kernel.hip) and the
// wavefront engine (pt_wavefront.hip): movers, primitive tests, the winner's hit record,
// textures, samplers, light pdfs, camera — each one the f64 restatement of the reference
// lines cited next to it, through rt_math.h so that results are bit-identical to the oracle.
#ifndef RT2022_PT_COMMON_HPP
#define RT2022_PT_COMMON_HPP

#include <hip/hip_runtime.h>

#include "../rt_math.h"
#include "pt_device.h"

namespace rt2022 {

using rtm::Ray;
using rtm::Rng;
using rtm::Vec3;

#define RT_DEV __device__ __forceinline__

namespace {


enum Op : uint32_t {
    OP_NODE = 0,     // AABB test of a BvhNode
    OP_SPHERE = 1,   // Sphere / MovingSphere
    OP_RECT = 2,
    OP_BOX = 3,
    OP_MEDIUM = 4,
    OP_MISC = 5,     // Triangle, Ring
    OP_CTX = 6,      // enter / leave a mover, expand a HittableList
    OP_SHADE = 7,    // traversal finished (or no path yet): shade, next ray / sample / item
    OP_IDLE = 8,     // no work left
    OP_COUNT = 8     // labels that take part in the vote
};

// Stack sentinels live in the unused kind codes 12..15 so that classify() is one table look-up.
constexpr uint32_t REF_MED1 = 12u << RT_REF_KIND_SHIFT;       // medium: first boundary query finished
constexpr uint32_t REF_MED2 = 13u << RT_REF_KIND_SHIFT;       // medium: second boundary query finished
constexpr uint32_t REF_EMPTY = 14u << RT_REF_KIND_SHIFT;      // stack exhausted
constexpr uint32_t REF_POPCTX = 15u << RT_REF_KIND_SHIFT;     // leave the innermost mover

// kind -> op, 4 bits per kind (NODE..LIST, 12/13 unused, EMPTY, POPCTX)
constexpr unsigned long long kClassifyTable = ((unsigned long long)OP_MEDIUM << 48) | ((unsigned long long)OP_MEDIUM << 52) |
                                     ((unsigned long long)OP_SHADE << 56) | ((unsigned long long)OP_CTX << 60) | (unsigned long long)OP_NODE | ((unsigned long long)OP_SPHERE << 4) | ((unsigned long long)OP_SPHERE << 8) |
                                     ((unsigned long long)OP_RECT << 12) | ((unsigned long long)OP_BOX << 16) | ((unsigned long long)OP_MISC << 20) |
                                     ((unsigned long long)OP_MISC << 24) | ((unsigned long long)OP_MEDIUM << 28) | ((unsigned long long)OP_CTX << 32) |
                                     ((unsigned long long)OP_CTX << 36) | ((unsigned long long)OP_CTX << 40) | ((unsigned long long)OP_CTX << 44);
RT_DEV uint32_t classify(uint32_t ref, unsigned long long table = kClassifyTable) {
    return (uint32_t)(table >> (RT_REF_KIND(ref) * 4)) & 0xFu;
}


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
    RT_DEV void flush_wave(StatsDev *) {}
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
    // Same, from a point every lane of the wave reaches together: one atomic per wave and counter.
    RT_DEV static unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
        return v;
    }
    RT_DEV void flush_wave(StatsDev *st) {
        if (!st) return;
        const bool first = (threadIdx.x & 63u) == 0;
        unsigned long long v;
        v = wave_sum(paths); if (first && v) atomicAdd(&st->paths, v);
        v = wave_sum(rays); if (first && v) atomicAdd(&st->rays, v);
        v = wave_sum(nodes); if (first && v) atomicAdd(&st->node_visits, v);
        v = wave_sum(lpdf); if (first && v) atomicAdd(&st->light_pdf_tests, v);
        v = wave_sum(ndraws); if (first && v) atomicAdd(&st->rng_draws, v);
#pragma unroll
        for (int i = 0; i < RT_KIND_COUNT; i++) { v = wave_sum(prims[i]); if (first && v) atomicAdd(&st->prim_tests[i], v); }
        paths = rays = nodes = lpdf = ndraws = 0;
#pragma unroll
        for (int i = 0; i < RT_KIND_COUNT; i++) prims[i] = 0;
    }
};

// ---- movers: Translate / RotateY / Zoom (hittable/mod.rs:165-174,235-264,321-330) ----
// (the mover's three parameters by value: the caller decides where its record comes from)
RT_DEV XRay xform_ray_p(uint32_t kind, double p0, double p1, double p2, XRay r) {
    if (kind == RT_KIND_TRANSLATE) {
        r.o = r.o - Vec3(p0, p1, p2);
    } else if (kind == RT_KIND_ROTATE_Y) {
        double sin_theta = p0, cos_theta = p1;
        double ox = cos_theta * r.o.x - sin_theta * r.o.z;
        double oz = sin_theta * r.o.x + cos_theta * r.o.z;
        double dx = cos_theta * r.d.x - sin_theta * r.d.z;
        double dz = sin_theta * r.d.x + cos_theta * r.d.z;
        r.o.x = ox; r.o.z = oz; r.d.x = dx; r.d.z = dz;
    } else {
        r.o = r.o / p0;
    }
    return r;
}
RT_DEV XRay xform_ray(const SceneDev &s, uint32_t ref, XRay r) {
    const rt_xform &x = s.xforms[RT_REF_INDEX(ref)];
    return xform_ray_p(RT_REF_KIND(ref), x.p[0], x.p[1], x.p[2], r);
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
// (1, r3: the four refs picked with shifts of two 64-bit words. The select form below — 0 — is what the compiler turns into a
// dynamically indexed private array: 32-48 bytes of scratch per lane in every traversal kernel with movers)
#ifndef RT2022_CHAIN_PACKED
#define RT2022_CHAIN_PACKED 1
#endif
struct Chain {
    uint32_t c0, c1, c2, c3;
    uint32_t n;
#if RT2022_CHAIN_PACKED
    // (the four refs picked with shifts of two 64-bit words, which the compiler cannot turn into an indexed private array)
    RT_DEV uint32_t at(uint32_t i) const {
        const uint64_t lo = (uint64_t)c0 | ((uint64_t)c1 << 32), hi = (uint64_t)c2 | ((uint64_t)c3 << 32);
        return (uint32_t)(((i & 2u) ? hi : lo) >> ((i & 1u) * 32u));
    }
    RT_DEV void push(uint32_t ref) {
        c0 = n == 0 ? ref : c0; c1 = n == 1 ? ref : c1; c2 = n == 2 ? ref : c2; c3 = n >= 3 ? ref : c3;
        n++;
    }
#else
    RT_DEV uint32_t at(uint32_t i) const { return i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : c3; }
    RT_DEV void push(uint32_t ref) {
        if (n == 0) c0 = ref; else if (n == 1) c1 = ref; else if (n == 2) c2 = ref; else c3 = ref;
        n++;
    }
#endif
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
    // (select values, not addresses: a phi over &r.o.x / &r.o.y keeps the ray in scratch memory)
    const double ox = r.o.x, oy = r.o.y, oz = r.o.z, dx = r.d.x, dy = r.d.y, dz = r.d.z;
    const bool xy = axis == RT_RECT_XY, xz = axis == RT_RECT_XZ;
    const double ok = xy ? oz : xz ? oy : ox, dk = xy ? dz : xz ? dy : dx;
    const double oa = (xy || xz) ? ox : oy, da = (xy || xz) ? dx : dy;
    const double ob = xy ? oy : oz, db = xy ? dy : dz;
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

// ---- traversal state --------------------------------------------------------------------
struct Winner {
    double t;
    uint32_t leaf;      // ref of the winning leaf (with its flip bit)
    uint32_t face;      // box side
    Chain chain;        // movers enclosing it
};

// get_sphere_uv, sphere.rs:30-34.
RT_DEV void sphere_uv(Vec3 p, double &u, double &v) {
    double theta = rtm::acos_(-p.y);
    double phi = rtm::atan2_(-p.z, p.x) + rtm::PI;
    u = phi / (2.0 * rtm::PI);
    v = theta / rtm::PI;
}

RT_DEV void rect_record(const RectP &q, uint32_t mat, const XRay &r, double t, HitRec &rec) {   // aarect.rs:51-71
    const double ox = r.o.x, oy = r.o.y, oz = r.o.z, dx = r.d.x, dy = r.d.y, dz = r.d.z;
    const bool xy = q.axis == RT_RECT_XY, xz = q.axis == RT_RECT_XZ;
    const double oa = (xy || xz) ? ox : oy, da = (xy || xz) ? dx : dy;
    const double ob = xy ? oy : oz, db = xy ? dy : dz;
    const Vec3 outward_normal(xy || xz ? 0.0 : 1.0, xz ? 1.0 : 0.0, xy ? 1.0 : 0.0);
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
// `want_uv` = false skips the sphere's acos / atan2 when no texture will read (u, v): a pure
// function of the hit, so leaving it out cannot change anything else.
RT_DEV void winner_record(const SceneDev &s, const Ray &wr, const Winner &w, HitRec &rec, bool want_uv = true) {
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
            rec.u = 0.0; rec.v = 0.0;
            if (want_uv) sphere_uv(outward_normal, rec.u, rec.v);
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
// One entry of `lights` with its primitive's numbers beside the ref, so that a kernel may keep the list in LDS.
struct LightRec {
    double v[5];           // Sphere: center x y z, radius, -;  rect: a0, a1, b0, b1, k
    uint32_t axis;         // rect axis
    uint32_t ref;          // the entry of `lights` (kind, FlipFace bit)
};
RT_DEV LightRec fetch_light(const SceneDev &s, uint32_t li) {
    LightRec q;
    q.ref = s.lights[li];
    q.axis = 0;
    q.v[0] = q.v[1] = q.v[2] = q.v[3] = q.v[4] = 0.0;
    const uint32_t kind = RT_REF_KIND(q.ref), idx = RT_REF_INDEX(q.ref);
    if (kind == RT_KIND_SPHERE) {
        const rt_sphere &p = s.spheres[idx];
        q.v[0] = p.center[0]; q.v[1] = p.center[1]; q.v[2] = p.center[2]; q.v[3] = p.radius;
    } else if (kind == RT_KIND_RECT) {
        const rt_rect &p = s.rects[idx];
        q.v[0] = p.a0; q.v[1] = p.a1; q.v[2] = p.b0; q.v[3] = p.b1; q.v[4] = p.k; q.axis = p.axis;
    }
    return q;
}
// One light's pdf_value(o, v): Sphere (sphere.rs:75-83), X?Rect (aarect.rs:74-83 and twins); anything else — a
// FlipFace'd light included — answers the trait default 0 (mod.rs:62-64).
template <bool STATS>
RT_DEV double light_pdf_value(const LightRec &q, Vec3 o, Vec3 v, Counters<STATS> &cnt) {
    const uint32_t kind = RT_REF_KIND(q.ref);
    double val = 0.0;
    XRay r{o, v};
    if (!(q.ref & RT_REF_FLIP)) {
        if (kind == RT_KIND_SPHERE) {
            const Vec3 center(q.v[0], q.v[1], q.v[2]);
            const double radius = q.v[3];
            cnt.light_pdf();
            double t;
            if (sphere_t(center, radius, r, v.length_sqr(), 0.001, rtm::INF, t)) {
                double cos_max = rtm::sqrt_(1.0 - radius * radius / (center - o).length_sqr());
                double solid_angle = 2.0 * rtm::PI * (1.0 - cos_max);
                val = 1.0 / solid_angle;
            }
        } else if (kind == RT_KIND_RECT) {
            cnt.light_pdf();
            double t;
            if (rect_t(q.axis, q.v[0], q.v[1], q.v[2], q.v[3], q.v[4], r, 0.001, rtm::INF, t)) {
                HitRec rec;
                RectP rp{q.axis, q.v[0], q.v[1], q.v[2], q.v[3], q.v[4]};
                rect_record(rp, 0u, r, t, rec);
                double area = (q.v[1] - q.v[0]) * (q.v[3] - q.v[2]);
                double dis_sqr = rec.t * rec.t * v.length_sqr();
                double cosv = rtm::fabs_(rtm::dot(v, rec.normal) / v.length());
                val = dis_sqr / (cosv * area);
            }
        }
    }
    return val;
}
// One light's random(o): Sphere (sphere.rs:85-90), X?Rect (aarect.rs:85-93 and twins); else (1, 0, 0) (mod.rs:65-67).
RT_DEV Vec3 light_random(const LightRec &q, Vec3 o, Rng &rng) {
    const uint32_t kind = RT_REF_KIND(q.ref);
    if (!(q.ref & RT_REF_FLIP)) {
        if (kind == RT_KIND_SPHERE) {
            const Vec3 center(q.v[0], q.v[1], q.v[2]);
            Vec3 direction = center - o;
            double dis_sqr = direction.length_sqr();
            rtm::Onb uvw = rtm::onb_from_w(direction);
            return uvw.local_vec(random_to_sphere(rng, q.v[3], dis_sqr));
        }
        if (kind == RT_KIND_RECT) {
            double a = rng.gen_range(q.v[0], q.v[1]);
            double b = rng.gen_range(q.v[2], q.v[3]);
            const double k = q.v[4];
            Vec3 random_point = q.axis == RT_RECT_XY ? Vec3(a, b, k) : q.axis == RT_RECT_XZ ? Vec3(a, k, b) : Vec3(k, a, b);
            return random_point - o;
        }
    }
    return Vec3(1.0, 0.0, 0.0);
}
// HittableList::pdf_value / random (mod.rs:121-132) over a list reached through `fetch(i)`.
template <bool STATS, class Fetch>
RT_DEV double lights_pdf_value_of(uint32_t n_lights, Fetch fetch, Vec3 o, Vec3 v, Counters<STATS> &cnt) {
    double sum = 0.0;
    for (uint32_t li = 0; li < n_lights; li++) sum += light_pdf_value<STATS>(fetch(li), o, v, cnt);
    return sum / (double)n_lights;
}
template <class Fetch>
RT_DEV Vec3 lights_random_of(uint32_t n_lights, Fetch fetch, Vec3 o, Rng &rng) {
    uint64_t target = rng.gen_index(n_lights);
    return light_random(fetch((uint32_t)target), o, rng);
}
template <bool STATS>
RT_DEV double lights_pdf_value(const SceneDev &s, Vec3 o, Vec3 v, Counters<STATS> &cnt) {
    return lights_pdf_value_of<STATS>(s.n_lights, [&](uint32_t li) { return fetch_light(s, li); }, o, v, cnt);
}
RT_DEV Vec3 lights_random(const SceneDev &s, Vec3 o, Rng &rng) {
    return lights_random_of(s.n_lights, [&](uint32_t li) { return fetch_light(s, li); }, o, rng);
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


} // namespace rt2022
#endif

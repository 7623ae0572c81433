// rt_oracle.cpp — CPU restatement of the reference's path-tracing hot loop.
//
// TEST INFRASTRUCTURE ONLY. This file is the parity oracle and the CPU baseline:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// it. The product (raytracer_2022_amd/) never links or calls it.
//
// It follows the reference op for op, in f64, keeping its structure — recursive
// ray_color, recursive left-then-right BVH with the three divisions per node,
// dynamic dispatch per object (here: a switch on the ref kind), uv per accepted
// sphere candidate, the span-1 double test — and reads the same flattened scene
// (include/rt2022.h) and the same seeded RNG streams (rt_math.h) as the HIP path.
//
// Pinning: the reference ships no tests, golden vectors or seedable RNG
//  and cannot be built here (Rust, no toolchain). The oracle is
// pinned instead by analytic known-answer tests (tests/test_oracle_kat.py) written
// against the reference lines cited at each function below; RNG bit streams and
// JPEG/OBJ decoding are "parity unpinned" (third-party crates absent).
//
// Reference: /root/reference/raytracer/src (cited as file:line).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../include/rt2022.h"
// Vectors, reflect / refract, Onb, the RNG's conversions and the casts: rt_math.h, shared with the device so that HIP == oracle
// bit for bit — or, -DRTO_OWN_MATH (`make own`), the oracle's own restatement of the same reference lines, sharing nothing with
// the product (rto_math.h says what that build is for; it takes its transcendentals from the platform libm like -DRTO_LIBM).
#ifdef RTO_OWN_MATH
#include "rto_math.h"
#ifndef RTO_LIBM
#define RTO_LIBM
#endif
#else
#include "../raytracer_2022_amd/csrc/rt_math.h"
#endif
#include "rt_oracle.h"

using rtm::Ray;
using rtm::Rng;
using rtm::Vec3;

// The five transcendentals of the render path (sphere.rs:30-34, constantmedium.rs:61, texture/mod.rs:52,77,
// pdf.rs:15-18, vec.rs:112-115). Default build: the fdlibm restatements of rt_math.h, shared with the device so that
// HIP == oracle bit for bit. -DRTO_LIBM (librt_oracle_libm.so, `make libm`): the platform libm — what the Rust
// reference itself links (f64::sin/cos/acos/atan2/ln are libm calls) — so that tools/libm_sensitivity.py can measure
// what the <= 1 ulp between the two does to a pixel. That build shares NO transcendental with the product.
namespace om {
#ifdef RTO_LIBM
inline double sin_(double x) { return ::sin(x); }
inline double cos_(double x) { return ::cos(x); }
inline double acos_(double x) { return ::acos(x); }
inline double atan2_(double y, double x) { return ::atan2(y, x); }
inline double log_(double x) { return ::log(x); }
constexpr int kLibm = 1;
#else
inline double sin_(double x) { return rtm::sin_(x); }
inline double cos_(double x) { return rtm::cos_(x); }
inline double acos_(double x) { return rtm::acos_(x); }
inline double atan2_(double y, double x) { return rtm::atan2_(y, x); }
inline double log_(double x) { return rtm::log_(x); }
constexpr int kLibm = 0;
#endif
} // namespace om
typedef Vec3 Color;
typedef Vec3 Point3;

namespace {

thread_local std::string g_err;

struct HitRecord {                                            // hittable/mod.rs:18-57
    Point3 p;
    Vec3 normal;
    double t = 0, u = 0, v = 0;
    bool front_face = false;
    uint32_t mat = 0;
    void set_face_normal(const Ray &r, Vec3 outward_normal) { // mod.rs:49-56
        front_face = rtm::dot(r.dir, outward_normal) < 0.0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

struct Ctx {
    const rt_scene_desc *s;
    Rng *rng;
    rt_stats *st;           // never null
    bool in_light_pdf = false;
};

inline Vec3 v3(const double d[3]) { return Vec3(d[0], d[1], d[2]); }

bool hit(Ctx &c, uint32_t ref, const Ray &r, double t_min, double t_max, HitRecord &rec);

// ---- AABB::hit, hittable/bvh/aabb.rs:15-32 -----------------------------------
bool aabb_hit(const rt_bvh_node &n, const Ray &r, double tmin, double tmax) {
    double t_min = tmin, t_max = tmax;
    for (int i = 0; i < 3; i++) {
        double inv_d = 1.0 / r.dir[i];
        double t0 = (n.bmin[i] - r.orig[i]) * inv_d;
        double t1 = (n.bmax[i] - r.orig[i]) * inv_d;
        if (inv_d < 0.0) { double tmp = t0; t0 = t1; t1 = tmp; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

// ---- BvhNode::hit, hittable/bvh/mod.rs:86-101 --------------------------------
bool node_hit(Ctx &c, const rt_bvh_node &n, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    c.st->node_visits++;
    if (!aabb_hit(n, r, t_min, t_max)) return false;
    HitRecord recl;
    if (hit(c, n.left, r, t_min, t_max, recl)) {
        HitRecord recr;
        if (hit(c, n.right, r, t_min, recl.t, recr)) rec = recr;
        else rec = recl;
        return true;
    }
    HitRecord recr;
    if (hit(c, n.right, r, t_min, t_max, recr)) { rec = recr; return true; }
    return false;
}

// ---- get_sphere_uv, hittable/sphere.rs:30-34 ---------------------------------
void sphere_uv(Point3 p, double &u, double &v) {
    double theta = om::acos_(-p.y);
    double phi = om::atan2_(-p.z, p.x) + rtm::PI;
    u = phi / (2.0 * rtm::PI);
    v = theta / rtm::PI;
}

// ---- Sphere::hit / MovingSphere::hit, sphere.rs:39-66,138-165 ----------------
bool sphere_hit_at(Point3 center, double radius, uint32_t mat, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    Vec3 oc = r.orig - center;
    double a = r.dir.length_sqr();
    double half_b = rtm::dot(oc, r.dir);
    double cc = oc.length_sqr() - radius * radius;
    double discriminant = half_b * half_b - a * cc;
    if (discriminant < 0.0) return false;
    double sqrtd = rtm::sqrt_(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    Vec3 outward_normal = (r.at(root) - center) / radius;
    double u, v;
    sphere_uv(outward_normal, u, v);
    rec.p = r.at(root);
    rec.normal = outward_normal;
    rec.t = root; rec.u = u; rec.v = v;
    rec.front_face = false;
    rec.mat = mat;
    rec.set_face_normal(r, outward_normal);
    return true;
}
Point3 moving_center(const rt_moving_sphere &s, double time) { // sphere.rs:124-127
    Vec3 c0 = v3(s.center0), c1 = v3(s.center1);
    return c0 + (c1 - c0) * ((time - s.time0) / (s.time1 - s.time0));
}

// ---- XYRect / XZRect / YZRect::hit, aarect.rs:46-72,129-155,212-238 ----------
bool rect_hit(const rt_rect &q, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    double ok, dk, oa, da, ob, db;
    Vec3 outward_normal;
    switch (q.axis) {
        case RT_RECT_XY: ok = r.orig.z; dk = r.dir.z; oa = r.orig.x; da = r.dir.x; ob = r.orig.y; db = r.dir.y; outward_normal = Vec3(0.0, 0.0, 1.0); break;
        case RT_RECT_XZ: ok = r.orig.y; dk = r.dir.y; oa = r.orig.x; da = r.dir.x; ob = r.orig.z; db = r.dir.z; outward_normal = Vec3(0.0, 1.0, 0.0); break;
        default:         ok = r.orig.x; dk = r.dir.x; oa = r.orig.y; da = r.dir.y; ob = r.orig.z; db = r.dir.z; outward_normal = Vec3(1.0, 0.0, 0.0); break;
    }
    double t = (q.k - ok) / dk;
    if (t < t_min || t > t_max) return false;
    double a = oa + t * da;
    double b = ob + t * db;
    if (a < q.a0 || a > q.a1 || b < q.b0 || b > q.b1) return false;
    rec.p = r.at(t);
    rec.normal = outward_normal;
    rec.t = t;
    rec.u = (a - q.a0) / (q.a1 - q.a0);
    rec.v = (b - q.b0) / (q.b1 - q.b0);
    rec.front_face = true;
    rec.mat = q.mat;
    rec.set_face_normal(r, outward_normal);
    return true;
}

// ---- Boxes, boxes.rs:19-82 (six rects through HittableList::hit, mod.rs:90-100)
void box_side(const rt_box &b, int i, rt_rect &q) {
    const double *p0 = b.p0, *p1 = b.p1;
    q.mat = b.mat;
    switch (i) {
        case 0: q.axis = RT_RECT_XY; q.a0 = p0[0]; q.a1 = p1[0]; q.b0 = p0[1]; q.b1 = p1[1]; q.k = p1[2]; break;
        case 1: q.axis = RT_RECT_XY; q.a0 = p0[0]; q.a1 = p1[0]; q.b0 = p0[1]; q.b1 = p1[1]; q.k = p0[2]; break;
        case 2: q.axis = RT_RECT_XZ; q.a0 = p0[0]; q.a1 = p1[0]; q.b0 = p0[2]; q.b1 = p1[2]; q.k = p1[1]; break;
        case 3: q.axis = RT_RECT_XZ; q.a0 = p0[0]; q.a1 = p1[0]; q.b0 = p0[2]; q.b1 = p1[2]; q.k = p0[1]; break;
        case 4: q.axis = RT_RECT_YZ; q.a0 = p0[1]; q.a1 = p1[1]; q.b0 = p0[2]; q.b1 = p1[2]; q.k = p1[0]; break;
        default: q.axis = RT_RECT_YZ; q.a0 = p0[1]; q.a1 = p1[1]; q.b0 = p0[2]; q.b1 = p1[2]; q.k = p0[0]; break;
    }
}
bool box_hit(const rt_box &b, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    bool any = false;
    double closest_so_far = t_max;
    for (int i = 0; i < 6; i++) {
        rt_rect q;
        box_side(b, i, q);
        HitRecord tmp;
        if (rect_hit(q, r, t_min, closest_so_far, tmp)) { closest_so_far = tmp.t; rec = tmp; any = true; }
    }
    return any;
}

// ---- Triangle::hit, triangle.rs:30-77 ----------------------------------------
bool tri_inside(Point3 a, Point3 b, Point3 c, Point3 p) {      // triangle.rs:33-46
    return rtm::dot(rtm::cross(c - a, p - a), rtm::cross(c - a, b - a)) >= 0.0 &&
           rtm::dot(rtm::cross(a - b, p - b), rtm::cross(a - b, c - b)) >= 0.0 &&
           rtm::dot(rtm::cross(b - c, p - c), rtm::cross(b - c, a - c)) >= 0.0;
}
bool triangle_hit(const rt_triangle &tr, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    Point3 a = v3(tr.a), b = v3(tr.b), c = v3(tr.c);
    Vec3 n = rtm::to_unit(rtm::cross(b - a, c - a));
    double t = rtm::dot(r.dir, n);
    t = rtm::dot(a - r.orig, n) / t;
    if (t != t || t < t_min || t > t_max) return false;
    Point3 p = r.orig + r.dir * t;
    if (!tri_inside(a, b, c, p)) return false;
    double a1 = a.x - b.x, b1 = a.x - c.x, c1 = a.x - p.x;
    double a2 = a.y - b.y, b2 = a.y - c.y, c2 = a.y - p.y;
    double beta = (c1 * b2 - b1 * c2) / (a1 * b2 - b1 * a2);
    double gama = (a1 * c2 - a2 * c1) / (a1 * b2 - b1 * a2);
    rec.p = p; rec.normal = n; rec.t = t; rec.u = beta; rec.v = gama;
    rec.front_face = true; rec.mat = tr.mat;
    rec.set_face_normal(r, n);
    return true;
}

// ---- Ring::hit, ring.rs:36-53 -------------------------------------------------
bool ring_hit(const rt_ring &g, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    double t = -r.orig.y / r.dir.y;
    if (t != t || t < t_min || t > t_max) return false;
    Point3 p = r.at(t);
    double d = p.x * p.x + p.z * p.z;
    if (d < g.dis_min || d > g.dis_max) return false;
    rec.p = p; rec.normal = Vec3(0.0, 1.0, 0.0); rec.t = t; rec.u = 0.0; rec.v = 0.0;
    rec.front_face = false; rec.mat = g.mat;
    rec.set_face_normal(r, Vec3(0.0, 1.0, 0.0));
    return true;
}

// ---- ConstantMedium::hit, constantmedium.rs:49-83 -----------------------------
bool medium_hit(Ctx &c, const rt_medium &m, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    HitRecord rec1, rec2;
    if (!hit(c, m.boundary, r, -rtm::INF, rtm::INF, rec1)) return false;
    if (!hit(c, m.boundary, r, rec1.t + 0.0001, rtm::INF, rec2)) return false;
    rec1.t = rtm::fmax_(rec1.t, t_min);
    rec2.t = rtm::fmin_(rec2.t, t_max);
    if (rec1.t >= rec2.t) return false;
    rec1.t = rtm::fmax_(rec1.t, 0.0);
    double ray_length = r.dir.length();
    double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
    double rnd = c.rng->gen_f64();
    double hit_distance = m.neg_inv_density * (om::log_(rnd) / om::log_(rtm::E_));   // rnd.log(E)
    if (hit_distance > distance_inside_boundary) return false;
    rec.p = r.at(rec1.t + hit_distance / ray_length);
    rec.normal = Vec3(1.0, 0.0, 0.0);
    rec.t = rec1.t + hit_distance / ray_length;
    rec.u = 0.0; rec.v = 0.0;
    rec.front_face = true;
    rec.mat = m.mat;
    return true;
}

// ---- Translate / RotateY / Zoom, hittable/mod.rs:165-174,235-264,321-330 -----
bool xform_hit(Ctx &c, const rt_xform &x, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    if (x.kind == RT_KIND_TRANSLATE) {
        Vec3 offset = v3(x.p);
        Ray moved_r(r.orig - offset, r.dir, r.tm);
        if (!hit(c, x.child, moved_r, t_min, t_max, rec)) return false;
        rec.p += offset;
        rec.set_face_normal(moved_r, rec.normal);
        return true;
    }
    if (x.kind == RT_KIND_ROTATE_Y) {
        double sin_theta = x.p[0], cos_theta = x.p[1];
        Vec3 origin = r.orig, direction = r.dir;
        origin.x = cos_theta * r.orig.x - sin_theta * r.orig.z;
        origin.z = sin_theta * r.orig.x + cos_theta * r.orig.z;
        direction.x = cos_theta * r.dir.x - sin_theta * r.dir.z;
        direction.z = sin_theta * r.dir.x + cos_theta * r.dir.z;
        Ray rotated_r(origin, direction, r.tm);
        if (!hit(c, x.child, rotated_r, t_min, t_max, rec)) return false;
        Vec3 p = rec.p, normal = rec.normal;
        p.x = cos_theta * rec.p.x + sin_theta * rec.p.z;
        p.z = -sin_theta * rec.p.x + cos_theta * rec.p.z;
        normal.x = cos_theta * rec.normal.x + sin_theta * rec.normal.z;
        normal.z = -sin_theta * rec.normal.x + cos_theta * rec.normal.z;
        rec.p = p;
        rec.set_face_normal(rotated_r, normal);
        return true;
    }
    // Zoom: origin scaled, direction and t not (mod.rs:322-325).
    double rate = x.p[0];
    Ray moved_r(r.orig / rate, r.dir, r.tm);
    if (!hit(c, x.child, moved_r, t_min, t_max, rec)) return false;
    rec.p *= rate;
    rec.set_face_normal(moved_r, rec.normal);
    return true;
}

// ---- HittableList::hit, hittable/mod.rs:90-100 --------------------------------
bool list_hit(Ctx &c, const rt_list &l, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    bool any = false;
    double closest_so_far = t_max;
    for (uint32_t i = 0; i < l.count; i++) {
        HitRecord tmp;
        if (hit(c, c.s->list_items[l.first + i], r, t_min, closest_so_far, tmp)) { closest_so_far = tmp.t; rec = tmp; any = true; }
    }
    return any;
}

// ---- dyn Hittable::hit dispatch (+ FlipFace, mod.rs:281-288) ------------------
bool hit(Ctx &c, uint32_t ref, const Ray &r, double t_min, double t_max, HitRecord &rec) {
    const rt_scene_desc &s = *c.s;
    uint32_t kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
    bool h;
    if (kind != RT_KIND_NODE) c.st->prim_tests[kind]++;
    switch (kind) {
        case RT_KIND_NODE: h = node_hit(c, s.nodes[idx], r, t_min, t_max, rec); break;
        case RT_KIND_SPHERE: { const rt_sphere &q = s.spheres[idx]; h = sphere_hit_at(v3(q.center), q.radius, q.mat, r, t_min, t_max, rec); break; }
        case RT_KIND_MOVING_SPHERE: {
            const rt_moving_sphere &q = s.moving_spheres[idx];
            // center(r.tm) is evaluated at sphere.rs:139 and again at :158 — same value.
            h = sphere_hit_at(moving_center(q, r.tm), q.radius, q.mat, r, t_min, t_max, rec);
            break;
        }
        case RT_KIND_RECT: h = rect_hit(s.rects[idx], r, t_min, t_max, rec); break;
        case RT_KIND_BOX: h = box_hit(s.boxes[idx], r, t_min, t_max, rec); break;
        case RT_KIND_TRIANGLE: h = triangle_hit(s.triangles[idx], r, t_min, t_max, rec); break;
        case RT_KIND_RING: h = ring_hit(s.rings[idx], r, t_min, t_max, rec); break;
        case RT_KIND_MEDIUM: h = medium_hit(c, s.media[idx], r, t_min, t_max, rec); break;
        case RT_KIND_TRANSLATE: case RT_KIND_ROTATE_Y: case RT_KIND_ZOOM: h = xform_hit(c, s.xforms[idx], r, t_min, t_max, rec); break;
        case RT_KIND_LIST: h = list_hit(c, s.lists[idx], r, t_min, t_max, rec); break;
        default: h = false;
    }
    if (h && (ref & RT_REF_FLIP)) rec.front_face = !rec.front_face;
    return h;
}

// ---- Perlin, texture/perlin.rs:52-112 -----------------------------------------
double perlin_noise(const rt_perlin &pl, Point3 p) {
    double u = p.x - rtm::floor_(p.x);
    double v = p.y - rtm::floor_(p.y);
    double w = p.z - rtm::floor_(p.z);
    u = u * u * (3.0 - 2.0 * u);
    v = v * v * (3.0 - 2.0 * v);
    w = w * w * (3.0 - 2.0 * w);
    int32_t i = rtm::f64_as_i32(rtm::floor_(p.x));
    int32_t j = rtm::f64_as_i32(rtm::floor_(p.y));
    int32_t k = rtm::f64_as_i32(rtm::floor_(p.z));
    Vec3 c[2][2][2];
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                // i + di wraps like Rust release arithmetic.
                int32_t ii = (int32_t)((uint32_t)i + (uint32_t)di), jj = (int32_t)((uint32_t)j + (uint32_t)dj), kk = (int32_t)((uint32_t)k + (uint32_t)dk);
                int32_t id = pl.perm_x[ii & 255] ^ pl.perm_y[jj & 255] ^ pl.perm_z[kk & 255];
                c[di][dj][dk] = v3(pl.randvec[id]);
            }
    // trilinear_interp, perlin.rs:81-99 (second Hermite smoothing).
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int d = 0; d < 2; d++) {
                Vec3 weight_v(u - (double)a, v - (double)b, w - (double)d);
                accum += rtm::dot(c[a][b][d], weight_v)
                       * ((double)a * uu + (double)(1 - a) * (1.0 - uu))
                       * ((double)b * vv + (double)(1 - b) * (1.0 - vv))
                       * ((double)d * ww + (double)(1 - d) * (1.0 - ww));
            }
    return accum;
}
double perlin_turb(const rt_perlin &pl, Point3 p, int depth) {    // perlin.rs:100-112
    double accum = 0.0;
    Point3 tmp_p = p;
    double weight = 1.0;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pl, tmp_p);
        weight *= 0.5;
        tmp_p *= 2.0;
    }
    return rtm::fabs_(accum);
}

// ---- Texture::value, texture/mod.rs:25-139 -------------------------------------
Color texture_value(const rt_scene_desc &s, uint32_t tex, double u, double v, Point3 p) {
    const rt_texture &t = s.textures[tex];
    switch (t.kind) {
        case RT_TEX_SOLID: return v3(t.color);
        case RT_TEX_CHECKER: {                                     // mod.rs:51-60
            double sines = om::sin_(p.x * 10.0) * om::sin_(p.y * 10.0) * om::sin_(p.z * 10.0);
            return sines < 0.0 ? texture_value(s, t.a, u, v, p) : texture_value(s, t.b, u, v, p);
        }
        case RT_TEX_NOISE: {                                       // mod.rs:75-79
            double k = 1.0 + om::sin_(t.scale * p.z + 10.0 * perlin_turb(s.perlins[t.a], p, 7));
            return Color(1.0, 1.0, 1.0) * 0.5 * k;
        }
        default: {                                                 // mod.rs:110-139
            const rt_image &im = s.images[t.a];
            if ((uint64_t)im.width * im.height == 0) return Color(0.0, 1.0, 1.0);
            double uc = rtm::clamp_(u, 0.0, 1.0), vc = rtm::clamp_(v, 0.0, 1.0);
            uint64_t i = rtm::f64_as_usize(uc * (double)im.width);
            uint64_t j = rtm::f64_as_usize(vc * (double)im.height);
            if (i >= im.width) i = im.width - 1;
            if (j >= im.height) j = im.height - 1;
            double color_scale = 1.0 / 255.999;
            const uint8_t *px = s.image_data + im.offset + 3 * (j * im.width + i);
            return Color((double)px[0] * color_scale, (double)px[1] * color_scale, (double)px[2] * color_scale);
        }
    }
}

// ---- samplers, basic/vec.rs:69-117, basic/pdf.rs:12-21 ------------------------
// (Loops cut at RT_MAX_REJECT tries like the device code: unreachable, see rt_math.h.)
Vec3 random_in_unit_sphere(Rng &rng) {
    Vec3 p;
    for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
        double x = rng.gen_range(-1.0, 1.0), y = rng.gen_range(-1.0, 1.0), z = rng.gen_range(-1.0, 1.0);
        p = Vec3(x, y, z);
        if (p.length() < 1.0) return p;
    }
    return p;
}
Vec3 random_in_unit_disk(Rng &rng) {
    Vec3 p;
    for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
        double x = rng.gen_range(-1.0, 1.0), y = rng.gen_range(-1.0, 1.0);
        p = Vec3(x, y, 0.0);
        if (p.length() < 1.0) return p;
    }
    return p;
}
Vec3 random_to_sphere(Rng &rng, double radius, double dis_sqr) {
    double r1 = rng.gen_f64();
    double r2 = rng.gen_f64();
    double z = 1.0 + r2 * (rtm::sqrt_(1.0 - radius * radius / dis_sqr) - 1.0);
    double phi = 2.0 * rtm::PI * r1;
    double x = om::cos_(phi) * rtm::sqrt_(1.0 - z * z);
    double y = om::sin_(phi) * rtm::sqrt_(1.0 - z * z);
    return Vec3(x, y, z);
}
Vec3 random_cosine_direction(Rng &rng) {
    double r1 = rng.gen_f64();
    double r2 = rng.gen_f64();
    double z = rtm::sqrt_(1.0 - r2);
    double phi = 2.0 * rtm::PI * r1;
    double x = om::cos_(phi) * rtm::sqrt_(r2);
    double y = om::sin_(phi) * rtm::sqrt_(r2);
    return Vec3(x, y, z);
}

// ---- Hittable::pdf_value / random for light objects ----------------------------
// Sphere: sphere.rs:75-90. Rects: aarect.rs:74-93,157-176,240-259. Everything
// else (incl. any wrapper, FlipFace too): the trait defaults 0 / (1,0,0), mod.rs:62-67.
double light_pdf_value(Ctx &c, uint32_t ref, Point3 o, Vec3 v) {
    const rt_scene_desc &s = *c.s;
    if (ref & RT_REF_FLIP) return 0.0;
    uint32_t kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
    HitRecord rec;
    if (kind == RT_KIND_SPHERE) {
        const rt_sphere &q = s.spheres[idx];
        c.st->light_pdf_tests++;
        if (!sphere_hit_at(v3(q.center), q.radius, q.mat, Ray(o, v, 0.0), 0.001, rtm::INF, rec)) return 0.0;
        double cos_max = rtm::sqrt_(1.0 - q.radius * q.radius / (v3(q.center) - o).length_sqr());
        double solid_angle = 2.0 * rtm::PI * (1.0 - cos_max);
        return 1.0 / solid_angle;
    }
    if (kind == RT_KIND_RECT) {
        const rt_rect &q = s.rects[idx];
        c.st->light_pdf_tests++;
        if (!rect_hit(q, Ray(o, v, 0.0), 0.001, rtm::INF, rec)) return 0.0;
        double area = (q.a1 - q.a0) * (q.b1 - q.b0);
        double dis_sqr = rec.t * rec.t * v.length_sqr();
        double cosv = rtm::fabs_(rtm::dot(v, rec.normal) / v.length());
        return dis_sqr / (cosv * area);
    }
    return 0.0;
}
Vec3 light_random(Ctx &c, uint32_t ref, Point3 o) {
    const rt_scene_desc &s = *c.s;
    if (ref & RT_REF_FLIP) return Vec3(1.0, 0.0, 0.0);
    uint32_t kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
    if (kind == RT_KIND_SPHERE) {
        const rt_sphere &q = s.spheres[idx];
        Vec3 direction = v3(q.center) - o;
        double dis_sqr = direction.length_sqr();
        rtm::Onb uvw = rtm::onb_from_w(direction);
        return uvw.local_vec(random_to_sphere(*c.rng, q.radius, dis_sqr));
    }
    if (kind == RT_KIND_RECT) {
        const rt_rect &q = s.rects[idx];
        double a = c.rng->gen_range(q.a0, q.a1);
        double b = c.rng->gen_range(q.b0, q.b1);
        Point3 random_point;
        switch (q.axis) {
            case RT_RECT_XY: random_point = Point3(a, b, q.k); break;
            case RT_RECT_XZ: random_point = Point3(a, q.k, b); break;
            default: random_point = Point3(q.k, a, b); break;
        }
        return random_point - o;
    }
    return Vec3(1.0, 0.0, 0.0);
}
// HittableList::pdf_value / random, hittable/mod.rs:121-132.
double lights_pdf_value(Ctx &c, Point3 o, Vec3 v) {
    uint32_t len = c.s->n_lights;
    double sum = 0.0;
    for (uint32_t i = 0; i < len; i++) sum += light_pdf_value(c, c.s->lights[i], o, v);
    return sum / (double)len;
}
Vec3 lights_random(Ctx &c, Point3 o) {
    uint64_t target = c.rng->gen_index(c.s->n_lights);
    return light_random(c, c.s->lights[target], o);
}

// ---- CosPdf, basic/pdf.rs:29-54 -------------------------------------------------
double cos_pdf_value(const rtm::Onb &uvw, Vec3 direction) {
    double cosv = rtm::dot(rtm::to_unit(direction), uvw.w);
    return cosv <= 0.0 ? 0.0 : cosv / rtm::PI;
}

// ---- Material::scatter / scattering_pdf / emitted, material/mod.rs --------------
struct ScatterRecord {                                           // mod.rs:216-231
    bool has_specular = false;
    Ray specular_ray;
    Color attenuation;
    bool has_pdf = false;
    rtm::Onb cos_uvw;
};
double reflectance(double cosv, double ref_idx) {                // mod.rs:112-116
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    double x = 1.0 - cosv;
    double x2 = x * x;
    return r0 + (1.0 - r0) * (x2 * x2 * x);                      // powi(5)
}
bool scatter(Ctx &c, const rt_material &m, const Ray &r_in, const HitRecord &rec, ScatterRecord &srec) {
    switch (m.kind) {
        case RT_MAT_LAMBERTIAN:                                  // mod.rs:51-57
            srec.has_specular = false;
            srec.attenuation = texture_value(*c.s, m.tex, rec.u, rec.v, rec.p);
            srec.has_pdf = true;
            srec.cos_uvw = rtm::onb_from_w(rec.normal);
            return true;
        case RT_MAT_METAL: {                                     // mod.rs:85-96
            Vec3 reflected = rtm::reflect(rtm::to_unit(r_in.dir), rec.normal);
            srec.has_specular = true;
            srec.specular_ray = Ray(rec.p, reflected + random_in_unit_sphere(*c.rng) * m.param, 0.0);
            srec.attenuation = v3(m.albedo);
            srec.has_pdf = false;
            return true;
        }
        case RT_MAT_DIELECTRIC: {                                // mod.rs:120-147
            double refraction_ratio = rec.front_face ? 1.0 / m.param : m.param;
            Vec3 unit_direction = rtm::to_unit(r_in.dir);
            double cos_theta = rtm::fmin_(rtm::dot(-unit_direction, rec.normal), 1.0);
            double sin_theta = rtm::sqrt_(1.0 - cos_theta * cos_theta);
            bool cannot_refract = refraction_ratio * sin_theta > 1.0;
            double random_double = c.rng->gen_range(0.0, 1.0);
            Vec3 direction = (cannot_refract || reflectance(cos_theta, refraction_ratio) > random_double)
                                 ? rtm::reflect(unit_direction, rec.normal)
                                 : rtm::refract(unit_direction, rec.normal, refraction_ratio);
            srec.has_specular = true;
            srec.specular_ray = Ray(rec.p, direction, r_in.tm);
            srec.attenuation = Color(1.0, 1.0, 1.0);
            srec.has_pdf = false;
            return true;
        }
        case RT_MAT_ISOTROPIC: {                                 // mod.rs:207-213
            srec.has_specular = true;
            srec.specular_ray = Ray(rec.p, random_in_unit_sphere(*c.rng), r_in.tm);
            srec.attenuation = texture_value(*c.s, m.tex, rec.u, rec.v, rec.p);
            srec.has_pdf = false;
            return true;
        }
        default: return false;                                   // DiffuseLight: trait default, mod.rs:16-18
    }
}
double scattering_pdf(const rt_material &m, const HitRecord &rec, const Ray &scattered) {
    if (m.kind != RT_MAT_LAMBERTIAN) return 0.0;                 // mod.rs:19-21
    double cosine = rtm::dot(rec.normal, rtm::to_unit(scattered.dir));   // mod.rs:58-65
    return cosine < 0.0 ? 0.0 : cosine / rtm::PI;
}
Color emitted(Ctx &c, const rt_material &m, const HitRecord &rec) {
    if (m.kind != RT_MAT_DIFFUSE_LIGHT) return Color(0.0, 0.0, 0.0);    // mod.rs:22-24
    if (rec.front_face) return texture_value(*c.s, m.tex, rec.u, rec.v, rec.p);   // mod.rs:174-180
    return Color(0.0, 0.0, 0.0);
}

// ---- ray_color, main.rs:233-278 --------------------------------------------------
Color ray_color(Ctx &c, const Ray &r, Color background, double t_min, int depth) {
    if (depth <= 0) return Color(0.0, 0.0, 0.0);
    c.st->rays++;
    HitRecord rec;
    if (!hit(c, c.s->root, r, t_min, rtm::F64_MAX, rec)) return background;
    const rt_material &m = c.s->materials[rec.mat];
    Color emit = emitted(c, m, rec);
    ScatterRecord srec;
    if (!scatter(c, m, r, rec, srec)) return emit;
    if (srec.has_specular)
        return srec.attenuation * ray_color(c, srec.specular_ray, background, t_min, depth - 1);
    Vec3 dir;
    double pdf_val;
    if (c.s->n_lights == 0) {
        // Build decision: the reference panics on an empty light
        // list; here the mixture degenerates to the cosine pdf alone.
        dir = srec.cos_uvw.local_vec(random_cosine_direction(*c.rng));
        pdf_val = cos_pdf_value(srec.cos_uvw, dir);
    } else {
        // MixturePdf(HittablePdf(lights, rec.p), cos): pdf.rs:94-104.
        if (c.rng->gen_range(0.0, 1.0) < 0.5) dir = lights_random(c, rec.p);
        else dir = srec.cos_uvw.local_vec(random_cosine_direction(*c.rng));
        pdf_val = 0.5 * lights_pdf_value(c, rec.p, dir) + 0.5 * cos_pdf_value(srec.cos_uvw, dir);
    }
    Ray scattered(rec.p, dir, r.tm);
    double spdf = scattering_pdf(m, rec, scattered);
    return emit + ((srec.attenuation * spdf) * ray_color(c, scattered, background, t_min, depth - 1)) / pdf_val;
}

// ---- Camera::get_ray, basic/camera.rs:64-73 --------------------------------------
Ray get_ray(const rt_camera &cam, double s, double t, Rng &rng) {
    Vec3 rd = random_in_unit_disk(rng) * cam.lens_radius;
    Vec3 offset = v3(cam.u) * rd.x + v3(cam.v) * rd.y;
    Vec3 origin = v3(cam.origin);
    Vec3 orig = origin + offset;
    Vec3 dir = v3(cam.lower_left_corner) + v3(cam.horizontal) * s + v3(cam.vertical) * t - origin - offset;
    double tm = rng.gen_range(cam.time0, cam.time1);
    return Ray(orig, dir, tm);
}

// ---- one pixel: main.rs:141-152 ---------------------------------------------------
void render_pixel(const rt_scene_desc &s, const rt_camera &cam, const rt_params &p, uint32_t g, uint32_t x,
                  rt_stats &st, double out[3]) {
    uint32_t frame = g / p.height, y = g % p.height;
    uint64_t pixel = (uint64_t)y * p.width + x;
    Color background = v3(p.background);
    uint32_t chunk = (p.spp_chunk == 0 || p.spp_chunk > p.spp) ? p.spp : p.spp_chunk;
    Color total(0.0, 0.0, 0.0);
    for (uint32_t s0 = 0; s0 < p.spp; s0 += chunk) {
        Color pixel_color(0.0, 0.0, 0.0);
        uint32_t s1 = s0 + chunk < p.spp ? s0 + chunk : p.spp;
        for (uint32_t smp = s0; smp < s1; smp++) {
            Rng rng(rtm::path_key(p.seed, frame, pixel, smp));
            Ctx c{&s, &rng, &st};
            double rand_u = rng.gen_f64();
            double rand_v = rng.gen_f64();
            double u = ((double)x + rand_u) / (double)(p.width - 1);
            double v = ((double)y + rand_v) / (double)(p.height - 1);
            Ray r = get_ray(cam, u, v, rng);
            st.paths++;
            pixel_color += ray_color(c, r, background, p.t_min, (int)p.max_depth);
            st.rng_draws += rng.draws;
        }
        total += pixel_color;       // one chunk: 0 + x == x, the reference's plain running sum
    }
    out[0] = total.x; out[1] = total.y; out[2] = total.z;
}

int validate(const rt_scene_desc *s, const rt_camera *cam, const rt_params *p) {
    if (!s || !cam || !p) { g_err = "null argument"; return RT_ERR_INVALID; }
    if (s->abi_version != RT2022_ABI_VERSION) { g_err = "abi_version mismatch"; return RT_ERR_INVALID; }
    if (p->width == 0 || p->height == 0 || p->n_frames == 0) { g_err = "empty image"; return RT_ERR_INVALID; }
    if (!(cam->time0 < cam->time1)) { g_err = "camera time0 >= time1 (gen_range panics, camera.rs:71)"; return RT_ERR_INVALID; }
    if (p->n_rows && !p->row_ids) { g_err = "row_ids is null"; return RT_ERR_INVALID; }
    for (uint32_t i = 0; i < p->n_rows; i++)
        if (p->row_ids[i] >= (uint64_t)p->height * p->n_frames) { g_err = "row id out of range"; return RT_ERR_INVALID; }
    return RT_OK;
}

void add_stats(rt_stats &a, const rt_stats &b) {
    a.paths += b.paths; a.rays += b.rays; a.node_visits += b.node_visits;
    for (int k = 0; k < RT_KIND_COUNT; k++) a.prim_tests[k] += b.prim_tests[k];
    a.light_pdf_tests += b.light_pdf_tests; a.rng_draws += b.rng_draws;
}

} // namespace

extern "C" {

const char *rto_last_error(void) { return g_err.c_str(); }

// The reference's threading scheme (main.rs:109-162): T workers, worker i takes the
// contiguous section [i*n/T, (i+1)*n/T) of the row list, the last one the remainder.
int rt_render_cpu(const rt_scene_desc *scene, const rt_camera *cam, const rt_params *params,
                  double *out_rgb_sum, rt_stats *stats, int n_threads) {
    int rc = validate(scene, cam, params);
    if (rc != RT_OK) return rc;
    if (!out_rgb_sum && params->n_rows) { g_err = "out_rgb_sum is null"; return RT_ERR_INVALID; }
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > params->n_rows && params->n_rows > 0) n_threads = (int)params->n_rows;
    std::vector<rt_stats> per(n_threads);
    for (auto &s : per) std::memset(&s, 0, sizeof s);
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int tid) {
        uint32_t section = params->n_rows / (uint32_t)n_threads;
        uint32_t beg = (uint32_t)tid * section;
        uint32_t end = tid == n_threads - 1 ? params->n_rows : beg + section;
        for (uint32_t yi = beg; yi < end; yi++)
            for (uint32_t x = 0; x < params->width; x++)
                render_pixel(*scene, *cam, *params, params->row_ids[yi], x, per[tid],
                             out_rgb_sum + ((size_t)yi * params->width + x) * 3);
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; t++) pool.emplace_back(work, t);
        for (auto &t : pool) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (auto &s : per) add_stats(*stats, s);
        stats->ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    }
    return RT_OK;
}

// write_color, main.rs:280-299.
void rto_write_color(const double rgb_sum[3], int32_t spp, uint8_t out[3]) {
    for (int i = 0; i < 3; i++) {
        double c = rgb_sum[i];
        if (c != c) c = 0.0;
        double v = rtm::floor_(rtm::clamp_(rtm::sqrt_(c / (double)spp), 0.0, 0.999) * 255.999);
        out[i] = (uint8_t)v;
    }
}

// ------------------------------------------------------------- test hooks ----
int rto_hit(const rt_scene_desc *scene, uint32_t ref, const double ray[7], double t_min, double t_max,
            uint64_t rng_state, rto_hit_record *out, rt_stats *stats) {
    rt_stats local; std::memset(&local, 0, sizeof local);
    Rng rng(rng_state);
    Ctx c{scene, &rng, stats ? stats : &local};
    HitRecord rec;
    Ray r(Vec3(ray[0], ray[1], ray[2]), Vec3(ray[3], ray[4], ray[5]), ray[6]);
    bool h = hit(c, ref, r, t_min, t_max, rec);
    if (out) {
        out->hit = h ? 1 : 0;
        out->p[0] = rec.p.x; out->p[1] = rec.p.y; out->p[2] = rec.p.z;
        out->normal[0] = rec.normal.x; out->normal[1] = rec.normal.y; out->normal[2] = rec.normal.z;
        out->t = rec.t; out->u = rec.u; out->v = rec.v;
        out->front_face = rec.front_face ? 1 : 0;
        out->mat = rec.mat;
        out->rng_draws = rng.draws;
    }
    return RT_OK;
}

int rto_ray_color(const rt_scene_desc *scene, const double ray[7], const double background[3], double t_min,
                  int depth, uint64_t rng_state, double out_rgb[3], rt_stats *stats) {
    rt_stats local; std::memset(&local, 0, sizeof local);
    Rng rng(rng_state);
    Ctx c{scene, &rng, stats ? stats : &local};
    Ray r(Vec3(ray[0], ray[1], ray[2]), Vec3(ray[3], ray[4], ray[5]), ray[6]);
    Color col = ray_color(c, r, v3(background), t_min, depth);
    out_rgb[0] = col.x; out_rgb[1] = col.y; out_rgb[2] = col.z;
    (stats ? stats : &local)->rng_draws += rng.draws;
    return RT_OK;
}

int rto_get_ray(const rt_camera *cam, double s, double t, uint64_t rng_state, double out_ray[7]) {
    Rng rng(rng_state);
    Ray r = get_ray(*cam, s, t, rng);
    out_ray[0] = r.orig.x; out_ray[1] = r.orig.y; out_ray[2] = r.orig.z;
    out_ray[3] = r.dir.x; out_ray[4] = r.dir.y; out_ray[5] = r.dir.z;
    out_ray[6] = r.tm;
    return (int)rng.draws;
}

int rto_texture_value(const rt_scene_desc *scene, uint32_t tex, double u, double v, const double p[3], double out_rgb[3]) {
    Color c = texture_value(*scene, tex, u, v, v3(p));
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
    return RT_OK;
}

double rto_perlin_noise(const rt_perlin *pl, const double p[3]) { return perlin_noise(*pl, v3(p)); }
double rto_perlin_turb(const rt_perlin *pl, const double p[3], int depth) { return perlin_turb(*pl, v3(p), depth); }

double rto_lights_pdf_value(const rt_scene_desc *scene, const double o[3], const double v[3]) {
    rt_stats local; std::memset(&local, 0, sizeof local);
    Rng rng(0);
    Ctx c{scene, &rng, &local};
    return lights_pdf_value(c, v3(o), v3(v));
}
int rto_lights_random(const rt_scene_desc *scene, const double o[3], uint64_t rng_state, double out_dir[3]) {
    rt_stats local; std::memset(&local, 0, sizeof local);
    Rng rng(rng_state);
    Ctx c{scene, &rng, &local};
    Vec3 d = lights_random(c, v3(o));
    out_dir[0] = d.x; out_dir[1] = d.y; out_dir[2] = d.z;
    return (int)rng.draws;
}

// Material::scatter on a synthetic hit record. Returns 0 = absorbed (None),
// 1 = specular ray in out_ray, 2 = diffuse (cosine pdf around rec.normal).
int rto_scatter(const rt_scene_desc *scene, uint32_t mat, const double ray_in[7], const rto_hit_record *rec_in,
                uint64_t rng_state, double out_ray[7], double out_attenuation[3], double out_emitted[3]) {
    rt_stats local; std::memset(&local, 0, sizeof local);
    Rng rng(rng_state);
    Ctx c{scene, &rng, &local};
    HitRecord rec;
    rec.p = v3(rec_in->p); rec.normal = v3(rec_in->normal);
    rec.t = rec_in->t; rec.u = rec_in->u; rec.v = rec_in->v; rec.front_face = rec_in->front_face != 0; rec.mat = mat;
    Ray r(Vec3(ray_in[0], ray_in[1], ray_in[2]), Vec3(ray_in[3], ray_in[4], ray_in[5]), ray_in[6]);
    const rt_material &m = scene->materials[mat];
    Color e = emitted(c, m, rec);
    out_emitted[0] = e.x; out_emitted[1] = e.y; out_emitted[2] = e.z;
    ScatterRecord srec;
    if (!scatter(c, m, r, rec, srec)) return 0;
    out_attenuation[0] = srec.attenuation.x; out_attenuation[1] = srec.attenuation.y; out_attenuation[2] = srec.attenuation.z;
    if (srec.has_specular) {
        const Ray &s = srec.specular_ray;
        out_ray[0] = s.orig.x; out_ray[1] = s.orig.y; out_ray[2] = s.orig.z;
        out_ray[3] = s.dir.x; out_ray[4] = s.dir.y; out_ray[5] = s.dir.z; out_ray[6] = s.tm;
        return 1;
    }
    return 2;
}

// rt_math.h on the host, one value at a time (op codes in rt_oracle.h).
double rto_math(int op, double a, double b) {
    switch (op) {
        case RTO_SIN: return rtm::sin_(a);
        case RTO_COS: return rtm::cos_(a);
        case RTO_ACOS: return rtm::acos_(a);
        case RTO_ATAN2: return rtm::atan2_(a, b);
        case RTO_LOG: return rtm::log_(a);
        case RTO_SQRT: return rtm::sqrt_(a);
        case RTO_DIV: return a / b;
        default: return 0.0;
    }
}
// 1: this library was built with -DRTO_LIBM (the render path's transcendentals are the platform libm's).
int rto_uses_libm(void) { return om::kLibm; }
// 1: this library was built with -DRTO_OWN_MATH (oracle/rto_math.h instead of the product's rt_math.h: nothing shared).
int rto_uses_own_math(void) {
#ifdef RTO_OWN_MATH
    return 1;
#else
    return 0;
#endif
}
// The transcendentals AS THE RENDER PATH OF THIS BUILD CALLS THEM (rto_math above is always rt_math.h).
double rto_path_math(int op, double a, double b) {
    switch (op) {
        case RTO_SIN: return om::sin_(a);
        case RTO_COS: return om::cos_(a);
        case RTO_ACOS: return om::acos_(a);
        case RTO_ATAN2: return om::atan2_(a, b);
        case RTO_LOG: return om::log_(a);
        default: return rto_math(op, a, b);
    }
}
void rto_math_array(int op, const double *a, const double *b, double *out, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) out[i] = rto_math(op, a[i], b ? b[i] : 0.0);
}

// RNG known answers: n words / f64 / ranges from a given state.
void rto_rng_u64(uint64_t state, uint64_t *out, uint64_t n) { Rng r(state); for (uint64_t i = 0; i < n; i++) out[i] = r.next_u64(); }
void rto_rng_f64(uint64_t state, double *out, uint64_t n) { Rng r(state); for (uint64_t i = 0; i < n; i++) out[i] = r.gen_f64(); }
void rto_rng_range(uint64_t state, double lo, double hi, double *out, uint64_t n) { Rng r(state); for (uint64_t i = 0; i < n; i++) out[i] = r.gen_range(lo, hi); }
void rto_rng_index(uint64_t state, uint64_t bound, uint64_t *out, uint64_t n) { Rng r(state); for (uint64_t i = 0; i < n; i++) out[i] = r.gen_index(bound); }
uint64_t rto_path_key(uint64_t seed, uint32_t frame, uint64_t pixel, uint32_t sample) { return rtm::path_key(seed, frame, pixel, sample); }

} // extern "C"

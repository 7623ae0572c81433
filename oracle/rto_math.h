// rto_math.h — the oracle's OWN restatement of the small arithmetic the render path stands on.
//
// TEST INFRASTRUCTURE ONLY (like everything under oracle/).
//
// The default oracle build shares raytracer_2022_amd/csrc/rt_math.h with the product, which is what makes the HIP path and the
// oracle agree bit for bit — and what makes every HIP-vs-oracle test blind to a mistake IN that header ("common
// mode"). `make own` builds the same rt_oracle.cpp against THIS header instead (-DRTO_OWN_MATH, which implies -DRTO_LIBM: the
// five transcendentals come from the platform libm): a build that shares not one line of arithmetic with the product.
// tests/test_oracle_own_math.py holds the two against each other: the own-math build must equal the -DRTO_LIBM build (shared
// vectors / RNG / casts, libm transcendentals) BIT FOR BIT on every scene builder — so rt_math.h's vectors, reflect / refract,
// Onb, RNG conversions and casts are pinned by a second, separately written statement of the reference's lines — and the
// libm-vs-fdlibm step is the one tools/libm_sensitivity.py measures.
//
// Written from the reference (cited per item; /root/reference/raytracer/src) and from rand 0.8.5's published algorithms,
// deliberately in other forms than rt_math.h where a form is free (128-bit products, ldexp, array members, std:: calls), so
// that the two cannot share a slip of the pen. Same names and signatures: rt_oracle.cpp compiles against either.
#ifndef RTO_MATH_H
#define RTO_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#define RT_MAX_REJECT 128          // (the rejection loops' cut — unreachable, < 2^-128 — must be the same number as the product's for the same draws)

namespace rtm {

inline uint64_t d2u(double x) { uint64_t u; std::memcpy(&u, &x, sizeof u); return u; }
inline double u2d(uint64_t u) { double x; std::memcpy(&x, &u, sizeof x); return x; }

inline double sqrt_(double x) { return std::sqrt(x); }          // f64::sqrt — IEEE, correctly rounded
inline double fabs_(double x) { return std::fabs(x); }
inline double floor_(double x) { return std::floor(x); }
// f64::min / f64::max: the other operand when one is a NaN.
inline double fmin_(double a, double b) { return std::isnan(a) ? b : std::isnan(b) ? a : (a < b ? a : b); }
inline double fmax_(double a, double b) { return std::isnan(a) ? b : std::isnan(b) ? a : (a > b ? a : b); }
// f64::clamp (write_color, main.rs:285-287): a NaN stays a NaN.
inline double clamp_(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

constexpr double PI = 3.141592653589793;                         // std::f64::consts::PI
constexpr double E_ = 2.718281828459045;                         // std::f64::consts::E
constexpr double INF = std::numeric_limits<double>::infinity();
constexpr double F64_MAX = std::numeric_limits<double>::max();   // f64::MAX (main.rs:243)

// The transcendentals of this build are the platform libm's (the names exist because rto_math() serves them one at a time).
inline double sin_(double x) { return std::sin(x); }
inline double cos_(double x) { return std::cos(x); }
inline double acos_(double x) { return std::acos(x); }
inline double atan2_(double y, double x) { return std::atan2(y, x); }
inline double log_(double x) { return std::log(x); }

// basic/vec.rs:12-46,137-325 — three f64, component-wise operators, true division.
struct Vec3 {
    double x, y, z;
    Vec3() : x(0.0), y(0.0), z(0.0) {}
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { const double c[3] = {x, y, z}; return c[i]; }
    void set(int i, double v) { double *c[3] = {&x, &y, &z}; *c[i] = v; }
    double length_sqr() const { return x * x + y * y + z * z; }                  // vec.rs:36-38
    double length() const { return std::sqrt(length_sqr()); }                    // vec.rs:40-42
};
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3 &operator+=(Vec3 &a, Vec3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline Vec3 &operator*=(Vec3 &a, double s) { a.x *= s; a.y *= s; a.z *= s; return a; }
inline double dot(Vec3 l, Vec3 r) { return l.x * r.x + l.y * r.y + l.z * r.z; }                    // vec.rs:24-26
inline Vec3 cross(Vec3 l, Vec3 r) {                                                                  // vec.rs:28-34
    Vec3 o;
    o.x = l.y * r.z - l.z * r.y;
    o.y = l.z * r.x - l.x * r.z;
    o.z = l.x * r.y - l.y * r.x;
    return o;
}
inline Vec3 to_unit(Vec3 a) { const double len = a.length(); return a / len; }                      // vec.rs:44-46
inline Vec3 reflect(Vec3 v, Vec3 n) { const double vn = dot(v, n); return v - (n * vn) * 2.0; }   // vec.rs:119-121: v - n * dot(v, n) * 2.
inline Vec3 refract(Vec3 uv, Vec3 n, double etai_over_etat) {                                       // vec.rs:123-128
    const double cos_theta = fmin_(dot(-uv, n), 1.0);
    const Vec3 perp = (uv + n * cos_theta) * etai_over_etat;
    const double k = std::sqrt(std::fabs(1.0 - perp.length_sqr()));
    const Vec3 parallel = (-n) * k;
    return perp + parallel;
}

struct Ray {                                                                                         // basic/ray.rs:4-20
    Vec3 orig, dir;
    double tm;
    Ray() : tm(0.0) {}
    Ray(Vec3 o, Vec3 d, double t) : orig(o), dir(d), tm(t) {}
    Vec3 at(double t) const { return orig + dir * t; }
};

struct Onb {                                                                                         // basic/onb.rs:4-36
    Vec3 u, v, w;
    Vec3 local_vec(Vec3 a) const { return u * a.x + v * a.y + w * a.z; }                             // onb.rs:22-24
};
inline Onb onb_from_w(Vec3 n) {                                                                      // onb.rs:26-36
    Onb o;
    o.w = to_unit(n);
    const Vec3 a = std::fabs(o.w.x) > 0.9 ? Vec3(0.0, 1.0, 0.0) : Vec3(1.0, 0.0, 0.0);
    o.v = to_unit(cross(o.w, a));
    o.u = cross(o.w, o.v);
    return o;
}

// RNG. The reference draws from rand::thread_rng() (unseedable); the build's decision is a
// SplitMix64 stream per (seed, frame, pixel, sample) — the constants below are that decision's and Steele / Lea / Flood's
// published ones — with rand 0.8.5's conversions (Standard for f64, UniformFloat / UniformInt::sample_single) restated.
inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
struct Rng {
    uint64_t s;
    uint32_t draws;
    Rng() : s(0), draws(0) {}
    explicit Rng(uint64_t state) : s(state), draws(0) {}
    uint64_t next_u64() { draws += 1; s += 0x9E3779B97F4A7C15ull; return mix64(s); }
    uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
    // Standard: 53 random bits scaled into [0, 1).
    double gen_f64() { return std::ldexp((double)(next_u64() >> 11), -53); }
    // UniformFloat<f64>::sample_single(low, high): 52 random bits as a float in [1, 2), minus one, times the scale, plus low; drawn again if that rounds to `high`.
    double gen_range(double low, double high) {
        const double scale = high - low;
        double res = low;
        for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
            const uint64_t fraction = next_u64() >> 12;
            const double value0_1 = u2d((uint64_t)1023 << 52 | fraction) - 1.0;
            res = value0_1 * scale + low;
            if (res < high) break;
        }
        return res;
    }
    // UniformInt<usize>::sample_single(0, n): widening multiply, accept when the low half falls in the zone.
    uint64_t gen_index(uint64_t n) {
        const uint64_t zone = (n << __builtin_clzll(n)) - 1;
        uint64_t hi = 0;
        for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
            const unsigned __int128 wide = (unsigned __int128)next_u64() * n;
            hi = (uint64_t)(wide >> 64);
            if ((uint64_t)wide <= zone) break;
        }
        return hi;
    }
    uint32_t gen_index_u32(uint32_t n) {                       // the same on u32 (main.rs:97)
        const uint32_t zone = (n << __builtin_clz(n)) - 1;
        uint32_t hi = 0;
        for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
            const uint64_t wide = (uint64_t)next_u32() * n;
            hi = (uint32_t)(wide >> 32);
            if ((uint32_t)wide <= zone) break;
        }
        return hi;
    }
};
// Stream key of one path: (seed, frame, absolute pixel index, sample) — the build's keying.
inline uint64_t path_key(uint64_t seed, uint32_t frame, uint64_t pixel, uint32_t sample) {
    uint64_t h = mix64(seed + ((uint64_t)frame + 1) * 0x9E3779B97F4A7C15ull);
    h = mix64(h ^ ((pixel + 1) * 0xD1B54A32D192ED03ull));
    h = mix64(h ^ (((uint64_t)sample + 1) * 0x8CB92BA72F3D8DD7ull));
    return h;
}

// Rust `as`: float to integer saturates, NaN gives 0 (perlin.rs: `as i32`; texture/mod.rs: `as usize`).
inline int32_t f64_as_i32(double x) {
    if (std::isnan(x)) return 0;
    if (x <= (double)std::numeric_limits<int32_t>::min()) return std::numeric_limits<int32_t>::min();
    if (x >= (double)std::numeric_limits<int32_t>::max()) return std::numeric_limits<int32_t>::max();
    return (int32_t)std::trunc(x);
}
inline uint64_t f64_as_usize(double x) {
    if (std::isnan(x) || x <= 0.0) return 0;
    if (x >= 18446744073709551616.0) return std::numeric_limits<uint64_t>::max();
    return (uint64_t)std::trunc(x);
}

} // namespace rtm
#endif

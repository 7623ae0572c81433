// rt_math.h — f64 arithmetic shared by the HIP kernels, the host scene layer and
// the CPU oracle, written so that host (g++) and device (hipcc, gfx950) produce
// bit-identical results:
//   * only + - * / sqrt floor and comparisons, which are IEEE-exact on both sides
//     when compiled with -ffp-contract=off (never -ffast-math);
//   * own sin / cos / acos / atan2 / log (libm and OCML differ in the last ulp,
//     and one flipped branch in a chaotic path moves a pixel by ~1e-3);
//   * the path RNG and the u64 -> f64 / range conversions of rand 0.8.5.
//
// Reference lines mirrored here: raytracer/src/basic/vec.rs:24-128 (Vec3 ops,
// reflect/refract), basic/ray.rs:18-20, basic/onb.rs:26-36.
//
// The transcendental kernels restate the published fdlibm / FreeBSD msun
// algorithms (k_sin.c, k_cos.c, e_rem_pio2.c medium path, e_acos.c, s_atan.c,
// e_atan2.c, e_log.c) with their coefficient tables; accuracy < 1 ulp in the
// ranges the path uses, checked against libm in tests/test_math.py. Those
// files carry this notice, which is preserved here as it asks:
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
//
// (third-party public code, not part of the reference repository)
#ifndef RT2022_RT_MATH_H
#define RT2022_RT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#define RT_HD_NOINLINE __host__ __device__ inline
#else
#define RT_HD inline
#define RT_HD_NOINLINE inline
#endif

namespace rtm {

// ------------------------------------------------------------------ bits ---
RT_HD uint64_t d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
RT_HD double u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
RT_HD int32_t hi32(double x) { return (int32_t)(d2u(x) >> 32); }
RT_HD uint32_t lo32(double x) { return (uint32_t)d2u(x); }
RT_HD double with_hi(double x, int32_t hi) { return u2d(((uint64_t)(uint32_t)hi << 32) | (d2u(x) & 0xFFFFFFFFull)); }
RT_HD double make_d(int32_t hi, uint32_t lo) { return u2d(((uint64_t)(uint32_t)hi << 32) | lo); }

RT_HD double sqrt_(double x) { return __builtin_sqrt(x); }
RT_HD double fabs_(double x) { return __builtin_fabs(x); }
RT_HD double floor_(double x) { return __builtin_floor(x); }
RT_HD bool isnan_(double x) { return x != x; }

// Rust f64::min / f64::max: a NaN operand is ignored.
RT_HD double fmin_(double a, double b) { if (a != a) return b; if (b != b) return a; return a < b ? a : b; }
RT_HD double fmax_(double a, double b) { if (a != a) return b; if (b != b) return a; return a > b ? a : b; }
// Rust f64::clamp: NaN stays NaN.
RT_HD double clamp_(double x, double lo, double hi) { if (x < lo) return lo; if (x > hi) return hi; return x; }

constexpr double PI = 3.14159265358979323846;      // std::f64::consts::PI
constexpr double E_ = 2.71828182845904523536;      // std::f64::consts::E
constexpr double INF = __builtin_huge_val();
constexpr double F64_MAX = 1.7976931348623157e308; // f64::MAX (main.rs:243)

// ----------------------------------------------------------- sin / cos -----
namespace detail {
constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

RT_HD double k_sin(double x, double y) {
    double z = x * x;
    double w = z * z;
    double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
RT_HD double k_cos(double x, double y) {
    double z = x * x;
    double w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

constexpr double INVPIO2 = 6.36619772367581382433e-01;
constexpr double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
constexpr double PIO2_2 = 6.07710050630396597660e-11, PIO2_2T = 2.02226624879595063154e-21;
constexpr double PIO2_3 = 2.02226624871116645580e-21, PIO2_3T = 8.47842766036889956997e-32;
// pi/2 in four 53-bit pieces, for the large-argument path.
constexpr double P1 = 0x1.921fb54442d18p+0, P2 = 0x1.1a62633145c07p-54,
                 P3 = -0x1.f1976b7ed8fbcp-110, P4 = 0x1.4cf98e804177dp-164;

// x -> (n mod 4, y0 + y1) with x = n*pi/2 + y0 + y1, |y0| <= ~pi/4.
RT_HD int rem_pio2(double x, double &y0, double &y1) {
    int32_t hx = hi32(x);
    int32_t ix = hx & 0x7fffffff;
    if (ix <= 0x3fe921fb) { y0 = x; y1 = 0.0; return 0; }          // |x| <= pi/4
    if (ix < 0x413921fb) {                                          // |x| < 2^20 * pi/2
        double fn = (x * INVPIO2 + 0x1.8p52) - 0x1.8p52;
        int n = (int)fn;
        double r = x - fn * PIO2_1;
        double w = fn * PIO2_1T;
        int j = ix >> 20;
        y0 = r - w;
        int i = j - ((hi32(y0) >> 20) & 0x7ff);
        if (i > 16) {
            double t = r;
            w = fn * PIO2_2;
            r = t - w;
            w = fn * PIO2_2T - ((t - r) - w);
            y0 = r - w;
            i = j - ((hi32(y0) >> 20) & 0x7ff);
            if (i > 49) {
                t = r;
                w = fn * PIO2_3;
                r = t - w;
                w = fn * PIO2_3T - ((t - r) - w);
                y0 = r - w;
            }
        }
        y1 = (r - y0) - w;
        return n & 3;
    }
    if (ix >= 0x7ff00000) { y0 = x - x; y1 = 0.0; return 0; }       // inf / NaN -> NaN
    if (ix >= 0x43300000) { y0 = 0.0; y1 = 0.0; return 0; }         // |x| >= 2^52: no fraction left worth having
    // 2^20*pi/2 <= |x| < 2^52: fused Cody-Waite over four pieces of pi/2.
    double fn = (x * INVPIO2 + 0x1.8p52) - 0x1.8p52;
    double r = __builtin_fma(-fn, P1, x);
    double r2 = __builtin_fma(-fn, P2, r);
    double r3 = __builtin_fma(-fn, P3, r2);
    double r4 = __builtin_fma(-fn, P4, r3);
    y0 = r4; y1 = 0.0;
    return (int)((int64_t)fn & 3);
}
} // namespace detail

RT_HD_NOINLINE double sin_(double x) {
    double y0, y1;
    int n = detail::rem_pio2(x, y0, y1);
    switch (n) {
        case 0: return detail::k_sin(y0, y1);
        case 1: return detail::k_cos(y0, y1);
        case 2: return -detail::k_sin(y0, y1);
        default: return -detail::k_cos(y0, y1);
    }
}
RT_HD_NOINLINE double cos_(double x) {
    double y0, y1;
    int n = detail::rem_pio2(x, y0, y1);
    switch (n) {
        case 0: return detail::k_cos(y0, y1);
        case 1: return -detail::k_sin(y0, y1);
        case 2: return -detail::k_cos(y0, y1);
        default: return detail::k_sin(y0, y1);
    }
}
// Both at once (same values as sin_/cos_ separately).
RT_HD_NOINLINE void sincos_(double x, double &s, double &c) {
    double y0, y1;
    int n = detail::rem_pio2(x, y0, y1);
    double ks = detail::k_sin(y0, y1), kc = detail::k_cos(y0, y1);
    switch (n) {
        case 0: s = ks; c = kc; break;
        case 1: s = kc; c = -ks; break;
        case 2: s = -ks; c = -kc; break;
        default: s = -kc; c = ks; break;
    }
}

// ---------------------------------------------------------------- acos -----
RT_HD_NOINLINE double acos_(double x) {
    constexpr double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                     pi = 3.14159265358979311600e+00;
    constexpr double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                     pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                     pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                     qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                     qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    int32_t hx = hi32(x);
    int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x3ff00000) {                       // |x| >= 1
        if (((uint32_t)(ix - 0x3ff00000) | lo32(x)) == 0) {
            if (hx > 0) return 0.0;
            return pi + 2.0 * pio2_lo;
        }
        return (x - x) / (x - x);                 // NaN
    }
    if (ix < 0x3fe00000) {                        // |x| < 0.5
        if (ix <= 0x3c600000) return pio2_hi + pio2_lo;
        double z = x * x;
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx < 0) {                          // x < -0.5
        double z = (1.0 + x) * 0.5;
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double s = sqrt_(z);
        double r = p / q;
        double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    } else {                                      // x > 0.5
        double z = (1.0 - x) * 0.5;
        double s = sqrt_(z);
        double df = u2d(d2u(s) & 0xFFFFFFFF00000000ull);
        double c = (z - df * df) / (s + df);
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double r = p / q;
        double w = r * s + c;
        return 2.0 * (df + w);
    }
}

// --------------------------------------------------------- atan / atan2 ----
RT_HD_NOINLINE double atan_(double x) {
    constexpr double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01,
                     hi2 = 9.82793723247329054082e-01, hi3 = 1.57079632679489655800e+00;
    constexpr double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17,
                     lo2 = 1.39033110312309984516e-17, lo3 = 6.12323399573676603587e-17;
    constexpr double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                     aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                     aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                     aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                     aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                     aT10 = 1.62858201153657823623e-02;
    int32_t hx = hi32(x);
    int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x44100000) {                       // |x| >= 2^66
        if (ix > 0x7ff00000 || (ix == 0x7ff00000 && lo32(x) != 0)) return x + x;   // NaN
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    int id;
    if (ix < 0x3fdc0000) {                        // |x| < 0.4375
        if (ix < 0x3e400000) return x;            // |x| < 2^-27
        id = -1;
    } else {
        x = fabs_(x);
        if (ix < 0x3ff30000) {                    // |x| < 1.1875
            if (ix < 0x3fe60000) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
            else { id = 1; x = (x - 1.0) / (x + 1.0); }
        } else {
            if (ix < 0x40038000) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
            else { id = 3; x = -1.0 / x; }
        }
    }
    double z = x * x;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    double ahi = id == 0 ? hi0 : id == 1 ? hi1 : id == 2 ? hi2 : hi3;
    double alo = id == 0 ? lo0 : id == 1 ? lo1 : id == 2 ? lo2 : lo3;
    z = ahi - ((x * (s1 + s2) - alo) - x);
    return hx < 0 ? -z : z;
}

RT_HD_NOINLINE double atan2_(double y, double x) {
    constexpr double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00,
                     pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    int32_t hx = hi32(x), hy = hi32(y);
    uint32_t lx = lo32(x), ly = lo32(y);
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (x != x || y != y) return x + y;
    if (hx == 0x3ff00000 && lx == 0) return atan_(y);           // x == 1.0
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                // 2*sign(x) + sign(y)
    if ((iy | (int32_t)ly) == 0) {                              // y == 0
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if ((ix | (int32_t)lx) == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;   // x == 0
    if (ix == 0x7ff00000) {                                     // x inf
        if (iy == 0x7ff00000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0 * pi_o_4 + tiny;
                default: return -3.0 * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0;
                case 1: return -0.0;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7ff00000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int32_t k = (iy - ix) >> 20;
    double z;
    if (k > 60) { z = pi_o_2 + 0.5 * pi_lo; m &= 1; }
    else if (hx < 0 && k < -60) z = 0.0;
    else z = atan_(fabs_(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// ----------------------------------------------------------------- log -----
RT_HD_NOINLINE double log_(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                     two54 = 1.80143985094819840000e+16,
                     Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                     Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                     Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    int32_t hx = hi32(x);
    uint32_t lx = lo32(x);
    int32_t k = 0;
    if (hx < 0x00100000) {                                  // x < 2^-1022
        if (((hx & 0x7fffffff) | (int32_t)lx) == 0) return -INF;   // log(+-0) = -inf
        if (hx < 0) return (x - x) / (x - x);               // log(-#) = NaN
        k -= 54; x *= two54; hx = hi32(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = with_hi(x, hx | (i ^ 0x3ff00000));                  // normalise x or x/2
    k += (i >> 20);
    double f = x - 1.0;
    double dk;
    if ((0x000fffff & (2 + hx)) < 3) {                      // -2^-20 <= f < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    dk = (double)k;
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int32_t j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    } else {
        if (k == 0) return f - s * (f - R);
        return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
    }
}

// ---------------------------------------------------------------- Vec3 -----
// Operator forms follow basic/vec.rs:137-325 (component-wise, true division).
struct Vec3 {
    double x, y, z;
    RT_HD Vec3() : x(0.0), y(0.0), z(0.0) {}
    RT_HD Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    RT_HD double operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
    RT_HD void set(int i, double v) { if (i == 0) x = v; else if (i == 1) y = v; else z = v; }
    RT_HD double length_sqr() const { return x * x + y * y + z * z; }            // vec.rs:36-38
    RT_HD double length() const { return sqrt_(length_sqr()); }                  // vec.rs:40-42
};
RT_HD Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
RT_HD Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD Vec3 operator*(Vec3 a, double s) { return Vec3(a.x * s, a.y * s, a.z * s); }
RT_HD Vec3 operator/(Vec3 a, double s) { return Vec3(a.x / s, a.y / s, a.z / s); }
RT_HD Vec3 &operator+=(Vec3 &a, Vec3 b) { a = a + b; return a; }
RT_HD Vec3 &operator*=(Vec3 &a, double s) { a = a * s; return a; }
RT_HD double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                    // vec.rs:24-26
RT_HD Vec3 cross(Vec3 a, Vec3 b) {                                                                  // vec.rs:28-34
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_HD Vec3 to_unit(Vec3 a) { return a / a.length(); }                                               // vec.rs:44-46
RT_HD Vec3 reflect(Vec3 v, Vec3 n) { return v - n * dot(v, n) * 2.0; }                            // vec.rs:119-121
RT_HD Vec3 refract(Vec3 uv, Vec3 n, double etai_over_etat) {                                        // vec.rs:123-128
    double cos_theta = fmin_(dot(-uv, n), 1.0);
    Vec3 r_out_perp = (uv + n * cos_theta) * etai_over_etat;
    Vec3 r_out_parallel = -n * sqrt_(fabs_(1.0 - r_out_perp.length_sqr()));
    return r_out_perp + r_out_parallel;
}

struct Ray {                                                                                         // basic/ray.rs:4-20
    Vec3 orig, dir;
    double tm;
    RT_HD Ray() : tm(0.0) {}
    RT_HD Ray(Vec3 o, Vec3 d, double t) : orig(o), dir(d), tm(t) {}
    RT_HD Vec3 at(double t) const { return orig + dir * t; }
};

struct Onb {                                                                                         // basic/onb.rs:4-36
    Vec3 u, v, w;
    RT_HD Vec3 local_vec(Vec3 a) const { return u * a.x + v * a.y + w * a.z; }
};
RT_HD Onb onb_from_w(Vec3 n) {
    Onb o;
    o.w = to_unit(n);
    Vec3 a = fabs_(o.w.x) > 0.9 ? Vec3(0.0, 1.0, 0.0) : Vec3(1.0, 0.0, 0.0);
    o.v = to_unit(cross(o.w, a));
    o.u = cross(o.w, o.v);
    return o;
}

// ----------------------------------------------------------------- RNG -----
// The reference draws everything from rand::thread_rng() (ChaCha12, OS-seeded,
// unseedable). Build decision: one counter-based stream per
// (seed, frame, pixel, sample), SplitMix64 output function; draws inside a path
// are consumed in the reference's order. Conversions restate rand 0.8.5
// (Cargo.lock:371-383; source not in /root/reference — "parity unpinned").
RT_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

RT_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
    uint64_t a_lo = (uint32_t)a, a_hi = a >> 32, b_lo = (uint32_t)b, b_hi = b >> 32;
    uint64_t p0 = a_lo * b_lo, p1 = a_lo * b_hi, p2 = a_hi * b_lo, p3 = a_hi * b_hi;
    uint64_t mid = (p0 >> 32) + (uint32_t)p1 + (uint32_t)p2;
    return p3 + (p1 >> 32) + (p2 >> 32) + (mid >> 32);
}

// Every rejection loop on the path (range redraws, unit disk / sphere samplers) is
// cut after this many tries. Acceptance is >= 0.5 per try, so the cut is
// unreachable (< 2^-128) — it only guarantees that no GPU wave can spin forever.
#define RT_MAX_REJECT 128

struct Rng {
    uint64_t s;
    uint32_t draws;     // words drawn since seeding (statistics only)
    RT_HD Rng() : s(0), draws(0) {}
    RT_HD explicit Rng(uint64_t state) : s(state), draws(0) {}
    RT_HD uint64_t next_u64() {
        s += 0x9E3779B97F4A7C15ull;
        draws++;
        return mix64(s);
    }
    RT_HD uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
    // Standard: Rng::gen::<f64>() — 53 bits, [0,1).
    RT_HD double gen_f64() { return (double)(next_u64() >> 11) * 0x1.0p-53; }
    // UniformFloat::sample_single: gen_range(low..high), 52 bits, redraw if res >= high.
    // (The redraw fires only when rounding lands on `high`; the loop is bounded so
    // that a NaN bound cannot hang a GPU wave — rand would have panicked instead.)
    RT_HD double gen_range(double low, double high) {
        double scale = high - low;
        double res = low;
        for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
            double value1_2 = u2d((next_u64() >> 12) | 0x3FF0000000000000ull);
            double value0_1 = value1_2 - 1.0;
            res = value0_1 * scale + low;
            if (res < high) return res;
        }
        return res;
    }
    // UniformInt<usize>::sample_single: gen_range(0..n), widening multiply + zone.
    RT_HD uint64_t gen_index(uint64_t n) {
        uint64_t zone = (n << __builtin_clzll(n)) - 1;
        uint64_t hi = 0;
        for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
            uint64_t v = next_u64();
            uint64_t lo = v * n;
            hi = mulhi64(v, n);
            if (lo <= zone) return hi;
        }
        return hi;
    }
    // UniformInt<u32>::sample_single: gen_range(0..n) on u32 (main.rs:97).
    RT_HD uint32_t gen_index_u32(uint32_t n) {
        uint32_t zone = (n << __builtin_clz(n)) - 1;
        uint32_t hi = 0;
        for (int tries = 0; tries < RT_MAX_REJECT; tries++) {
            uint32_t v = next_u32();
            uint64_t m = (uint64_t)v * n;
            uint32_t lo = (uint32_t)m;
            hi = (uint32_t)(m >> 32);
            if (lo <= zone) return hi;
        }
        return hi;
    }
};

// Stream key of one path: (seed, frame, absolute pixel index y*W+x, sample).
RT_HD uint64_t path_key(uint64_t seed, uint32_t frame, uint64_t pixel, uint32_t sample) {
    uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ull * ((uint64_t)frame + 1));
    h = mix64(h ^ (0xD1B54A32D192ED03ull * (pixel + 1)));
    h = mix64(h ^ (0x8CB92BA72F3D8DD7ull * ((uint64_t)sample + 1)));
    return h;
}

// Float -> integer casts with Rust `as` semantics (saturating, NaN -> 0).
RT_HD int32_t f64_as_i32(double x) {
    if (x != x) return 0;
    if (x <= -2147483648.0) return (int32_t)0x80000000;
    if (x >= 2147483647.0) return 0x7fffffff;
    return (int32_t)x;
}
RT_HD uint64_t f64_as_usize(double x) {
    if (x != x || x <= 0.0) return 0;
    if (x >= 18446744073709551615.0) return 0xFFFFFFFFFFFFFFFFull;
    return (uint64_t)x;
}

} // namespace rtm
#endif

/* TEST INFRASTRUCTURE — CPU oracle arithmetic (plain C).
 *
 * The oracle's own copy of the "hanabi-math" arithmetic definition: IEEE binary32 for
 * + - * / sqrt, and binary64 polynomial kernels (no libm) for the transcendental WGSL
 * builtins, so that oracle and HIP kernels agree bit for bit. The definition is stated in
 * DESIGN.md; tests/test_math.py checks it against libm (<= 1 ulp) independently.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use anything under
 * oracle/; the product never includes this file.
 */
#ifndef HANABI_ORACLE_MATH_H
#define HANABI_ORACLE_MATH_H
#include <stdbool.h>
#include <stdint.h>

#define HNB_HD static inline


HNB_HD uint32_t f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
HNB_HD float u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
HNB_HD uint64_t d2u(double f) { uint64_t u; __builtin_memcpy(&u, &f, 8); return u; }
HNB_HD double u2d(uint64_t u) { double f; __builtin_memcpy(&f, &u, 8); return f; }

HNB_HD float f_nan() { return u2f(0x7fc00000u); }
HNB_HD float f_inf() { return u2f(0x7f800000u); }
HNB_HD bool f_isnan(float x) { return x != x; }
HNB_HD bool f_signbit(float x) { return (f2u(x) >> 31) != 0; }
HNB_HD bool d_signbit(double x) { return (d2u(x) >> 63) != 0; }

// ---- exact binary32 helpers ------------------------------------------------------
HNB_HD float f_abs(float x) { return u2f(f2u(x) & 0x7fffffffu); }
HNB_HD float f_sqrt(float x) { return __builtin_sqrtf(x); }
HNB_HD float f_floor(float x) { return __builtin_floorf(x); }
HNB_HD float f_ceil(float x) { return __builtin_ceilf(x); }
HNB_HD float f_trunc(float x) { return __builtin_truncf(x); }
HNB_HD float f_round_even(float x) { return __builtin_rintf(x); }
// WGSL: min(e1,e2) = e2 < e1 ? e2 : e1 ; max(e1,e2) = e1 < e2 ? e2 : e1
HNB_HD float f_min(float a, float b) { return (b < a) ? b : a; }
HNB_HD float f_max(float a, float b) { return (a < b) ? b : a; }
HNB_HD float f_clamp(float x, float lo, float hi) { return f_min(f_max(x, lo), hi); }
HNB_HD float f_saturate(float x) { return f_clamp(x, 0.0f, 1.0f); }
HNB_HD float f_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
HNB_HD float f_fract(float x) { return x - f_floor(x); }
HNB_HD float f_step(float edge, float x) { return (edge <= x) ? 1.0f : 0.0f; }
HNB_HD float f_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
HNB_HD float f_smoothstep(float lo, float hi, float x) {
    float t = f_clamp((x - lo) / (hi - lo), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
HNB_HD float f_rem(float x, float y) { return x - y * f_trunc(x / y); }
HNB_HD float f_inv_sqrt(float x) { return 1.0f / f_sqrt(x); }

// ---- binary64 kernels ------------------------------------------------------------
// Round to nearest integer (ties to even) with two IEEE additions; valid |x| < 2^51.
HNB_HD double d_rint(double x) {
    const double magic = 6755399441055744.0;  // 1.5 * 2^52
    double t = x + magic;  // never folded: built without fast-math / reassociation
    return t - magic;
}

// sin and cos of a finite double with |x| <= 2^40.
HNB_HD void d_sincos(double x, double* s_out, double* c_out) {
    const double two_over_pi = 0.63661977236758134308;
    const double p1 = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double p2 = 6.07710050630396597660e-11;  // next 33 bits
    const double p3 = 2.02226624879595063154e-21;  // remainder
    double k = d_rint(x * two_over_pi);
    double r = ((x - k * p1) - k * p2) - k * p3;
    long long q = (long long)k;
    double z = r * r;
    // Taylor series, |r| <= pi/4 (+ slack): truncation < 1e-18
    double ps = -1.0 / 355687428096000.0;                 // -1/17!
    ps = ps * z + 1.0 / 1307674368000.0;                  // 1/15!
    ps = ps * z - 1.0 / 6227020800.0;                     // -1/13!
    ps = ps * z + 1.0 / 39916800.0;                       // 1/11!
    ps = ps * z - 1.0 / 362880.0;                         // -1/9!
    ps = ps * z + 1.0 / 5040.0;                           // 1/7!
    ps = ps * z - 1.0 / 120.0;                            // -1/5!
    ps = ps * z + 1.0 / 6.0;                              // 1/3!
    double sn = r - (r * z) * ps;
    double pc = 1.0 / 6402373705728000.0;                 // 1/18!
    pc = pc * z - 1.0 / 20922789888000.0;                 // -1/16!
    pc = pc * z + 1.0 / 87178291200.0;                    // 1/14!
    pc = pc * z - 1.0 / 479001600.0;                      // -1/12!
    pc = pc * z + 1.0 / 3628800.0;                        // 1/10!
    pc = pc * z - 1.0 / 40320.0;                          // -1/8!
    pc = pc * z + 1.0 / 720.0;                            // 1/6!
    pc = pc * z - 1.0 / 24.0;                             // -1/4!
    pc = pc * z + 0.5;                                    // 1/2!
    double cs = 1.0 - z * pc;
    switch ((int)(q & 3)) {
        case 0: *s_out = sn; *c_out = cs; break;
        case 1: *s_out = cs; *c_out = -sn; break;
        case 2: *s_out = -sn; *c_out = -cs; break;
        default: *s_out = -cs; *c_out = sn; break;
    }
}

// 2^k for integer k in [-1022, 1023]
HNB_HD double d_pow2i(long long k) { return u2d((uint64_t)(k + 1023) << 52); }

// exp(x) for finite x; caller clamps to [-120, 100]
HNB_HD double d_exp(double x) {
    const double log2e = 1.44269504088896338700;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    double k = d_rint(x * log2e);
    double r = (x - k * ln2_hi) - k * ln2_lo;
    double p = 1.0 / 6227020800.0;      // 1/13!
    p = p * r + 1.0 / 479001600.0;      // 1/12!
    p = p * r + 1.0 / 39916800.0;       // 1/11!
    p = p * r + 1.0 / 3628800.0;        // 1/10!
    p = p * r + 1.0 / 362880.0;         // 1/9!
    p = p * r + 1.0 / 40320.0;          // 1/8!
    p = p * r + 1.0 / 5040.0;           // 1/7!
    p = p * r + 1.0 / 720.0;            // 1/6!
    p = p * r + 1.0 / 120.0;            // 1/5!
    p = p * r + 1.0 / 24.0;             // 1/4!
    p = p * r + 1.0 / 6.0;              // 1/3!
    p = p * r + 0.5;                    // 1/2!
    p = p * r + 1.0;
    p = p * r + 1.0;
    return p * d_pow2i((long long)k);
}

// natural log of a finite, strictly positive, normal double
HNB_HD double d_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    uint64_t b = d2u(x);
    long long e = (long long)((b >> 52) & 2047) - 1023;
    double m = u2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    double f = (m - 1.0) / (m + 1.0);
    double z = f * f;
    double p = 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    double lm = 2.0 * f * p;
    double de = (double)e;
    return de * ln2_hi + (de * ln2_lo + lm);
}

// sqrt of a double in [0, 2^100] whose value is representable-ish in f32 range:
// f32 seed (IEEE sqrtf) + 2 Newton steps. Deterministic, ~1e-16 relative.
HNB_HD double d_sqrt(double a) {
    if (!(a > 0.0)) return (a == 0.0) ? 0.0 : (a - a) / (a - a);  // 0 -> 0, neg/NaN -> NaN
    double s = (double)f_sqrt((float)a);
    s = 0.5 * (s + a / s);
    s = 0.5 * (s + a / s);
    return s;
}

// atan of any double (NaN -> NaN)
HNB_HD double d_atan(double x) {
    if (x != x) return x;
    const double pi_2 = 1.57079632679489655800;
    const double pi_4 = 0.78539816339744827900;
    bool neg = d_signbit(x);
    double t = neg ? -x : x;
    bool inv = t > 1.0;
    if (inv) t = 1.0 / t;
    bool shift = t > 0.41421356237309503;  // tan(pi/8)
    if (shift) t = (t - 1.0) / (t + 1.0);
    double z = t * t;
    // sum_{n=0}^{23} (-1)^n z^n / (2n+1), |z| <= 0.1716: truncation < 1e-19
    double p = -1.0 / 47.0;
    p = p * z + 1.0 / 45.0;
    p = p * z - 1.0 / 43.0;
    p = p * z + 1.0 / 41.0;
    p = p * z - 1.0 / 39.0;
    p = p * z + 1.0 / 37.0;
    p = p * z - 1.0 / 35.0;
    p = p * z + 1.0 / 33.0;
    p = p * z - 1.0 / 31.0;
    p = p * z + 1.0 / 29.0;
    p = p * z - 1.0 / 27.0;
    p = p * z + 1.0 / 25.0;
    p = p * z - 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z - 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z - 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z - 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z - 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z - 1.0 / 3.0;
    p = p * z + 1.0;
    double r = t * p;
    if (shift) r = pi_4 + r;
    if (inv) r = pi_2 - r;
    return neg ? -r : r;
}

HNB_HD double d_atan2(double y, double x) {
    const double pi = 3.14159265358979311600;
    const double pi_2 = 1.57079632679489655800;
    if (x != x || y != y) return x + y;
    if (x > 0.0) return d_atan(y / x);
    if (x < 0.0) return d_atan(y / x) + (d_signbit(y) ? -pi : pi);
    if (y > 0.0) return pi_2;
    if (y < 0.0) return -pi_2;
    if (d_signbit(x)) return d_signbit(y) ? -pi : pi;
    return y;
}

// ---- binary32 entry points -----------------------------------------------------------
#ifdef ORACLE_LIBM
/* Independent flavour of the oracle (libhanabi_oracle_libm.so): the transcendental WGSL builtins are evaluated by the host's
 * libm in binary64 and rounded once to binary32 — no code shared with the product's hnb_math.h. The GPU parity tests compare
 * the product with this flavour within north_star's 1e-5 relative tolerance; the polynomial flavour below stays the
 * bit-exact checker for lists and counters. */
#include <math.h>
HNB_HD float f_sin(float x) { return (float)sin((double)x); }
HNB_HD float f_cos(float x) { return (float)cos((double)x); }
HNB_HD float f_tan(float x) { return (float)tan((double)x); }
HNB_HD float f_atan(float x) { return (float)atan((double)x); }
HNB_HD float f_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }
HNB_HD float f_asin(float x) { return (float)asin((double)x); }
HNB_HD float f_acos(float x) { return (float)acos((double)x); }
HNB_HD float f_exp(float x) { return (float)exp((double)x); }
HNB_HD float f_exp2(float x) { return (float)exp2((double)x); }
HNB_HD float f_log(float x) { return (float)log((double)x); }
HNB_HD float f_log2(float x) { return (float)log2((double)x); }
HNB_HD float f_pow(float x, float y) { return (float)pow((double)x, (double)y); }
#else
HNB_HD bool trig_in_range(float x) { return f_abs(x) <= 1099511627776.0f; }  // 2^40

HNB_HD float f_sin(float x) {
    if (!(f_abs(x) <= 3.4028234663852886e38f)) return x - x;  // NaN/inf -> NaN
    if (!trig_in_range(x)) return 0.0f;
    double s, c; d_sincos((double)x, &s, &c); return (float)s;
}
HNB_HD float f_cos(float x) {
    if (!(f_abs(x) <= 3.4028234663852886e38f)) return x - x;
    if (!trig_in_range(x)) return 1.0f;
    double s, c; d_sincos((double)x, &s, &c); return (float)c;
}
HNB_HD float f_tan(float x) {
    if (!(f_abs(x) <= 3.4028234663852886e38f)) return x - x;
    if (!trig_in_range(x)) return 0.0f;
    double s, c; d_sincos((double)x, &s, &c); return (float)(s / c);
}
HNB_HD float f_atan(float x) { return (float)d_atan((double)x); }
HNB_HD float f_atan2(float y, float x) { return (float)d_atan2((double)y, (double)x); }
HNB_HD float f_asin(float x) {
    double xd = (double)x;
    return (float)d_atan2(xd, d_sqrt((1.0 - xd) * (1.0 + xd)));
}
HNB_HD float f_acos(float x) {
    double xd = (double)x;
    return (float)d_atan2(d_sqrt((1.0 - xd) * (1.0 + xd)), xd);
}
HNB_HD float f_exp(float x) {
    if (x != x) return x;
    double xd = (double)x;
    if (xd > 100.0) xd = 100.0;
    if (xd < -120.0) xd = -120.0;
    return (float)d_exp(xd);
}
HNB_HD float f_exp2(float x) {
    if (x != x) return x;
    double xd = (double)x;
    if (xd > 140.0) xd = 140.0;
    if (xd < -170.0) xd = -170.0;
    double k = d_rint(xd);
    double r = (xd - k) * 0.69314718055994528623;
    return (float)(d_exp(r) * d_pow2i((long long)k));
}
HNB_HD double d_log_f(float x, bool* special, float* sv) {
    *special = true;
    if (x != x) { *sv = x; return 0.0; }
    if (x < 0.0f) { *sv = f_nan(); return 0.0; }
    if (x == 0.0f) { *sv = -f_inf(); return 0.0; }
    if (x == f_inf()) { *sv = x; return 0.0; }
    *special = false;
    return d_log((double)x);
}
HNB_HD float f_log(float x) {
    bool sp; float sv; double l = d_log_f(x, &sp, &sv);
    return sp ? sv : (float)l;
}
HNB_HD float f_log2(float x) {
    bool sp; float sv; double l = d_log_f(x, &sp, &sv);
    return sp ? sv : (float)(l * 1.44269504088896338700);
}
// WGSL pow(x, y): defined here as exp(y * ln x) for x > 0; x < 0 -> NaN.
HNB_HD float f_pow(float x, float y) {
    if (x != x || y != y) return x + y;
    if (y == 0.0f) return 1.0f;
    if (x < 0.0f) return f_nan();
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : f_inf();
    if (x == f_inf()) return (y > 0.0f) ? f_inf() : 0.0f;
    double t = (double)y * d_log((double)x);
    if (t > 100.0) t = 100.0;
    if (t < -120.0) t = -120.0;
    return (float)d_exp(t);
}

#endif  /* ORACLE_LIBM */

// ---- conversions (WGSL value constructors: truncate + saturate, NaN -> 0) -------------
HNB_HD int32_t f_to_i32(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)x;
}
HNB_HD uint32_t f_to_u32(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

// ---- integer division (WGSL: x/0 = x, x%0 = 0, INT_MIN/-1 = INT_MIN, INT_MIN%-1 = 0) ---
HNB_HD int32_t i_div(int32_t a, int32_t b) {
    if (b == 0) return a;
    if (a == (int32_t)0x80000000u && b == -1) return a;
    return a / b;
}
HNB_HD int32_t i_rem(int32_t a, int32_t b) {
    if (b == 0) return 0;
    if (a == (int32_t)0x80000000u && b == -1) return 0;
    return a % b;
}
HNB_HD uint32_t u_div(uint32_t a, uint32_t b) { return b == 0u ? a : a / b; }
HNB_HD uint32_t u_rem(uint32_t a, uint32_t b) { return b == 0u ? 0u : a % b; }

// ---- pack / unpack (WGSL spec formulas) ------------------------------------------------
HNB_HD uint32_t pack_unorm8(float c) { return (uint32_t)f_floor(0.5f + 255.0f * f_min(1.0f, f_max(0.0f, c))); }
HNB_HD uint32_t pack_snorm8(float c) {
    return (uint32_t)(int32_t)f_floor(0.5f + 127.0f * f_min(1.0f, f_max(-1.0f, c))) & 0xffu;
}
HNB_HD float unpack_unorm8(uint32_t b) { return (float)(b & 0xffu) / 255.0f; }
HNB_HD float unpack_snorm8(uint32_t b) {
    int32_t v = (int32_t)(int8_t)(b & 0xffu);
    return f_max((float)v / 127.0f, -1.0f);
}

// ---- PRNG: PCG-RXS-M-XS hash (reference src/render/vfx_common.wgsl:266-275) -------------
HNB_HD uint32_t pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
HNB_HD float to_float01(uint32_t u) { return u2f((u & 0x007fffffu) | 0x3f800000u) - 1.0f; }

#endif

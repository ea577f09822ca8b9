/* TEST INFRASTRUCTURE — CPU oracle arithmetic (plain C).
 *
 * The oracle's own copy of the "hanabi-math" arithmetic definition: IEEE binary32 for
 * + - * / sqrt, and binary32 polynomial kernels built from those and fmaf (no libm) for the
 * transcendental WGSL builtins, so that oracle and HIP kernels agree bit for bit. The
 * definition is stated in DESIGN.md; tests/test_math.py checks it against libm independently
 * (sampled), tools/math_sweep.c over all binary32 arguments (profiles/r04_math_sweep.txt).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use anything under
 * oracle/; the product never includes this file.
 */
#ifndef HANABI_ORACLE_MATH_H
#define HANABI_ORACLE_MATH_H
#include <stdbool.h>
#include <stdint.h>

#define HNB_HD static inline
#define HNB_TABLE static const


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

// ---- transcendental builtins ---------------------------------------------------------
// hanabi-math v3. The contract of the path is 1e-5 relative on positions and velocities (BASELINE.json north_star), not the last bit
// of libm: every builtin is a binary32 kernel whose every step is ONE correctly rounded IEEE-754 binary32 operation - + - * / sqrt or
// a fused multiply-add (v_fma_f32 on gfx950; vfmadd or the C library's exact fmaf() on the host: the same result everywhere, which
// -ffp-contract=off alone cannot promise for an a * b + c the compiler is free to fuse or not) - on polynomials whose coefficients
// tools/gen_math_coeffs.py derives and tools/gen_math_kernels.py writes into both copies of this section. Accuracy against the
// host's binary64 libm over ALL binary32 arguments (tools/math_sweep.c, profiles/r04_math_sweep.txt): sin cos asin atan atan2 <= 2 ulp,
// exp exp2 log log2 acos <= 1 ulp, tan <= 4 ulp, pow <= 8 ulp (<= 2 ulp for |y log2 x| <= 32). (Rounds 1-3 evaluated the same
// functions in binary64 and rounded once: <= 1 ulp, at half rate on this part - v_fma_f64 - and 3-4x the instructions: the burst
// inits of sphere- and cone-shaped spawns were bound by it.)
//
// The one binary64 piece left: sin / cos / tan of |x| > 65536, where a three-term binary32 Cody-Waite reduction runs out of bits.
// Such arguments take a (divergent, rarely entered) branch to the round-3 kernel: x - k pi/2 in binary64, minimax polynomials,
// one rounding to binary32; |x| > 2^40 is defined as x = 0 (sin 0, cos 1), NaN / inf -> NaN.
HNB_HD double d_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// Round to nearest integer (ties to even) with two IEEE additions; valid |x| < 2^51. *low32: that integer modulo 2^32 (the low
// mantissa bits of the biased sum), without a float -> int conversion.
HNB_HD double d_rint_bits(double x, uint32_t* low32) {
    const double magic = 6755399441055744.0;  // 1.5 * 2^52
    double t = x + magic;  // never folded: built without fast-math / reassociation
    *low32 = (uint32_t)d2u(t);
    return t - magic;
}
// sin and cos of a finite double with |x| <= 2^40.
HNB_HD void d_sincos(double x, double* s_out, double* c_out) {
    const double two_over_pi = 0x1.45f306dc9c883p-1;
    const double p1 = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double p2 = 6.07710050630396597660e-11;  // next 33 bits
    const double p3 = 2.02226624879595063154e-21;  // remainder
    uint32_t q;
    const double k = d_rint_bits(x * two_over_pi, &q);
    double r = d_fma(-k, p1, x);
    r = d_fma(-k, p2, r);
    r = d_fma(-k, p3, r);
    const double z = r * r;
    // |r| <= pi/4 (+ slack): sin r = r + r z S(z) (relative error 2^-55), cos r = 1 - z/2 + z^2 C(z) (2^-49)
    double ps = 0x1.5e0ae6796256cp-33;
    ps = d_fma(ps, z, -0x1.ae600a73bc9bcp-26);
    ps = d_fma(ps, z, 0x1.71de379600d7fp-19);
    ps = d_fma(ps, z, -0x1.a01a019e83411p-13);
    ps = d_fma(ps, z, 0x1.1111111110bb1p-7);
    ps = d_fma(ps, z, -0x1.5555555555555p-3);
    const double sn = d_fma(r * z, ps, r);
    double pc = 0x1.1c819b161a46fp-29;
    pc = d_fma(pc, z, -0x1.27e25ef4d05dfp-22);
    pc = d_fma(pc, z, 0x1.a019ff5333bf2p-16);
    pc = d_fma(pc, z, -0x1.6c16c16b61208p-10);
    pc = d_fma(pc, z, 0x1.5555555555436p-5);
    const double cs = d_fma(z * z, pc, d_fma(-0.5, z, 1.0));
    // quadrant q mod 4: (sin, cos) = (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn)
    const double a = (q & 1u) ? cs : sn, b = (q & 1u) ? sn : cs;
    *s_out = (q & 2u) ? -a : a;
    *c_out = (((q + 1u) & 2u) != 0u) ? -b : b;
}
// which: 0 sin, 1 cos, 2 tan of a binary32 with |x| > 65536 (or NaN)
HNB_HD float f_trig_big(float x, int which) {
    const bool finite = f_abs(x) <= 3.4028234663852886e38f, in_range = f_abs(x) <= 1099511627776.0f;  // 2^40
    double s, c; d_sincos(in_range ? (double)x : 0.0, &s, &c);
    const float r = which == 0 ? (float)s : (which == 1 ? (float)c : (float)(s / c));
    return finite ? r : x - x;  // NaN / inf -> NaN
}

#ifdef ORACLE_LIBM
/* Independent flavour of the oracle (libhanabi_oracle_libm.so): the transcendental WGSL builtins are evaluated by the host's
 * libm in binary64 and rounded once to binary32 - no code shared with the product's hnb_math.h. The GPU parity tests compare
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
// ---- binary32 kernels ------------------------------------------------------------------
HNB_HD float f_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// Round to nearest integer (ties to even) with two IEEE additions; valid |x| < 2^22. *low32: that integer modulo 2^32 (two's complement,
// from the low mantissa bits of the biased sum), without a float -> int conversion.
HNB_HD float f_rint_bits(float x, uint32_t* low32) {
    const float magic = 12582912.0f;  // 1.5 * 2^23
    const float t = x + magic;  // never folded: built without fast-math / reassociation
    *low32 = f2u(t) - 0x4b400000u;
    return t - magic;
}
// sin and cos of a float with |x| <= 65536: k = rint(x 2/pi), r = x - k pi/2 with pi/2 in three binary32 pieces (one fma each: the
// products are exact inside the fma), |r| <= pi/4 + 2^-9; sin r = r + r z S(z), cos r = 1 - z/2 + z^2 C(z), z = r^2
HNB_HD void f_sincos_small(float x, float* s_out, float* c_out) {
    uint32_t q;
    const float k = f_rint_bits(x * 0x1.45f306p-1f, &q);
    float r = f_fma(-k, 0x1.921fb6p+0f, x);
    r = f_fma(-k, -0x1.777a5cp-25f, r);
    r = f_fma(-k, -0x1.ee59dap-50f, r);
    const float z = r * r;
    float ps = -0x1.9ac63ep-13f;
    ps = f_fma(ps, z, 0x1.110c22p-7f);
    ps = f_fma(ps, z, -0x1.555552p-3f);
    const float sn = f_fma(r * z, ps, r);
    float pc = 0x1.9bd5d8p-16f;
    pc = f_fma(pc, z, -0x1.6c12cep-10f);
    pc = f_fma(pc, z, 0x1.555554p-5f);
    const float cs = f_fma(z * z, pc, f_fma(-0.5f, z, 1.0f));
    // quadrant q mod 4: (sin, cos) = (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn)
    const float a = (q & 1u) ? cs : sn, b = (q & 1u) ? sn : cs;
    *s_out = (q & 2u) ? -a : a;
    *c_out = (((q + 1u) & 2u) != 0u) ? -b : b;
}
// 2^k as a float, k in [-126, 127]
HNB_HD float f_pow2i(int32_t k) { return u2f((uint32_t)(k + 127) << 23); }
// p 2^k for p in [1/2, 5/2] and any k: two scalings, the first exact, the second rounding once (into the subnormals, to zero or to infinity)
HNB_HD float f_scale2(float p, int32_t k) {
    k = k < -252 ? -252 : (k > 254 ? 254 : k);
    const int32_t k1 = k >> 1;   // floor(k / 2): both halves in [-126, 127]
    return (p * f_pow2i(k1)) * f_pow2i(k - k1);
}
// exp(r) for |r| <= ln2/2 + 2^-9: 1 + r + r^2 E(r)
HNB_HD float f_exp_poly(float r) {
    float p = 0x1.6d127p-10f;
    p = f_fma(p, r, 0x1.120cd4p-7f);
    p = f_fma(p, r, 0x1.555518p-5f);
    p = f_fma(p, r, 0x1.5554dcp-3f);
    p = f_fma(p, r, 0x1.0p-1f);
    return f_fma(r * r, p, r) + 1.0f;
}
// 2^r for |r| <= 1/2 + 2^-9: 1 + r E2(r)
HNB_HD float f_exp2_poly(float r) {
    float p = 0x1.44149ap-13f;
    p = f_fma(p, r, 0x1.5f0a1cp-10f);
    p = f_fma(p, r, 0x1.3b2a52p-7f);
    p = f_fma(p, r, 0x1.c6af6ap-5f);
    p = f_fma(p, r, 0x1.ebfbep-3f);
    p = f_fma(p, r, 0x1.62e43p-1f);
    return f_fma(r, p, 1.0f);
}
// a finite x > 0 as m 2^e with m in [2/3, 4/3): returns f = m - 1 (exact)
HNB_HD float f_log_reduce(float x, float* e_out) {
    uint32_t ix = f2u(x);
    float bias = 0.0f;
    if (ix < 0x00800000u) { ix = f2u(x * 8388608.0f); bias = -23.0f; }   // subnormal: scaled by 2^23 (exact)
    const uint32_t i = (ix - 0x3f2aaaabu) & 0xff800000u;
    *e_out = (float)((int32_t)i >> 23) + bias;
    return u2f(ix - i) - 1.0f;
}
// ln(1 + f) for |f| <= 1/3 as head + *lo (|*lo| <= ulp(head) / 2; relative error of the sum 2^-28): f - f^2/2 + f^3 L(f), with f^2
// carried exactly (product and its fma residual) and the rounding error of f - f^2/2 recovered
HNB_HD float f_ln1p_parts(float f, float* lo) {
    const float h = f * f, hl = f_fma(f, f, -h);          // f^2 = h + hl exactly
    float p = 0x1.9fe062p-4f;
    p = f_fma(p, f, -0x1.c129fcp-4f);
    p = f_fma(p, f, 0x1.687256p-4f);
    p = f_fma(p, f, -0x1.8cdb84p-4f);
    p = f_fma(p, f, 0x1.c7b586p-4f);
    p = f_fma(p, f, -0x1.0051d4p-3f);
    p = f_fma(p, f, 0x1.2490bcp-3f);
    p = f_fma(p, f, -0x1.5553aep-3f);
    p = f_fma(p, f, 0x1.99999cp-3f);
    p = f_fma(p, f, -0x1.000002p-2f);
    p = f_fma(p, f, 0x1.555556p-2f);
    const float c = (h * f) * p;                          // f^3 L(f)
    const float s = f_fma(-0.5f, h, f);                   // f - h/2, rounded ...
    const float serr = f_fma(-0.5f, h, f - s);            // ... and what the rounding lost ((f - s) is exact: |h/2| <= |f|/6)
    const float t = serr + f_fma(-0.5f, hl, c);           // (|t| < |s| / 8)
    const float head = s + t;
    *lo = (s - head) + t;                                  // exact: head + *lo == s + t
    return head;
}
// log2(x) of a finite x > 0 as head + *tail (relative error of the sum 2^-28): what pow needs to keep y log2 x accurate
HNB_HD float f_log2_parts(float x, float* tail) {
    float e, lo;
    const float f = f_log_reduce(x, &e);
    const float s = f_ln1p_parts(f, &lo);
    const float th = s * 0x1.715476p+0f;
    const float tl = f_fma(s, 0x1.715476p+0f, -th) + f_fma(s, 0x1.4ae0cp-26f, lo * 0x1.715476p+0f);
    const float hi = e + th;                               // |e| >= 1 > |th| or e == 0: (e - hi) + th is the exact rounding error
    const float t2 = ((e - hi) + th) + tl;
    const float head = hi + t2;
    *tail = (hi - head) + t2;
    return head;
}
// asin(s) = s + s z P(z) for z = s^2 <= 1/4
HNB_HD float f_asin_core(float s, float z) {
    float p = 0x1.3810d4p-5f;
    p = f_fma(p, z, 0x1.b3018ep-6f);
    p = f_fma(p, z, 0x1.70a7dp-5f);
    p = f_fma(p, z, 0x1.33272cp-4f);
    p = f_fma(p, z, 0x1.55555ep-3f);
    return f_fma(s * z, p, s);
}

// ---- binary32 entry points (what WGSL `sin(x)` etc. mean in this framework) ----------
HNB_HD bool trig_small(float x) { return f_abs(x) <= 65536.0f; }
HNB_HD float f_sin(float x) {
    if (!trig_small(x)) return f_trig_big(x, 0);
    float s, c; f_sincos_small(x, &s, &c);
    return s;
}
HNB_HD float f_cos(float x) {
    if (!trig_small(x)) return f_trig_big(x, 1);
    float s, c; f_sincos_small(x, &s, &c);
    return c;
}
HNB_HD float f_tan(float x) {
    if (!trig_small(x)) return f_trig_big(x, 2);
    float s, c; f_sincos_small(x, &s, &c);
    return s / c;
}
HNB_HD float f_exp(float x) {
    if (x != x) return x;
    const float xc = x > 90.0f ? 90.0f : (x < -105.0f ? -105.0f : x);   // (beyond: infinity / zero either way)
    uint32_t ki;
    const float k = f_rint_bits(xc * 0x1.715476p+0f, &ki);
    float r = f_fma(-k, 0x1.62e43p-1f, xc);
    r = f_fma(-k, -0x1.05c61p-29f, r);
    return f_scale2(f_exp_poly(r), (int32_t)ki);
}
HNB_HD float f_exp2(float x) {
    if (x != x) return x;
    const float xc = x > 130.0f ? 130.0f : (x < -152.0f ? -152.0f : x);
    uint32_t ki;
    const float k = f_rint_bits(xc, &ki);
    return f_scale2(f_exp2_poly(xc - k), (int32_t)ki);
}
// log of NaN, a negative number, zero, infinity
HNB_HD float f_log_special(float x, bool* special) {
    *special = true;
    if (x != x) return x;
    if (x < 0.0f) return f_nan();
    if (x == 0.0f) return -f_inf();
    if (x == f_inf()) return x;
    *special = false;
    return 0.0f;
}
HNB_HD float f_log(float x) {
    bool sp; const float sv = f_log_special(x, &sp);
    if (sp) return sv;
    float e, lo;
    const float f = f_log_reduce(x, &e);
    const float s = f_ln1p_parts(f, &lo);
    return f_fma(e, 0x1.62e43p-1f, s + f_fma(e, -0x1.05c61p-29f, lo));   // e ln2 + ln m, ln2 in two pieces
}
HNB_HD float f_log2(float x) {
    bool sp; const float sv = f_log_special(x, &sp);
    if (sp) return sv;
    float tl;
    const float hi = f_log2_parts(x, &tl);
    return hi + tl;
}
// WGSL pow(x, y): defined here as exp2(y * log2 x) for x > 0 (log2 x carried as head + tail); x < 0 -> NaN.
HNB_HD float f_pow(float x, float y) {
    if (x != x || y != y) return x + y;
    if (y == 0.0f) return 1.0f;
    if (x < 0.0f) return f_nan();
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : f_inf();
    if (x == f_inf()) return (y > 0.0f) ? f_inf() : 0.0f;
    float tl;
    const float hi = f_log2_parts(x, &tl);
    float ph = y * hi;
    const float pl = f_fma(y, hi, -ph) + y * tl;
    const bool sat = !(f_abs(ph) < 200.0f);                // infinite / huge: the result saturates whatever the low bits are
    if (sat) ph = ph > 0.0f ? 200.0f : -200.0f;
    uint32_t ki;
    const float k = f_rint_bits(ph, &ki);
    const float r = (ph - k) + (sat ? 0.0f : pl);
    return f_scale2(f_exp2_poly(r), (int32_t)ki);
}
// asin / acos; |x| > 1 -> NaN. |x| <= 1/2: t = asin x directly; beyond: t = asin sqrt((1 - |x|) / 2) ((1 - |x|) / 2 is exact) and
// asin |x| = pi/2 - 2 t. One polynomial evaluation either way (selects, no branch: a wave has lanes on both sides).
HNB_HD float f_asin(float x) {
    const float ax = f_abs(x);
    if (!(ax <= 1.0f)) return f_nan();
    const bool small = ax <= 0.5f;
    const float z = small ? x * x : (1.0f - ax) * 0.5f;
    const float s = small ? x : f_sqrt(z);
    const float t = f_asin_core(s, z);
    const float r = f_fma(-2.0f, t, 0x1.921fb6p+0f) + -0x1.777a5cp-25f;
    return small ? t : (x < 0.0f ? -r : r);
}
HNB_HD float f_acos(float x) {
    const float ax = f_abs(x);
    if (!(ax <= 1.0f)) return f_nan();
    const bool small = ax <= 0.5f;
    const float z = small ? x * x : (1.0f - ax) * 0.5f;
    const float s = small ? x : f_sqrt(z);
    const float t = f_asin_core(s, z);
    const float far = x < 0.0f ? f_fma(-2.0f, t, 0x1.921fb6p+1f) + -0x1.777a5cp-24f : 2.0f * t;
    return small ? (0x1.921fb6p+0f - t) + -0x1.777a5cp-25f : far;
}
// atan (NaN -> NaN): at most one division. t = |x| <= tan(pi/8): atan t; t <= tan(3 pi/8): pi/4 + atan((t-1)/(t+1)); beyond:
// pi/2 - atan(1/t); atan u = u + u z A(z), z = u^2 <= tan^2(pi/8); pi/4 and pi/2 in two pieces
HNB_HD float f_atan(float x) {
    if (x != x) return x;
    const bool neg = f_signbit(x);
    const float t = f_abs(x);
    const bool mid = t > 0x1.a8279ap-2f && t <= 0x1.3504f4p+1f, big = t > 0x1.3504f4p+1f;
    const float num = mid ? t - 1.0f : (big ? -1.0f : t), den = mid ? t + 1.0f : (big ? t : 1.0f);
    const float u = num / den;
    const float z = u * u;
    float p = -0x1.083894p-4f;
    p = f_fma(p, z, 0x1.b80c54p-4f);
    p = f_fma(p, z, -0x1.242008p-3f);
    p = f_fma(p, z, 0x1.99973p-3f);
    p = f_fma(p, z, -0x1.555554p-2f);
    float r = f_fma(u * z, p, u);
    r = (mid ? 0x1.921fb6p-1f : (big ? 0x1.921fb6p+0f : 0.0f)) + (r + (mid ? -0x1.777a5cp-26f : (big ? -0x1.777a5cp-25f : 0.0f)));
    return neg ? -r : r;
}
HNB_HD float f_atan2(float y, float x) {
    if (x != x || y != y) return x + y;
    if (x > 0.0f) return f_atan(y / x);
    if (x < 0.0f) {
        const float a = f_atan(y / x);
        return f_signbit(y) ? (a - -0x1.777a5cp-24f) - 0x1.921fb6p+1f : (a + -0x1.777a5cp-24f) + 0x1.921fb6p+1f;
    }
    if (y > 0.0f) return 0x1.921fb6p+0f;
    if (y < 0.0f) return -0x1.921fb6p+0f;
    if (f_signbit(x)) return f_signbit(y) ? -0x1.921fb6p+1f : 0x1.921fb6p+1f;
    return y;
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

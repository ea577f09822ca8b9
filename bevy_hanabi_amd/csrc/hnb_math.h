// hanabi-math: the arithmetic definition of every WGSL builtin the hot path uses.
//
// The reference executes `sin/cos/acos/pow/log/normalize/smoothstep/...` as WGSL
// builtins whose accuracy is implementation-defined (naga -> driver; see SURVEY.md
// §8(c) "third-party arithmetic"). This header pins ONE definition so that the HIP
// kernels and the CPU oracle produce bit-identical particles:
//
//   * f32 + - * / sqrt, comparisons, floor/ceil/trunc/roundEven: IEEE-754 binary32,
//     one rounding per source-level operation (everything is built with
//     -ffp-contract=off and correctly rounded f32 divide/sqrt).
//   * transcendental functions: evaluated in binary64 with only + - * / fma and bit
//     manipulation (no libm, no hardware approximations), then rounded once to
//     binary32. Error before the final rounding is < 2^-45 relative, i.e. the result
//     is the correctly rounded f32 value except in rare near-ties (about one argument
//     in a million: within 1 ulp of libm always, tests/test_math.py).
//     Trigonometric arguments with |x| > 2^40 are defined as x = 0 (sin 0, cos 1).
//   * normalize(v) = v * (1 / length(v)): one IEEE division and three multiplications
//     (WGSL leaves the accuracy of normalize to the implementation: "inherited from
//     v / length(v)", whose own division may be 2.5 ulp off). Domain: the reciprocal never
//     overflows where v / length(v) is finite - dot(v, v) is an f32, so it is either 0
//     (length 0: both forms give inf / NaN) or >= 2^-149, length >= 2^-74.5 and
//     1 / length <= 2^74.5 (tests/test_math.py::test_normalize_reciprocal_domain).
//
// Compiles as plain C++ (host) and as HIP device code (HNB_HD).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#else  // hiprtc keeps the fixed-width types in __hip_internal
using __hip_internal::int8_t; using __hip_internal::uint8_t; using __hip_internal::int16_t; using __hip_internal::uint16_t;
using __hip_internal::int32_t; using __hip_internal::uint32_t; using __hip_internal::int64_t; using __hip_internal::uint64_t;
#endif

#if defined(__HIPCC__)
#define HNB_HD __host__ __device__ __forceinline__
#define HNB_HD_MEMBER __host__ __device__ __forceinline__
#else
#define HNB_HD static inline
#define HNB_HD_MEMBER inline
#endif

#define HNB_TABLE static constexpr   // constant tables: emitted for the host and (as constant data) for the device

namespace hnb {

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
// Every step is ONE correctly rounded IEEE-754 binary64 operation - + - * / or a fused multiply-add (v_fma_f64 on gfx950, vfmadd
// or the C library's exact fma() on the host: the same result everywhere, which -ffp-contract=off alone cannot promise for an
// a * b + c the compiler is free to fuse or not) - on minimax polynomials (tools/gen_math_coeffs.py derives the coefficients and
// prints their error; tools/gen_math_kernels.py writes this section): approximation errors are below 2^-47 relative, far inside
// the final rounding to binary32. Branch-free where both sides of a branch would run in a wave anyway.
HNB_HD double d_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// p * z + c with a LITERAL c: the same fma. On the device it is spelled out so that the coefficient travels in an SGPR pair
// (v_fma_f64 v, v, v, s): the compiler's own choice, v_fmac_f64, first moves every coefficient into the destination VGPR pair -
// two more VALU instructions per Horner step, a third of the VALU work of a sphere-shaped spawn.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double d_fma_c(double p, double z, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
    return r;
}
#else
HNB_HD double d_fma_c(double p, double z, double c) { return __builtin_fma(p, z, c); }
#endif
// Round to nearest integer (ties to even) with two IEEE additions; valid |x| < 2^51. *low32: that integer modulo 2^32 (the low
// mantissa bits of the biased sum), without a float -> int conversion.
HNB_HD double d_rint_bits(double x, uint32_t* low32) {
    const double magic = 6755399441055744.0;  // 1.5 * 2^52
    double t = x + magic;  // never folded: built without fast-math / reassociation
    *low32 = (uint32_t)d2u(t);
    return t - magic;
}
HNB_HD double d_rint(double x) { uint32_t lo; return d_rint_bits(x, &lo); }

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
    ps = d_fma_c(ps, z, -0x1.ae600a73bc9bcp-26);
    ps = d_fma_c(ps, z, 0x1.71de379600d7fp-19);
    ps = d_fma_c(ps, z, -0x1.a01a019e83411p-13);
    ps = d_fma_c(ps, z, 0x1.1111111110bb1p-7);
    ps = d_fma_c(ps, z, -0x1.5555555555555p-3);
    const double sn = d_fma(r * z, ps, r);
    double pc = 0x1.1c819b161a46fp-29;
    pc = d_fma_c(pc, z, -0x1.27e25ef4d05dfp-22);
    pc = d_fma_c(pc, z, 0x1.a019ff5333bf2p-16);
    pc = d_fma_c(pc, z, -0x1.6c16c16b61208p-10);
    pc = d_fma_c(pc, z, 0x1.5555555555436p-5);
    const double cs = d_fma(z * z, pc, d_fma(-0.5, z, 1.0));
    // quadrant q mod 4: (sin, cos) = (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn)
    const double a = (q & 1u) ? cs : sn, b = (q & 1u) ? sn : cs;
    *s_out = (q & 2u) ? -a : a;
    *c_out = (((q + 1u) & 2u) != 0u) ? -b : b;
}

// 2^k for integer k in [-1022, 1023]
HNB_HD double d_pow2i(int32_t k) { return u2d((uint64_t)(uint32_t)(k + 1023) << 52); }

// exp(x) for finite x; caller clamps to [-120, 100]
HNB_HD double d_exp(double x) {
    const double log2e = 0x1.71547652b82fep+0;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    uint32_t ki;
    const double k = d_rint_bits(x * log2e, &ki);
    double r = d_fma(-k, ln2_hi, x);
    r = d_fma(-k, ln2_lo, r);
    // |r| <= ln2 / 2: exp r = 1 + r + r^2 E(r), relative error 2^-49
    double p = 0x1.2880501c9131ap-22;
    p = d_fma_c(p, r, 0x1.72c7432675d87p-19);
    p = d_fma_c(p, r, 0x1.a019c99a94203p-16);
    p = d_fma_c(p, r, 0x1.a019ad9325b18p-13);
    p = d_fma_c(p, r, 0x1.6c16c173921fdp-10);
    p = d_fma_c(p, r, 0x1.1111111c4acecp-7);
    p = d_fma_c(p, r, 0x1.5555555554cb3p-5);
    p = d_fma_c(p, r, 0x1.5555555553b6ep-3);
    p = d_fma_c(p, r, 0x1.0000000000000p-1);
    const double t = d_fma(r * r, p, r);
    return (1.0 + t) * d_pow2i((int32_t)ki);
}

// natural log of a finite, strictly positive, normal double: x = 2^k z with z in [0.6875, 1.375); the 64 intervals of z (top six
// fraction bits) each have a centre c with 1/c and log c tabulated, so log x = k ln2 + log c + log1p(r), r = z / c - 1 (one fma),
// |r| <= 2^-6, log1p r = r - r^2/2 + r^3 L(r) with relative error 2^-48. The two intervals that meet at z = 1 use c = 1: x near 1
// keeps its relative accuracy (log x = log1p(x - 1), x - 1 exact).
HNB_TABLE double kLogTab[64][2] = {   // {1/c, log c}
    {0x1.724287f46debcp+0, -0x1.79e26687cfb3dp-2},
    {0x1.6e1f76b4337c7p+0, -0x1.6e60ee6af1973p-2},
    {0x1.6a13cd1537290p+0, -0x1.630030b3aac48p-2},
    {0x1.661ec6a5122f9p+0, -0x1.57bf753c8d1fbp-2},
    {0x1.623fa77016240p+0, -0x1.4c9e09e172c3dp-2},
    {0x1.5e75bb8d015e7p+0, -0x1.419b423d5e8c6p-2},
    {0x1.5ac056b015ac0p+0, -0x1.36b6776be1116p-2},
    {0x1.571ed3c506b3ap+0, -0x1.2bef07cdc9355p-2},
    {0x1.5390948f40febp+0, -0x1.214456d0eb8d5p-2},
    {0x1.5015015015015p+0, -0x1.16b5ccbacfb73p-2},
    {0x1.4cab88725af6ep+0, -0x1.0c42d676162e2p-2},
    {0x1.49539e3b2d067p+0, -0x1.01eae5626c691p-2},
    {0x1.460cbc7f5cf9ap+0, -0x1.ef5ade4dcffe5p-3},
    {0x1.42d6625d51f87p+0, -0x1.db13db0d48941p-3},
    {0x1.3fb013fb013fbp+0, -0x1.c6ffbc6f00f71p-3},
    {0x1.3c995a47babe7p+0, -0x1.b31d8575bce3bp-3},
    {0x1.3991c2c187f63p+0, -0x1.9f6c407089663p-3},
    {0x1.3698df3de0748p+0, -0x1.8beafeb38fe8fp-3},
    {0x1.33ae45b57bcb2p+0, -0x1.7898d85444c74p-3},
    {0x1.30d190130d190p+0, -0x1.6574ebe8c1339p-3},
    {0x1.2e025c04b8097p+0, -0x1.527e5e4a1b58dp-3},
    {0x1.2b404ad012b40p+0, -0x1.3fb45a59928cap-3},
    {0x1.288b01288b013p+0, -0x1.2d1610c86813dp-3},
    {0x1.25e22708092f1p+0, -0x1.1aa2b7e23f729p-3},
    {0x1.23456789abcdfp+0, -0x1.08598b59e3a07p-3},
    {0x1.20b470c67c0d9p+0, -0x1.ec739830a1126p-4},
    {0x1.1e2ef3b3fb874p+0, -0x1.c885801bc4b20p-4},
    {0x1.1bb4a4046ed29p+0, -0x1.a4e7640b1bc38p-4},
    {0x1.19453808ca29cp+0, -0x1.8197e2f40e3f0p-4},
    {0x1.16e0689427379p+0, -0x1.5e95a4d9791cdp-4},
    {0x1.1485f0e0acd3bp+0, -0x1.3bdf5a7d1ee5ep-4},
    {0x1.12358e75d3033p+0, -0x1.1973bd1465561p-4},
    {0x1.0fef010fef011p+0, -0x1.eea31c006b87cp-5},
    {0x1.0db20a88f4696p+0, -0x1.aaef2d0fb1108p-5},
    {0x1.0b7e6ec259dc8p+0, -0x1.67c94f2d4bb65p-5},
    {0x1.0953f39010954p+0, -0x1.252f32f8d1840p-5},
    {0x1.073260a47f7c6p+0, -0x1.c63d2ec14aad7p-6},
    {0x1.05197f7d73404p+0, -0x1.432a925980cbcp-6},
    {0x1.03091b51f5e1ap+0, -0x1.82448a388a283p-7},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.f44659e4a4271p-1, 0x1.7b91b07d5b126p-6},
    {0x1.ecc07b301ecc0p-1, 0x1.39e87b9febd68p-5},
    {0x1.e573ac901e574p-1, 0x1.b42dd711971b9p-5},
    {0x1.de5d6e3f8868ap-1, 0x1.16536eea37ae3p-4},
    {0x1.d77b654b82c34p-1, 0x1.51b073f06183cp-4},
    {0x1.d0cb58f6ec074p-1, 0x1.8c345d6319b23p-4},
    {0x1.ca4b3055ee191p-1, 0x1.c5e548f5bc743p-4},
    {0x1.c3f8f01c3f8f0p-1, 0x1.fec9131dbeabcp-4},
    {0x1.bdd2b899406f7p-1, 0x1.1b72ad52f67a2p-3},
    {0x1.b7d6c3dda338bp-1, 0x1.371fc201e8f75p-3},
    {0x1.b2036406c80d9p-1, 0x1.526e5e3a1b438p-3},
    {0x1.ac5701ac5701bp-1, 0x1.6d60fe719d21bp-3},
    {0x1.a6d01a6d01a6dp-1, 0x1.87fa06520c911p-3},
    {0x1.a16d3f97a4b02p-1, 0x1.a23bc1fe2b561p-3},
    {0x1.9c2d14ee4a102p-1, 0x1.bc286742d8cd4p-3},
    {0x1.970e4f80cb872p-1, 0x1.d5c216b4fbb94p-3},
    {0x1.920fb49d0e229p-1, 0x1.ef0adcbdc5935p-3},
    {0x1.8d3018d3018d3p-1, 0x1.0402594b4d041p-2},
    {0x1.886e5f0abb04ap-1, 0x1.1058bf9ae4ad4p-2},
    {0x1.83c977ab2beddp-1, 0x1.1c898c16999fbp-2},
    {0x1.7f405fd017f40p-1, 0x1.2895a13de86a4p-2},
    {0x1.7ad2208e0ecc3p-1, 0x1.347dd9a987d56p-2},
    {0x1.767dce434a9b1p-1, 0x1.404308686a7e4p-2},
};
HNB_HD double d_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const uint64_t ix = d2u(x);
    const uint32_t hi = (uint32_t)(ix >> 32) - 0x3fe60000u;      // (the low word of the offset is zero: only the high word changes)
    const uint32_t i = (hi >> 14) & 63u;
    const int32_t k = (int32_t)hi >> 20;                          // arithmetic shift: floor
    const double z = u2d(ix - ((uint64_t)(hi & 0xfff00000u) << 32));
    const double r = d_fma(z, kLogTab[i][0], -1.0);
    const double r2 = r * r;
    double p = 0x1.24a4c91a63c23p-3;
    p = d_fma_c(p, r, -0x1.556a254394372p-3);
    p = d_fma_c(p, r, 0x1.9999994c9b676p-3);
    p = d_fma_c(p, r, -0x1.ffffffa962285p-3);
    p = d_fma_c(p, r, 0x1.5555555555555p-2);
    const double l1p = d_fma(r2 * r, p, d_fma(-0.5, r2, r));
    const double kd = (double)k;
    return (kd * ln2_hi + kLogTab[i][1]) + (l1p + kd * ln2_lo);
}

// sqrt of a double in [2^-100, 2^100] (and 0; negative / NaN -> NaN): the binary32 root and its binary32 reciprocal (both IEEE, so
// identical on host and device) seed ONE Heron step whose division is a multiplication by that reciprocal: relative error < 2^-46.
HNB_HD double d_sqrt(double a) {
    const float sf = f_sqrt((float)a);
    const double s0 = (double)sf, h = 0.5 * (double)(1.0f / sf);
    const double s = d_fma(d_fma(-s0, s0, a), h, s0);
    return (a > 0.0) ? s : ((a == 0.0) ? 0.0 : u2d(0x7ff8000000000000ull));  // 0 -> 0, negative / NaN -> NaN
}

// asin(t) = t + t z P(z), z = t^2 <= 1/4: relative error 2^-50
HNB_HD double d_asin_poly(double z) {
    double p = 0x1.c8ea18fd14be4p-6;
    p = d_fma_c(p, z, -0x1.c05fb18feb0f9p-8);
    p = d_fma_c(p, z, 0x1.fa6d9d973284ep-7);
    p = d_fma_c(p, z, 0x1.5114f44b77900p-7);
    p = d_fma_c(p, z, 0x1.cf629dd6ae5fdp-7);
    p = d_fma_c(p, z, 0x1.1c0d42d9a72edp-6);
    p = d_fma_c(p, z, 0x1.6e8f37bce829dp-6);
    p = d_fma_c(p, z, 0x1.f1c6ff632bc58p-6);
    p = d_fma_c(p, z, 0x1.6db6dba9ded46p-5);
    p = d_fma_c(p, z, 0x1.3333333302d42p-4);
    p = d_fma_c(p, z, 0x1.55555555555bcp-3);
    return p;
}
// asin / acos of a double; |x| > 1 -> NaN. |x| <= 1/2: t = asin x directly; beyond: t = asin sqrt((1 - |x|) / 2) ((1 - |x|) / 2 is
// exact) and asin |x| = pi/2 - 2 t. One polynomial evaluation either way (selects, no branch: a wave has lanes on both sides).
HNB_HD double d_asin(double x) {
    const double pi_2_hi = 0x1.921fb54442d18p+0, pi_2_lo = 0x1.1a62633145c07p-54;
    const double ax = x < 0.0 ? -x : x;
    const bool small = ax <= 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    const double s = small ? x : d_sqrt(z);
    const double t = d_fma(s * z, d_asin_poly(z), s);
    const double r = d_fma(-2.0, t, pi_2_hi) + pi_2_lo;
    return small ? t : (x < 0.0 ? -r : r);
}
HNB_HD double d_acos(double x) {
    const double pi_2_hi = 0x1.921fb54442d18p+0, pi_2_lo = 0x1.1a62633145c07p-54;
    const double ax = x < 0.0 ? -x : x;
    const bool small = ax <= 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    const double s = small ? x : d_sqrt(z);
    const double t = d_fma(s * z, d_asin_poly(z), s);
    const double far = x < 0.0 ? d_fma(-2.0, t, 2.0 * pi_2_hi) + 2.0 * pi_2_lo : 2.0 * t;
    return small ? pi_2_hi - (t - pi_2_lo) : far;
}

// atan of any double (NaN -> NaN): at most one division. t = |x| <= tan(pi/8): atan t; t <= tan(3 pi/8): pi/4 + atan((t-1)/(t+1));
// beyond: pi/2 - atan(1/t); atan u = u + u z A(z), z = u^2 <= tan^2(pi/8), relative error 2^-47
HNB_HD double d_atan(double x) {
    if (x != x) return x;
    const double pi_2 = 0x1.921fb54442d18p+0, pi_4 = 0x1.921fb54442d18p-1;
    const bool neg = d_signbit(x);
    const double t = neg ? -x : x;
    const bool mid = t > 0.41421356237309503 && t <= 2.4142135623730951, big = t > 2.4142135623730951;
    const double num = mid ? t - 1.0 : (big ? -1.0 : t), den = mid ? t + 1.0 : (big ? t : 1.0);
    const double u = num / den;
    const double z = u * u;
    double p = -0x1.be20c62f176ddp-6;
    p = d_fma_c(p, z, 0x1.a769bd3353c1cp-5);
    p = d_fma_c(p, z, -0x1.0c52a4b5af878p-4);
    p = d_fma_c(p, z, 0x1.3a9d83feefaedp-4);
    p = d_fma_c(p, z, -0x1.74563c795c0ffp-4);
    p = d_fma_c(p, z, 0x1.c71c381ab56c6p-4);
    p = d_fma_c(p, z, -0x1.249248aa52bc5p-3);
    p = d_fma_c(p, z, 0x1.99999998d12f0p-3);
    p = d_fma_c(p, z, -0x1.55555555553a3p-2);
    double r = d_fma(u * z, p, u);
    r = (mid ? pi_4 : (big ? pi_2 : 0.0)) + r;
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

// ---- binary32 entry points (what WGSL `sin(x)` etc. mean in this framework) ----------
HNB_HD bool trig_in_range(float x) { return f_abs(x) <= 1099511627776.0f; }  // 2^40

HNB_HD float f_sin(float x) {
    const bool finite = f_abs(x) <= 3.4028234663852886e38f, in_range = trig_in_range(x);
    double s, c; d_sincos(in_range ? (double)x : 0.0, &s, &c);
    return finite ? (in_range ? (float)s : 0.0f) : x - x;  // NaN/inf -> NaN; |x| > 2^40 -> sin 0
}
HNB_HD float f_cos(float x) {
    const bool finite = f_abs(x) <= 3.4028234663852886e38f, in_range = trig_in_range(x);
    double s, c; d_sincos(in_range ? (double)x : 0.0, &s, &c);
    return finite ? (in_range ? (float)c : 1.0f) : x - x;
}
HNB_HD float f_tan(float x) {
    const bool finite = f_abs(x) <= 3.4028234663852886e38f, in_range = trig_in_range(x);
    double s, c; d_sincos(in_range ? (double)x : 0.0, &s, &c);
    return finite ? (in_range ? (float)(s / c) : 0.0f) : x - x;
}
HNB_HD float f_atan(float x) { return (float)d_atan((double)x); }
HNB_HD float f_atan2(float y, float x) { return (float)d_atan2((double)y, (double)x); }
HNB_HD float f_asin(float x) { return (float)d_asin((double)x); }
HNB_HD float f_acos(float x) { return (float)d_acos((double)x); }
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
    uint32_t ki;
    const double k = d_rint_bits(xd, &ki);
    const double r = (xd - k) * 0.69314718055994528623;
    return (float)(d_exp(r) * d_pow2i((int32_t)ki));
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
    // computed for every lane, the special cases selected afterwards (d_log of a non-positive or non-finite x is garbage, never a fault)
    double t = (double)y * d_log((double)x);
    if (t > 100.0) t = 100.0;
    if (t < -120.0) t = -120.0;
    const float r = (float)d_exp(t != t ? 0.0 : t);
    if (x != x || y != y) return x + y;
    if (y == 0.0f) return 1.0f;
    if (x < 0.0f) return f_nan();
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : f_inf();
    if (x == f_inf()) return (y > 0.0f) ? f_inf() : 0.0f;
    return r;
}

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

}  // namespace hnb

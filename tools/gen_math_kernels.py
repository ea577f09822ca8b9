#!/usr/bin/env python3
"""Development tool: writes the "binary64 kernels" section of bevy_hanabi_amd/csrc/hnb_math.h and oracle/oracle_math.h (the two copies of
the arithmetic definition, textually identical in that section) from the coefficients tools/gen_math_coeffs.py derives.
    python tools/gen_math_coeffs.py > /tmp/coeffs.txt && python tools/gen_math_kernels.py /tmp/coeffs.txt"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/coeffs.txt").read()


def block(name):
    m = re.search(r"// " + name + r"\n((?:    .*\n)+)", txt)
    return [ln.strip().split(",")[0] for ln in m.group(1).strip("\n").split("\n")]


def horner(var, z, cs, indent="    "):   # cs: low order first
    out = [f"{indent}double {var} = {cs[-1]};"]
    for c in reversed(cs[:-1]):
        out.append(f"{indent}{var} = d_fma_c({var}, {z}, {c});")
    return "\n".join(out)


tab = re.search(r"// max \|r\|.*\n((?:    \{.*\n)+)", txt).group(1)
tab = "\n".join(ln.split("//")[0].rstrip() for ln in tab.strip("\n").split("\n"))
S, C, AS, E, L, AT = (block(n) for n in ("SIN_6", "COS_5", "ASIN_11", "EXP_9", "LOG1P_64_5", "ATAN8_9"))
body = f'''// ---- binary64 kernels ------------------------------------------------------------
// Every step is ONE correctly rounded IEEE-754 binary64 operation - + - * / or a fused multiply-add (v_fma_f64 on gfx950, vfmadd
// or the C library's exact fma() on the host: the same result everywhere, which -ffp-contract=off alone cannot promise for an
// a * b + c the compiler is free to fuse or not) - on minimax polynomials (tools/gen_math_coeffs.py derives the coefficients and
// prints their error; tools/gen_math_kernels.py writes this section): approximation errors are below 2^-47 relative, far inside
// the final rounding to binary32. Branch-free where both sides of a branch would run in a wave anyway.
HNB_HD double d_fma(double a, double b, double c) {{ return __builtin_fma(a, b, c); }}
// p * z + c with a LITERAL c: the same fma. On the device it is spelled out so that the coefficient travels in an SGPR pair
// (v_fma_f64 v, v, v, s): the compiler's own choice, v_fmac_f64, first moves every coefficient into the destination VGPR pair -
// two more VALU instructions per Horner step, a third of the VALU work of a sphere-shaped spawn.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double d_fma_c(double p, double z, double c) {{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
    return r;
}}
#else
HNB_HD double d_fma_c(double p, double z, double c) {{ return __builtin_fma(p, z, c); }}
#endif
// Round to nearest integer (ties to even) with two IEEE additions; valid |x| < 2^51. *low32: that integer modulo 2^32 (the low
// mantissa bits of the biased sum), without a float -> int conversion.
HNB_HD double d_rint_bits(double x, uint32_t* low32) {{
    const double magic = 6755399441055744.0;  // 1.5 * 2^52
    double t = x + magic;  // never folded: built without fast-math / reassociation
    *low32 = (uint32_t)d2u(t);
    return t - magic;
}}
HNB_HD double d_rint(double x) {{ uint32_t lo; return d_rint_bits(x, &lo); }}

// sin and cos of a finite double with |x| <= 2^40.
HNB_HD void d_sincos(double x, double* s_out, double* c_out) {{
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
{horner("ps", "z", S)}
    const double sn = d_fma(r * z, ps, r);
{horner("pc", "z", C)}
    const double cs = d_fma(z * z, pc, d_fma(-0.5, z, 1.0));
    // quadrant q mod 4: (sin, cos) = (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn)
    const double a = (q & 1u) ? cs : sn, b = (q & 1u) ? sn : cs;
    *s_out = (q & 2u) ? -a : a;
    *c_out = (((q + 1u) & 2u) != 0u) ? -b : b;
}}

// 2^k for integer k in [-1022, 1023]
HNB_HD double d_pow2i(int32_t k) {{ return u2d((uint64_t)(uint32_t)(k + 1023) << 52); }}

// exp(x) for finite x; caller clamps to [-120, 100]
HNB_HD double d_exp(double x) {{
    const double log2e = 0x1.71547652b82fep+0;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    uint32_t ki;
    const double k = d_rint_bits(x * log2e, &ki);
    double r = d_fma(-k, ln2_hi, x);
    r = d_fma(-k, ln2_lo, r);
    // |r| <= ln2 / 2: exp r = 1 + r + r^2 E(r), relative error 2^-49
{horner("p", "r", E)}
    const double t = d_fma(r * r, p, r);
    return (1.0 + t) * d_pow2i((int32_t)ki);
}}

// natural log of a finite, strictly positive, normal double: x = 2^k z with z in [0.6875, 1.375); the 64 intervals of z (top six
// fraction bits) each have a centre c with 1/c and log c tabulated, so log x = k ln2 + log c + log1p(r), r = z / c - 1 (one fma),
// |r| <= 2^-6, log1p r = r - r^2/2 + r^3 L(r) with relative error 2^-48. The two intervals that meet at z = 1 use c = 1: x near 1
// keeps its relative accuracy (log x = log1p(x - 1), x - 1 exact).
HNB_TABLE double kLogTab[64][2] = {{   // {{1/c, log c}}
{tab}
}};
HNB_HD double d_log(double x) {{
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const uint64_t ix = d2u(x);
    const uint32_t hi = (uint32_t)(ix >> 32) - 0x3fe60000u;      // (the low word of the offset is zero: only the high word changes)
    const uint32_t i = (hi >> 14) & 63u;
    const int32_t k = (int32_t)hi >> 20;                          // arithmetic shift: floor
    const double z = u2d(ix - ((uint64_t)(hi & 0xfff00000u) << 32));
    const double r = d_fma(z, kLogTab[i][0], -1.0);
    const double r2 = r * r;
{horner("p", "r", L)}
    const double l1p = d_fma(r2 * r, p, d_fma(-0.5, r2, r));
    const double kd = (double)k;
    return (kd * ln2_hi + kLogTab[i][1]) + (l1p + kd * ln2_lo);
}}

// sqrt of a double in [2^-100, 2^100] (and 0; negative / NaN -> NaN): the binary32 root and its binary32 reciprocal (both IEEE, so
// identical on host and device) seed ONE Heron step whose division is a multiplication by that reciprocal: relative error < 2^-46.
HNB_HD double d_sqrt(double a) {{
    const float sf = f_sqrt((float)a);
    const double s0 = (double)sf, h = 0.5 * (double)(1.0f / sf);
    const double s = d_fma(d_fma(-s0, s0, a), h, s0);
    return (a > 0.0) ? s : ((a == 0.0) ? 0.0 : u2d(0x7ff8000000000000ull));  // 0 -> 0, negative / NaN -> NaN
}}

// asin(t) = t + t z P(z), z = t^2 <= 1/4: relative error 2^-50
HNB_HD double d_asin_poly(double z) {{
{horner("p", "z", AS)}
    return p;
}}
// asin / acos of a double; |x| > 1 -> NaN. |x| <= 1/2: t = asin x directly; beyond: t = asin sqrt((1 - |x|) / 2) ((1 - |x|) / 2 is
// exact) and asin |x| = pi/2 - 2 t. One polynomial evaluation either way (selects, no branch: a wave has lanes on both sides).
HNB_HD double d_asin(double x) {{
    const double pi_2_hi = 0x1.921fb54442d18p+0, pi_2_lo = 0x1.1a62633145c07p-54;
    const double ax = x < 0.0 ? -x : x;
    const bool small = ax <= 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    const double s = small ? x : d_sqrt(z);
    const double t = d_fma(s * z, d_asin_poly(z), s);
    const double r = d_fma(-2.0, t, pi_2_hi) + pi_2_lo;
    return small ? t : (x < 0.0 ? -r : r);
}}
HNB_HD double d_acos(double x) {{
    const double pi_2_hi = 0x1.921fb54442d18p+0, pi_2_lo = 0x1.1a62633145c07p-54;
    const double ax = x < 0.0 ? -x : x;
    const bool small = ax <= 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    const double s = small ? x : d_sqrt(z);
    const double t = d_fma(s * z, d_asin_poly(z), s);
    const double far = x < 0.0 ? d_fma(-2.0, t, 2.0 * pi_2_hi) + 2.0 * pi_2_lo : 2.0 * t;
    return small ? pi_2_hi - (t - pi_2_lo) : far;
}}

// atan of any double (NaN -> NaN): at most one division. t = |x| <= tan(pi/8): atan t; t <= tan(3 pi/8): pi/4 + atan((t-1)/(t+1));
// beyond: pi/2 - atan(1/t); atan u = u + u z A(z), z = u^2 <= tan^2(pi/8), relative error 2^-47
HNB_HD double d_atan(double x) {{
    if (x != x) return x;
    const double pi_2 = 0x1.921fb54442d18p+0, pi_4 = 0x1.921fb54442d18p-1;
    const bool neg = d_signbit(x);
    const double t = neg ? -x : x;
    const bool mid = t > 0.41421356237309503 && t <= 2.4142135623730951, big = t > 2.4142135623730951;
    const double num = mid ? t - 1.0 : (big ? -1.0 : t), den = mid ? t + 1.0 : (big ? t : 1.0);
    const double u = num / den;
    const double z = u * u;
{horner("p", "z", AT)}
    double r = d_fma(u * z, p, u);
    r = (mid ? pi_4 : (big ? pi_2 : 0.0)) + r;
    return neg ? -r : r;
}}

HNB_HD double d_atan2(double y, double x) {{
    const double pi = 3.14159265358979311600;
    const double pi_2 = 1.57079632679489655800;
    if (x != x || y != y) return x + y;
    if (x > 0.0) return d_atan(y / x);
    if (x < 0.0) return d_atan(y / x) + (d_signbit(y) ? -pi : pi);
    if (y > 0.0) return pi_2;
    if (y < 0.0) return -pi_2;
    if (d_signbit(x)) return d_signbit(y) ? -pi : pi;
    return y;
}}

'''
for path, marker_end in (("bevy_hanabi_amd/csrc/hnb_math.h", "// ---- binary32 entry points (what WGSL"), ("oracle/oracle_math.h", "// ---- binary32 entry points ---")):
    p = os.path.join(ROOT, path)
    s = open(p).read()
    a = s.index("// ---- binary64 kernels ---")
    b = s.index(marker_end)
    open(p, "w").write(s[:a] + body + s[b:])
print("wrote the binary64 kernels of hnb_math.h and oracle_math.h")

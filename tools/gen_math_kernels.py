#!/usr/bin/env python3
"""Development tool (hanabi-math v3): writes the transcendental section of bevy_hanabi_amd/csrc/hnb_math.h and oracle/oracle_math.h - the two
copies of the arithmetic definition, textually identical in that section - from tools/math_coeffs.json (tools/gen_math_coeffs.py).
    python tools/gen_math_coeffs.py && python tools/gen_math_kernels.py
Everything between the markers `// ---- transcendental builtins` and `// ---- conversions` of both headers is replaced."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = json.load(open(os.path.join(ROOT, "tools", "math_coeffs.json")))
K = C["consts"]


def hx(x):
    """binary32 hex literal (the value is a binary32: at most 6 hex digits of fraction)"""
    m = re.fullmatch(r"(-?0x[01]\.)([0-9a-f]+)(p[+-]\d+)", float(x).hex())
    frac = m.group(2).rstrip("0")
    assert len(frac) <= 6, x
    return f"{m.group(1)}{frac or '0'}{m.group(3)}f"


def horner(var, z, cs, indent="    "):   # cs: low order first
    out = [f"{indent}float {var} = {hx(cs[-1])};"]
    for c in reversed(cs[:-1]):
        out.append(f"{indent}{var} = f_fma({var}, {z}, {hx(c)});")
    return "\n".join(out)


S, CO, E, E2, L, AS, AT = (C[k] for k in ("SIN_3", "COS_3", "EXP_5", "EXP2_6", "LOG_11", "ASIN_5", "ATAN_5"))

BIG = '''// ---- transcendental builtins ---------------------------------------------------------
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
'''

KERNELS = f'''// ---- binary32 kernels ------------------------------------------------------------------
HNB_HD float f_fma(float a, float b, float c) {{ return __builtin_fmaf(a, b, c); }}
// Round to nearest integer (ties to even) with two IEEE additions; valid |x| < 2^22. *low32: that integer modulo 2^32 (two's complement,
// from the low mantissa bits of the biased sum), without a float -> int conversion.
HNB_HD float f_rint_bits(float x, uint32_t* low32) {{
    const float magic = 12582912.0f;  // 1.5 * 2^23
    const float t = x + magic;  // never folded: built without fast-math / reassociation
    *low32 = f2u(t) - 0x4b400000u;
    return t - magic;
}}
// sin and cos of a float with |x| <= 65536: k = rint(x 2/pi), r = x - k pi/2 with pi/2 in three binary32 pieces (one fma each: the
// products are exact inside the fma), |r| <= pi/4 + 2^-9; sin r = r + r z S(z), cos r = 1 - z/2 + z^2 C(z), z = r^2
HNB_HD void f_sincos_small(float x, float* s_out, float* c_out) {{
    uint32_t q;
    const float k = f_rint_bits(x * {hx(K["TWO_OVER_PI"])}, &q);
    float r = f_fma(-k, {hx(K["PIO2_1"])}, x);
    r = f_fma(-k, {hx(K["PIO2_2"])}, r);
    r = f_fma(-k, {hx(K["PIO2_3"])}, r);
    const float z = r * r;
{horner("ps", "z", S)}
    const float sn = f_fma(r * z, ps, r);
{horner("pc", "z", CO)}
    const float cs = f_fma(z * z, pc, f_fma(-0.5f, z, 1.0f));
    // quadrant q mod 4: (sin, cos) = (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn)
    const float a = (q & 1u) ? cs : sn, b = (q & 1u) ? sn : cs;
    *s_out = (q & 2u) ? -a : a;
    *c_out = (((q + 1u) & 2u) != 0u) ? -b : b;
}}
// 2^k as a float, k in [-126, 127]
HNB_HD float f_pow2i(int32_t k) {{ return u2f((uint32_t)(k + 127) << 23); }}
// p 2^k for p in [1/2, 5/2] and any k: two scalings, the first exact, the second rounding once (into the subnormals, to zero or to infinity)
HNB_HD float f_scale2(float p, int32_t k) {{
    k = k < -252 ? -252 : (k > 254 ? 254 : k);
    const int32_t k1 = k >> 1;   // floor(k / 2): both halves in [-126, 127]
    return (p * f_pow2i(k1)) * f_pow2i(k - k1);
}}
// exp(r) for |r| <= ln2/2 + 2^-9: 1 + r + r^2 E(r)
HNB_HD float f_exp_poly(float r) {{
{horner("p", "r", E)}
    return f_fma(r * r, p, r) + 1.0f;
}}
// 2^r for |r| <= 1/2 + 2^-9: 1 + r E2(r)
HNB_HD float f_exp2_poly(float r) {{
{horner("p", "r", E2)}
    return f_fma(r, p, 1.0f);
}}
// a finite x > 0 as m 2^e with m in [2/3, 4/3): returns f = m - 1 (exact)
HNB_HD float f_log_reduce(float x, float* e_out) {{
    uint32_t ix = f2u(x);
    float bias = 0.0f;
    if (ix < 0x00800000u) {{ ix = f2u(x * 8388608.0f); bias = -23.0f; }}   // subnormal: scaled by 2^23 (exact)
    const uint32_t i = (ix - 0x3f2aaaabu) & 0xff800000u;
    *e_out = (float)((int32_t)i >> 23) + bias;
    return u2f(ix - i) - 1.0f;
}}
// ln(1 + f) for |f| <= 1/3 as head + *lo (|*lo| <= ulp(head) / 2; relative error of the sum 2^-28): f - f^2/2 + f^3 L(f), with f^2
// carried exactly (product and its fma residual) and the rounding error of f - f^2/2 recovered
HNB_HD float f_ln1p_parts(float f, float* lo) {{
    const float h = f * f, hl = f_fma(f, f, -h);          // f^2 = h + hl exactly
{horner("p", "f", L)}
    const float c = (h * f) * p;                          // f^3 L(f)
    const float s = f_fma(-0.5f, h, f);                   // f - h/2, rounded ...
    const float serr = f_fma(-0.5f, h, f - s);            // ... and what the rounding lost ((f - s) is exact: |h/2| <= |f|/6)
    const float t = serr + f_fma(-0.5f, hl, c);           // (|t| < |s| / 8)
    const float head = s + t;
    *lo = (s - head) + t;                                  // exact: head + *lo == s + t
    return head;
}}
// log2(x) of a finite x > 0 as head + *tail (relative error of the sum 2^-28): what pow needs to keep y log2 x accurate
HNB_HD float f_log2_parts(float x, float* tail) {{
    float e, lo;
    const float f = f_log_reduce(x, &e);
    const float s = f_ln1p_parts(f, &lo);
    const float th = s * {hx(K["LOG2E"])};
    const float tl = f_fma(s, {hx(K["LOG2E"])}, -th) + f_fma(s, {hx(K["LOG2E_LO"])}, lo * {hx(K["LOG2E"])});
    const float hi = e + th;                               // |e| >= 1 > |th| or e == 0: (e - hi) + th is the exact rounding error
    const float t2 = ((e - hi) + th) + tl;
    const float head = hi + t2;
    *tail = (hi - head) + t2;
    return head;
}}
// asin(s) = s + s z P(z) for z = s^2 <= 1/4
HNB_HD float f_asin_core(float s, float z) {{
{horner("p", "z", AS)}
    return f_fma(s * z, p, s);
}}
'''

ENTRY = f'''// ---- binary32 entry points (what WGSL `sin(x)` etc. mean in this framework) ----------
HNB_HD bool trig_small(float x) {{ return f_abs(x) <= 65536.0f; }}
HNB_HD float f_sin(float x) {{
    if (!trig_small(x)) return f_trig_big(x, 0);
    float s, c; f_sincos_small(x, &s, &c);
    return s;
}}
HNB_HD float f_cos(float x) {{
    if (!trig_small(x)) return f_trig_big(x, 1);
    float s, c; f_sincos_small(x, &s, &c);
    return c;
}}
HNB_HD float f_tan(float x) {{
    if (!trig_small(x)) return f_trig_big(x, 2);
    float s, c; f_sincos_small(x, &s, &c);
    return s / c;
}}
HNB_HD float f_exp(float x) {{
    if (x != x) return x;
    const float xc = x > 90.0f ? 90.0f : (x < -105.0f ? -105.0f : x);   // (beyond: infinity / zero either way)
    uint32_t ki;
    const float k = f_rint_bits(xc * {hx(K["LOG2E"])}, &ki);
    float r = f_fma(-k, {hx(K["LN2_HI"])}, xc);
    r = f_fma(-k, {hx(K["LN2_LO"])}, r);
    return f_scale2(f_exp_poly(r), (int32_t)ki);
}}
HNB_HD float f_exp2(float x) {{
    if (x != x) return x;
    const float xc = x > 130.0f ? 130.0f : (x < -152.0f ? -152.0f : x);
    uint32_t ki;
    const float k = f_rint_bits(xc, &ki);
    return f_scale2(f_exp2_poly(xc - k), (int32_t)ki);
}}
// log of NaN, a negative number, zero, infinity
HNB_HD float f_log_special(float x, bool* special) {{
    *special = true;
    if (x != x) return x;
    if (x < 0.0f) return f_nan();
    if (x == 0.0f) return -f_inf();
    if (x == f_inf()) return x;
    *special = false;
    return 0.0f;
}}
HNB_HD float f_log(float x) {{
    bool sp; const float sv = f_log_special(x, &sp);
    if (sp) return sv;
    float e, lo;
    const float f = f_log_reduce(x, &e);
    const float s = f_ln1p_parts(f, &lo);
    return f_fma(e, {hx(K["LN2_HI"])}, s + f_fma(e, {hx(K["LN2_LO"])}, lo));   // e ln2 + ln m, ln2 in two pieces
}}
HNB_HD float f_log2(float x) {{
    bool sp; const float sv = f_log_special(x, &sp);
    if (sp) return sv;
    float tl;
    const float hi = f_log2_parts(x, &tl);
    return hi + tl;
}}
// WGSL pow(x, y): defined here as exp2(y * log2 x) for x > 0 (log2 x carried as head + tail); x < 0 -> NaN.
HNB_HD float f_pow(float x, float y) {{
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
}}
// asin / acos; |x| > 1 -> NaN. |x| <= 1/2: t = asin x directly; beyond: t = asin sqrt((1 - |x|) / 2) ((1 - |x|) / 2 is exact) and
// asin |x| = pi/2 - 2 t. One polynomial evaluation either way (selects, no branch: a wave has lanes on both sides).
HNB_HD float f_asin(float x) {{
    const float ax = f_abs(x);
    if (!(ax <= 1.0f)) return f_nan();
    const bool small = ax <= 0.5f;
    const float z = small ? x * x : (1.0f - ax) * 0.5f;
    const float s = small ? x : f_sqrt(z);
    const float t = f_asin_core(s, z);
    const float r = f_fma(-2.0f, t, {hx(K["PIO2_1"])}) + {hx(K["PIO2_2"])};
    return small ? t : (x < 0.0f ? -r : r);
}}
HNB_HD float f_acos(float x) {{
    const float ax = f_abs(x);
    if (!(ax <= 1.0f)) return f_nan();
    const bool small = ax <= 0.5f;
    const float z = small ? x * x : (1.0f - ax) * 0.5f;
    const float s = small ? x : f_sqrt(z);
    const float t = f_asin_core(s, z);
    const float far = x < 0.0f ? f_fma(-2.0f, t, {hx(K["PI_HI"])}) + {hx(K["PI_LO"])} : 2.0f * t;
    return small ? ({hx(K["PIO2_1"])} - t) + {hx(K["PIO2_2"])} : far;
}}
// atan (NaN -> NaN): at most one division. t = |x| <= tan(pi/8): atan t; t <= tan(3 pi/8): pi/4 + atan((t-1)/(t+1)); beyond:
// pi/2 - atan(1/t); atan u = u + u z A(z), z = u^2 <= tan^2(pi/8); pi/4 and pi/2 in two pieces
HNB_HD float f_atan(float x) {{
    if (x != x) return x;
    const bool neg = f_signbit(x);
    const float t = f_abs(x);
    const bool mid = t > {hx(K["TAN_PIO8"])} && t <= {hx(K["TAN_3PIO8"])}, big = t > {hx(K["TAN_3PIO8"])};
    const float num = mid ? t - 1.0f : (big ? -1.0f : t), den = mid ? t + 1.0f : (big ? t : 1.0f);
    const float u = num / den;
    const float z = u * u;
{horner("p", "z", AT)}
    float r = f_fma(u * z, p, u);
    r = (mid ? {hx(K["PIO4_HI"])} : (big ? {hx(K["PIO2_1"])} : 0.0f)) + (r + (mid ? {hx(K["PIO4_LO"])} : (big ? {hx(K["PIO2_2"])} : 0.0f)));
    return neg ? -r : r;
}}
HNB_HD float f_atan2(float y, float x) {{
    if (x != x || y != y) return x + y;
    if (x > 0.0f) return f_atan(y / x);
    if (x < 0.0f) {{
        const float a = f_atan(y / x);
        return f_signbit(y) ? (a - {hx(K["PI_LO"])}) - {hx(K["PI_HI"])} : (a + {hx(K["PI_LO"])}) + {hx(K["PI_HI"])};
    }}
    if (y > 0.0f) return {hx(K["PIO2_1"])};
    if (y < 0.0f) return -{hx(K["PIO2_1"])};
    if (f_signbit(x)) return f_signbit(y) ? -{hx(K["PI_HI"])} : {hx(K["PI_HI"])};
    return y;
}}
'''

LIBM = '''#ifdef ORACLE_LIBM
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
'''


def splice(path, section):
    s = open(path).read()
    a = s.index("// ---- transcendental builtins") if "// ---- transcendental builtins" in s else s.index("// ---- binary64 kernels")
    b = s.index("// ---- conversions")
    open(path, "w").write(s[:a] + section + "\n" + s[b:])


splice(os.path.join(ROOT, "bevy_hanabi_amd", "csrc", "hnb_math.h"), BIG + "\n" + KERNELS + "\n" + ENTRY)
splice(os.path.join(ROOT, "oracle", "oracle_math.h"), BIG + "\n" + LIBM + KERNELS + "\n" + ENTRY + "#endif  /* ORACLE_LIBM */\n")
print("wrote the transcendental section of hnb_math.h and oracle/oracle_math.h")

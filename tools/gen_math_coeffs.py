#!/usr/bin/env python3
"""Development tool (hanabi-math v3): derives the binary32 polynomial coefficients and constants of hnb_math.h / oracle/oracle_math.h with mpmath
(Chebyshev fits of the reduced functions, coefficients rounded to binary32, the fit error printed) and writes tools/math_coeffs.json, which
tools/gen_math_kernels.py turns into the "binary32 kernels" section of both headers. What the rounded coefficients are worth is measured, not
assumed: tools/math_sweep.c runs every kernel over ALL binary32 arguments against the host's binary64 libm (results in profiles/)."""
import json
import os
import struct

import mpmath as mp

mp.mp.prec = 200
def f32(x):
    return struct.unpack('<f', struct.pack('<f', float(x)))[0]
def fit(f, a, b, n):
    c, err = mp.chebyfit(f, [a, b], n, error=True)
    return [f32(x) for x in c[::-1]], err
def hexf(x):
    return float(x).hex() + "f"
out = {}
q = mp.pi/4 + mp.mpf("0.001")
S = lambda z: (mp.sin(mp.sqrt(z))/mp.sqrt(z) - 1)/z if z > mp.mpf('1e-25') else -mp.mpf(1)/6
C = lambda z: (mp.cos(mp.sqrt(z)) - 1 + z/2)/(z*z) if z > mp.mpf('1e-25') else mp.mpf(1)/24
for n in (3,4):
    out[f"SIN_{n}"], e = fit(S, 0, q*q, n); print("sin", n, mp.nstr(e,5))
    out[f"COS_{n}"], e = fit(C, 0, q*q, n); print("cos", n, mp.nstr(e,5))
# exp(r) = 1 + r + r^2 E(r), |r| <= ln2/2
h = mp.log(2)/2 + mp.mpf("0.001")
E = lambda r: (mp.exp(r) - 1 - r)/(r*r) if abs(r) > mp.mpf('1e-25') else mp.mpf(1)/2 + r/6
for n in (4,5,6):
    out[f"EXP_{n}"], e = fit(E, -h, h, n); print("exp", n, mp.nstr(e,5))
# exp2(r) = 1 + r*E2(r), |r| <= 0.5
E2 = lambda r: (mp.power(2, r) - 1)/r if abs(r) > mp.mpf('1e-25') else mp.log(2) + r*mp.log(2)**2/2
for n in (6,7):
    out[f"EXP2_{n}"], e = fit(E2, -mp.mpf("0.501"), mp.mpf("0.501"), n); print("exp2", n, mp.nstr(e,5))
# log1p(f) = f - f^2/2 + f^3 L(f), f in [-1/3, 1/3]
L = lambda f: (mp.log1p(f) - f + f*f/2)/(f**3) if abs(f) > mp.mpf('1e-15') else mp.mpf(1)/3 - f/4
for n in (8,9,10,11,12):
    out[f"LOG_{n}"], e = fit(L, -mp.mpf(1)/3 - mp.mpf("0.001"), mp.mpf(1)/3 + mp.mpf("0.001"), n); print("log", n, mp.nstr(e,5))
# asin(t) = t + t z P(z), z in [0, 0.25]
AS = lambda z: (mp.asin(mp.sqrt(z))/mp.sqrt(z) - 1)/z if z > mp.mpf('1e-25') else mp.mpf(1)/6
for n in (4,5,6):
    out[f"ASIN_{n}"], e = fit(AS, 0, mp.mpf("0.2501"), n); print("asin", n, mp.nstr(e,5))
# atan(u) = u + u z A(z), z in [0, tan^2(pi/8)]
t8 = mp.tan(mp.pi/8)
AT = lambda z: (mp.atan(mp.sqrt(z))/mp.sqrt(z) - 1)/z if z > mp.mpf('1e-25') else -mp.mpf(1)/3
for n in (4,5):
    out[f"ATAN_{n}"], e = fit(AT, 0, t8*t8 + mp.mpf("0.0001"), n); print("atan", n, mp.nstr(e,5))
consts = {
 "TWO_OVER_PI": f32(2/mp.pi),
 "PIO2_1": f32(mp.pi/2),
}
consts["PIO2_2"] = f32(mp.pi/2 - mp.mpf(consts["PIO2_1"]))
consts["PIO2_3"] = f32(mp.pi/2 - mp.mpf(consts["PIO2_1"]) - mp.mpf(consts["PIO2_2"]))
consts["LOG2E"] = f32(1/mp.log(2))
consts["LOG2E_LO"] = f32(1/mp.log(2) - mp.mpf(consts["LOG2E"]))
# ln2 hi with few bits so that k*ln2_hi is exact for |k| < 2^8? with fma no need: use nearest + residual
consts["LN2_HI"] = f32(mp.log(2)); consts["LN2_LO"] = f32(mp.log(2) - mp.mpf(consts["LN2_HI"]))
consts["PI_HI"] = f32(mp.pi); consts["PI_LO"] = f32(mp.pi - mp.mpf(consts["PI_HI"]))
consts["PIO4_HI"] = f32(mp.pi/4); consts["PIO4_LO"] = f32(mp.pi/4 - mp.mpf(consts["PIO4_HI"]))
consts["TAN_PIO8"] = f32(t8); consts["TAN_3PIO8"] = f32(mp.tan(3*mp.pi/8))
out["consts"] = consts
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "math_coeffs.json"), "w"), indent=1)
for k,v in out.items():
    if k!="consts": print(k, [hexf(x) for x in v])
print({k: hexf(v) for k,v in consts.items()})

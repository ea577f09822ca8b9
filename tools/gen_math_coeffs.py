#!/usr/bin/env python3
"""Development tool: derives the polynomial coefficients of hnb_math.h / oracle/oracle_math.h (binary64 kernels of the
transcendental builtins) with mpmath: Chebyshev fits (near-minimax) of the reduced functions, rounded to binary64, with the
resulting approximation error printed. The headers carry the printed hex literals; tests/test_math.py checks the result."""
import mpmath as mp

mp.mp.prec = 200


def fit(f, a, b, n):
    """degree n-1 polynomial (n coefficients, low order first) approximating f on [a, b], and its max abs error"""
    c, err = mp.chebyfit(f, [a, b], n, error=True)
    return [mp.mpf(x) for x in c[::-1]], err


def as_double(x):
    return float(x)


def show(name, coeffs):
    print(f"// {name}")
    for i, c in enumerate(coeffs):
        d = as_double(c)
        print(f"    {d.hex()},  // c{i} = {d!r}")


def max_err(fapprox, fexact, a, b, n=4000, rel=True):
    worst = mp.mpf(0)
    for i in range(n + 1):
        x = a + (b - a) * mp.mpf(i) / n
        e = fexact(x)
        d = abs(fapprox(x) - e)
        if rel and e != 0:
            d /= abs(e)
        worst = max(worst, d)
    return worst


def horner(cs, z):
    r = mp.mpf(0)
    for c in reversed(cs):
        r = r * z + mp.mpf(as_double(c))
    return r


q = (mp.pi / 4 + mp.mpf("0.0001"))   # slack for the rounding of k
# sin(r) = r + r*z*S(z), z = r^2
for n in (5, 6):
    S = lambda z: (mp.sin(mp.sqrt(z)) / mp.sqrt(z) - 1) / z if z else -mp.mpf(1) / 6
    cs, err = fit(S, 0, q * q, n)
    e = max_err(lambda r: r + r * r * r * horner(cs, r * r), mp.sin, mp.mpf("1e-6"), q)
    print(f"sin: {n} coefficients: fit err {mp.nstr(err, 5)}, relative error of sin {mp.nstr(e, 5)} = 2^{mp.nstr(mp.log(e, 2), 5)}")
    show(f"SIN_{n}", cs)
# cos(r) = 1 - z/2 + z^2*C(z)
for n in (5, 6):
    C = lambda z: (mp.cos(mp.sqrt(z)) - 1 + z / 2) / (z * z) if z else mp.mpf(1) / 24
    cs, err = fit(C, 0, q * q, n)
    e = max_err(lambda r: 1 - r * r / 2 + (r * r) ** 2 * horner(cs, r * r), mp.cos, 0, q)
    print(f"cos: {n} coefficients: fit err {mp.nstr(err, 5)}, relative error of cos {mp.nstr(e, 5)} = 2^{mp.nstr(mp.log(e, 2), 5)}")
    show(f"COS_{n}", cs)
# asin(x) = x + x*z*P(z), z = x^2 in [0, 0.25]
for n in (9, 10, 11, 12):
    P = lambda z: (mp.asin(mp.sqrt(z)) / mp.sqrt(z) - 1) / z if z else mp.mpf(1) / 6
    cs, err = fit(P, 0, mp.mpf("0.2501"), n)
    e = max_err(lambda x: x + x * x * x * horner(cs, x * x), mp.asin, mp.mpf("1e-6"), mp.mpf("0.5"))
    print(f"asin: {n} coefficients: fit err {mp.nstr(err, 5)}, relative error {mp.nstr(e, 5)} = 2^{mp.nstr(mp.log(e, 2), 5)}")
    if n in (10, 11, 12):
        show(f"ASIN_{n}", cs)
# exp(r) = 1 + r + r^2*E(r), |r| <= ln2/2
h = mp.log(2) / 2 + mp.mpf("0.0001")
for n in (8, 9, 10):
    E = lambda r: (mp.exp(r) - 1 - r) / (r * r) if r else mp.mpf(1) / 2
    cs, err = fit(E, -h, h, n)
    e = max_err(lambda r: 1 + r + r * r * horner(cs, r), mp.exp, -h, h)
    print(f"exp: {n} coefficients: fit err {mp.nstr(err, 5)}, relative error {mp.nstr(e, 5)} = 2^{mp.nstr(mp.log(e, 2), 5)}")
    show(f"EXP_{n}", cs)
# log1p(r) = r - r^2/2 + r^3*L(r), |r| <= 2^-6 (+ slack: the intervals next to 1 use invc = 1)
for lim, tag in ((mp.mpf(2) ** -6 * mp.mpf("1.02"), "64"),):
    for n in (5, 6, 7):
        L = lambda r: (mp.log1p(r) - r + r * r / 2) / (r ** 3) if r else mp.mpf(1) / 3
        cs, err = fit(L, -lim, lim, n)
        e = max_err(lambda r: r - r * r / 2 + r ** 3 * horner(cs, r), mp.log1p, -lim, lim, n=4001)
        print(f"log1p |r| <= {mp.nstr(lim, 6)}: {n} coefficients: fit err {mp.nstr(err, 5)}, relative error {mp.nstr(e, 5)} = 2^{mp.nstr(mp.log(e, 2), 5)}")
        show(f"LOG1P_{tag}_{n}", cs)
# atan(t) = t + t*z*A(z), |t| <= tan(pi/16)
lim = mp.tan(mp.pi / 16) * mp.mpf("1.0001")
for n in (8, 9, 10):
    Af = lambda z: (mp.atan(mp.sqrt(z)) / mp.sqrt(z) - 1) / z if z else -mp.mpf(1) / 3
    cs, err = fit(Af, 0, lim * lim, n)
    e = max_err(lambda t: t + t ** 3 * horner(cs, t * t), mp.atan, mp.mpf("1e-6"), lim)
    print(f"atan: {n} coefficients: fit err {mp.nstr(err, 5)}, relative error {mp.nstr(e, 5)} = 2^{mp.nstr(mp.log(e, 2), 5)}")
    show(f"ATAN_{n}", cs)

print("\n==== atan on |u| <= tan(pi/8)")
lim = (mp.sqrt(2) - 1) * mp.mpf("1.0001")
for n in (9, 10, 11, 12):
    Af = lambda z: (mp.atan(mp.sqrt(z)) / mp.sqrt(z) - 1) / z if z else -mp.mpf(1) / 3
    cs, err = fit(Af, 0, lim * lim, n)
    e = max_err(lambda t: t + t ** 3 * horner(cs, t * t), mp.atan, mp.mpf("1e-6"), lim)
    print(f"atan8: {n} coefficients: relative error {mp.nstr(e, 5)} = 2^{mp.nstr(mp.log(e, 2), 5)}")
    show(f"ATAN8_{n}", cs)

print("\n==== log table (64 intervals of z in [0.6875, 1.375))")
OFF = 0x3fe6000000000000
import struct
def u2d(u): return struct.unpack("<d", struct.pack("<Q", u))[0]
rows = []
worst_r = 0
for i in range(64):
    lo, hi = u2d(OFF + (i << 46)), u2d(OFF + ((i + 1) << 46))
    if i in (39, 40):
        invc, logc = 1.0, 0.0
    else:
        c = (mp.mpf(lo) + mp.mpf(hi)) / 2
        invc = float(1 / c)
        logc = float(-mp.log(mp.mpf(invc)))
    for z in (lo, hi):
        worst_r = max(worst_r, abs(float(mp.mpf(z) * mp.mpf(invc) - 1)))
    rows.append((invc, logc, lo, hi))
print(f"// max |r| = {worst_r} = 2^{float(mp.log(worst_r, 2)):.3f}")
for i, (invc, logc, lo, hi) in enumerate(rows):
    print(f"    {{{invc.hex()}, {logc.hex()}}},  // {i}: z in [{lo}, {hi})")
for name, v in (("ln2_hi", None),):
    pass
print("pi/2 hi/lo:", float(mp.pi / 2).hex(), float(mp.pi / 2 - mp.mpf(float(mp.pi / 2))).hex())
print("pi hi/lo:", float(mp.pi).hex(), float(mp.pi - mp.mpf(float(mp.pi))).hex())
print("pi/4:", float(mp.pi / 4).hex(), " 2/pi:", float(2 / mp.pi).hex(), " log2e:", float(1 / mp.log(2)).hex(), "ln2:", float(mp.log(2)).hex())

"""hanabi-math (oracle/oracle_math.h == bevy_hanabi_amd/csrc/hnb_math.h): the transcendental
functions are defined as "evaluate in binary64 with + - * / fma only, round once to binary32".
Checked here against numpy's binary64 libm rounded to f32: at most 1 ulp apart (in practice 0),
and exact on the special values WGSL defines. The product copy of the same header is compared
bit-for-bit against this one through tests/test_lowering_cpu.py (host build) and the GPU tests."""
import math

import numpy as np
import pytest

import oracle

RNG = np.random.default_rng(1234)
N = 4000

FN1 = {
    "sin": (0, np.sin, lambda: RNG.uniform(-50, 50, N)),
    "cos": (1, np.cos, lambda: RNG.uniform(-50, 50, N)),
    "tan": (2, np.tan, lambda: RNG.uniform(-1.5, 1.5, N)),
    "exp": (3, np.exp, lambda: RNG.uniform(-80, 80, N)),
    "log": (4, np.log, lambda: np.exp(RNG.uniform(-80, 80, N))),
    "log2": (5, np.log2, lambda: np.exp(RNG.uniform(-80, 80, N))),
    "atan": (6, np.arctan, lambda: RNG.uniform(-1e3, 1e3, N)),
    "asin": (7, np.arcsin, lambda: RNG.uniform(-1, 1, N)),
    "acos": (8, np.arccos, lambda: RNG.uniform(-1, 1, N)),
    "exp2": (9, np.exp2, lambda: RNG.uniform(-120, 120, N)),
    "sqrt": (10, np.sqrt, lambda: np.exp(RNG.uniform(-80, 80, N))),
    "inverseSqrt": (11, lambda x: 1.0 / np.sqrt(x), lambda: np.exp(RNG.uniform(-80, 80, N))),
}


def ulp_diff(a, b):
    a = np.float32(a).view(np.int32).astype(np.int64)
    b = np.float32(b).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


@pytest.mark.parametrize("name", sorted(FN1))
def test_unary_within_one_ulp_of_libm(name):
    code, ref, gen = FN1[name]
    xs = gen().astype(np.float32)
    got = np.array([oracle.math1(code, float(x)) for x in xs], dtype=np.float32)
    want = ref(xs.astype(np.float64)).astype(np.float32)
    finite = np.isfinite(want)
    d = ulp_diff(got[finite], want[finite])
    assert d.max() <= 1, f"{name}: max {d.max()} ulp at x={xs[finite][d.argmax()]!r}"
    assert (np.isfinite(got) == finite).all()


def test_pow_and_atan2_within_one_ulp():
    xs = np.exp(RNG.uniform(-10, 10, N)).astype(np.float32)
    ys = RNG.uniform(-8, 8, N).astype(np.float32)
    got = np.array([oracle.math2(0, float(x), float(y)) for x, y in zip(xs, ys)], dtype=np.float32)
    want = np.power(xs.astype(np.float64), ys.astype(np.float64)).astype(np.float32)
    ok = np.isfinite(want) & (want != 0)
    assert ulp_diff(got[ok], want[ok]).max() <= 1
    a = RNG.uniform(-100, 100, N).astype(np.float32)
    b = RNG.uniform(-100, 100, N).astype(np.float32)
    got = np.array([oracle.math2(1, float(x), float(y)) for x, y in zip(a, b)], dtype=np.float32)
    want = np.arctan2(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
    assert ulp_diff(got, want).max() <= 1


def test_special_values():
    m1, m2 = oracle.math1, oracle.math2
    assert m1(0, 0.0) == 0.0 and m1(1, 0.0) == 1.0 and m1(3, 0.0) == 1.0 and m1(9, 0.0) == 1.0
    assert m1(4, 1.0) == 0.0 and m1(5, 1.0) == 0.0 and m1(5, 8.0) == 3.0 and m1(9, 10.0) == 1024.0
    assert m1(7, 1.0) == float(np.float32(math.pi / 2)) and m1(8, -1.0) == float(np.float32(math.pi)) and m1(8, 1.0) == 0.0
    assert math.isnan(m1(7, 1.5)) and math.isnan(m1(8, -1.5)) and math.isnan(m1(4, -1.0)) and math.isnan(m1(10, -1.0))
    assert m1(4, 0.0) == -math.inf and m1(3, 1000.0) == math.inf and m1(3, -1000.0) == 0.0
    assert m1(10, 4.0) == 2.0 and m1(11, 4.0) == 0.5
    assert m2(0, 2.0, 10.0) == 1024.0 and m2(0, 8.0, 1.0 / 3.0) == 2.0 and m2(0, 0.0, 0.5) == 0.0 and m2(0, 5.0, 0.0) == 1.0
    assert m2(1, 0.0, 1.0) == 0.0 and m2(1, 1.0, 0.0) == float(np.float32(math.pi / 2)) and m2(1, 0.0, -1.0) == float(np.float32(math.pi))
    # WGSL `%` on floats is truncated remainder: sign follows the dividend
    assert m2(2, 5.5, 2.0) == 1.5 and m2(2, -5.5, 2.0) == -1.5 and m2(2, 5.5, -2.0) == 1.5


def test_large_argument_reduction():
    # Cody-Waite/Payne-Hanek range: sin/cos stay within 1 ulp far from zero
    for x in (1e4, 12345.678, 1e6, 3.4e7, -7.7e6):
        x32 = float(np.float32(x))
        for code, ref in ((0, math.sin), (1, math.cos)):
            got, want = np.float32(oracle.math1(code, x32)), np.float32(ref(x32))
            assert ulp_diff(got, want) <= 1, (x, code, got, want)


def test_binary64_kernels_before_the_final_rounding():
    """The binary64 kernels themselves (minimax polynomials + fma, tools/gen_math_coeffs.py) against libm's binary64 functions:
    relative error below 2^-45 everywhere sampled, i.e. about 2^21 times finer than the binary32 result they are rounded to."""
    rng = np.random.default_rng(77)
    n = 20000
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    cases = [(0, f32(rng.uniform(-50, 50, n)), np.sin), (1, f32(rng.uniform(-50, 50, n)), np.cos),
             (7, f32(rng.uniform(-1, 1, n)), np.arcsin), (8, f32(rng.uniform(-1, 1, n)), np.arccos),
             (3, f32(rng.uniform(-80, 80, n)), np.exp), (4, f32(np.exp(rng.uniform(-80, 80, n))), np.log),
             (4, f32(1 + rng.uniform(-0.05, 0.05, n)), np.log), (6, f32(rng.uniform(-1e3, 1e3, n)), np.arctan),
             (6, f32(rng.uniform(-3, 3, n)), np.arctan), (10, np.exp(rng.uniform(-17, -1.3, n)), np.sqrt)]
    for fn, xs, ref in cases:
        got = np.array([oracle.math1d(fn, float(x)) for x in xs])
        want = ref(xs)
        ok = np.isfinite(want) & (want != 0)
        rel = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
        assert rel.max() < 2.0 ** -45, (fn, float(rel.max()), float(xs[ok][rel.argmax()]))


def test_normalize_reciprocal_domain():
    """normalize(v) = v * (1 / length(v)) (hnb_math.h): ADVICE r03 asked whether 1 / length can overflow for a tiny vector where
    v / length would have been finite. It cannot: the squared length is an f32 - zero, or at least 2^-149 - so a non-zero length is at
    least 2^-74.5 and its reciprocal at most 2^74.5. Checked on the extreme vectors with the same f32 operations the kernels use."""
    f = np.float32
    tiny = [f(2.0) ** f(e) for e in (-149, -140, -126, -100, -75, -74, -70, -64)]
    with np.errstate(all="ignore"):
        for t in tiny:
            for v in (np.array([t, 0, 0], f), np.array([t, t, t], f), np.array([t, -t, f(0.5) * t], f)):
                d = f(f(v[0] * v[0]) + f(v[1] * v[1])) + f(v[2] * v[2])
                ln = np.sqrt(f(d))
                if d == 0:      # the squares underflowed: the division form is non-finite as well
                    assert not np.isfinite(v / ln).all() and not np.isfinite(v * (f(1) / ln)).all()
                    continue
                inv = f(1) / ln
                assert np.isfinite(inv) and inv <= f(2.0) ** f(75)
                n = v * inv
                assert np.isfinite(n).all() and abs(float(np.sqrt((n.astype(np.float64) ** 2).sum())) - 1.0) < 0.3   # (subnormal squares lose bits; unit-ish)
                assert np.abs(n - v / ln).max() <= 2 * np.spacing(np.abs(n).max())

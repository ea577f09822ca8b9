"""hanabi-math v3 (oracle/oracle_math.h == bevy_hanabi_amd/csrc/hnb_math.h): the transcendental functions are binary32 kernels
built from + - * / sqrt and fmaf only. Checked here on samples against numpy's binary64 libm rounded to f32 with the bounds the
exhaustive sweep established (tools/math_sweep.c over ALL binary32 arguments, profiles/r04_math_sweep.txt: sin cos asin atan atan2
<= 2 ulp, tan <= 4, exp exp2 log log2 acos <= 1, pow <= 8 and <= 2 where |y log2 x| <= 32), and exact on the special values WGSL
defines. The product copy of the same header is compared bit-for-bit against this one through tests/test_lowering_cpu.py (host
build) and the GPU tests."""
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


MAX_ULP = {"sin": 2, "cos": 2, "tan": 4, "asin": 2, "atan": 2, "inverseSqrt": 1}   # (inverseSqrt: 1 / sqrt(x), two IEEE roundings); every other function: 1


def ulp_diff(a, b):
    a = np.float32(a).view(np.int32).astype(np.int64)
    b = np.float32(b).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


@pytest.mark.parametrize("name", sorted(FN1))
def test_unary_within_its_ulp_bound_of_libm(name):
    code, ref, gen = FN1[name]
    xs = gen().astype(np.float32)
    got = np.array([oracle.math1(code, float(x)) for x in xs], dtype=np.float32)
    want = ref(xs.astype(np.float64)).astype(np.float32)
    finite = np.isfinite(want)
    d = ulp_diff(got[finite], want[finite])
    assert d.max() <= MAX_ULP.get(name, 1), f"{name}: max {d.max()} ulp at x={xs[finite][d.argmax()]!r}"
    assert d.mean() < 0.5, f"{name}: mean distance to the correctly rounded result {d.mean():.3f} ulp"
    assert (np.isfinite(got) == finite).all()


def test_pow_and_atan2_within_their_ulp_bounds():
    xs = np.exp(RNG.uniform(-10, 10, N)).astype(np.float32)
    ys = RNG.uniform(-8, 8, N).astype(np.float32)
    got = np.array([oracle.math2(0, float(x), float(y)) for x, y in zip(xs, ys)], dtype=np.float32)
    want = np.power(xs.astype(np.float64), ys.astype(np.float64)).astype(np.float32)
    ok = np.isfinite(want) & (want != 0) & (np.abs(want) > 1.2e-38)
    d = ulp_diff(got[ok], want[ok])
    assert d.max() <= 8
    moderate = np.abs(ys.astype(np.float64) * np.log2(xs.astype(np.float64)))[ok] <= 32.0
    assert d[moderate].max() <= 2 and d[moderate].mean() < 0.5
    a = RNG.uniform(-100, 100, N).astype(np.float32)
    b = RNG.uniform(-100, 100, N).astype(np.float32)
    got = np.array([oracle.math2(1, float(x), float(y)) for x, y in zip(a, b)], dtype=np.float32)
    want = np.arctan2(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
    assert ulp_diff(got, want).max() <= 2


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
    # |x| <= 65536: the three-term binary32 Cody-Waite reduction (<= 2 ulp); beyond: the binary64 reduction (<= 1 ulp); both sides of the seam
    for x in (1e4, 12345.678, 65535.996, 65536.0, 65536.008, 1e6, 3.4e7, -7.7e6):
        x32 = float(np.float32(x))
        for code, ref in ((0, math.sin), (1, math.cos)):
            got, want = np.float32(oracle.math1(code, x32)), np.float32(ref(x32))
            assert ulp_diff(got, want) <= (2 if abs(x32) <= 65536.0 else 1), (x, code, got, want)
    # |x| > 2^40 is defined as x = 0; NaN / inf -> NaN
    assert oracle.math1(0, 2.0e12) == 0.0 and oracle.math1(1, 2.0e12) == 1.0 and oracle.math1(2, -2.0e12) == 0.0
    assert math.isnan(oracle.math1(0, math.inf)) and math.isnan(oracle.math1(1, -math.inf)) and math.isnan(oracle.math1(2, math.nan))


def test_the_binary64_reduction_before_its_rounding():
    """The one binary64 kernel left - sin / cos of 65536 < |x| <= 2^40 - against libm's binary64 functions: relative error below 2^-45
    everywhere sampled, about 2^21 times finer than the binary32 result it is rounded to."""
    rng = np.random.default_rng(77)
    xs = (rng.uniform(65536.0, 2.0 ** 40, 20000) * rng.choice([-1.0, 1.0], 20000)).astype(np.float32).astype(np.float64)
    for fn, ref in ((0, np.sin), (1, np.cos)):
        got = np.array([oracle.math1d(fn, float(x)) for x in xs])
        want = ref(xs)
        ok = want != 0
        rel = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
        assert rel.max() < 2.0 ** -45, (fn, float(rel.max()), float(xs[ok][rel.argmax()]))


def test_exp_and_pow_saturate_and_round_into_the_subnormals_once():
    m1, m2 = oracle.math1, oracle.math2
    f = lambda v: float(np.float32(v))
    assert m1(3, 88.0) == f(math.exp(88.0)) or abs(m1(3, 88.0) / math.exp(88.0) - 1) < 2e-7
    assert m1(3, 89.0) == math.inf and m1(3, 1e30) == math.inf and m1(3, -104.0) == 0.0 and m1(3, -1e30) == 0.0
    assert m1(9, 127.0) == 2.0 ** 127 and m1(9, 128.0) == math.inf and m1(9, -149.0) == 2.0 ** -149 and m1(9, -150.0) == 0.0 and m1(9, -126.0) == 2.0 ** -126
    assert m1(9, -140.5) == f(2.0 ** -140.5)                       # a subnormal result: one rounding
    assert m2(0, 2.0, 127.0) == 2.0 ** 127 and m2(0, 2.0, 128.0) == math.inf and m2(0, 2.0, -149.0) == 2.0 ** -149 and m2(0, 0.5, 150.0) == 0.0
    assert m2(0, 1e30, 1e30) == math.inf and m2(0, 1e-30, 1e30) == 0.0 and m2(0, 1e30, -1e30) == 0.0
    assert m1(4, f(2.0 ** -140)) == f(math.log(2.0 ** -140)) and m1(5, f(2.0 ** -149)) == -149.0   # subnormal arguments of log / log2


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

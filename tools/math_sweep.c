/* Development tool (hanabi-math v3): every transcendental builtin of oracle/oracle_math.h (== bevy_hanabi_amd/csrc/hnb_math.h in that
 * section) over ALL binary32 arguments against the host's binary64 libm rounded to binary32; pow and atan2 on 4e8 pseudo-random pairs.
 *     gcc -O2 -fopenmp -ffp-contract=off -mfma -Ioracle tools/math_sweep.c -o /tmp/math_sweep -lm && /tmp/math_sweep > profiles/r04_math_sweep.txt
 * (-mfma only makes fmaf an instruction; the library call gives the same bits). About five minutes on 8 cores. */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include "oracle_math.h"

static long ulp(float a, float b) {
    if (a != a && b != b) return 0;
    if (a != a || b != b) return 1L << 40;
    int32_t ia = (int32_t)f2u(a), ib = (int32_t)f2u(b);
    if (ia < 0) ia = (int32_t)0x80000000 - ia;
    if (ib < 0) ib = (int32_t)0x80000000 - ib;
    const long d = (long)ia - (long)ib;
    return d < 0 ? -d : d;
}
typedef float (*f1)(float);
typedef double (*d1)(double);
static void sweep(const char* name, f1 f, d1 ref, float lo, float hi) {
    long worst = 0, n = 0, bad = 0, hist[5] = {0, 0, 0, 0, 0};
    uint32_t worst_x = 0;
#pragma omp parallel
    {
        long w = 0, nn = 0, bs = 0, h[5] = {0, 0, 0, 0, 0};
        uint32_t wx = 0;
#pragma omp for schedule(static)
        for (long i = 0; i < (1L << 32); ++i) {
            const uint32_t u = (uint32_t)i;
            const float x = u2f(u);
            if (x != x || !(x >= lo && x <= hi)) continue;
            const float got = f(x), want = (float)ref((double)x);
            if (want != want || isinf(want)) { if (!((got != got && want != want) || got == want)) bs++; continue; }
            const long d = ulp(got, want);
            nn++; h[d > 4 ? 4 : d]++;
            if (d > w) { w = d; wx = u; }
        }
#pragma omp critical
        { if (w > worst) { worst = w; worst_x = wx; } n += nn; bad += bs; for (int k = 0; k < 5; ++k) hist[k] += h[k]; }
    }
    printf("%-6s [%g, %g]: %ld arguments, max %ld ulp (at x = %a), exact %.4f %%, 1 ulp %.4f %%, 2 ulp %.5f %%, 3 ulp %.6f %%, >= 4 ulp %.6f %%; inf / NaN mismatches %ld\n",
           name, lo, hi, n, worst, u2f(worst_x), 100.0 * hist[0] / n, 100.0 * hist[1] / n, 100.0 * hist[2] / n, 100.0 * hist[3] / n, 100.0 * hist[4] / n, bad);
    fflush(stdout);
}
static uint64_t mix(uint64_t i) { uint64_t h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; return h; }
int main(void) {
    sweep("sin", f_sin, sin, -65536.f, 65536.f);
    sweep("cos", f_cos, cos, -65536.f, 65536.f);
    sweep("tan", f_tan, tan, -65536.f, 65536.f);
    sweep("sin", f_sin, sin, 65536.f, 1099511627776.0f);    /* the binary64 reduction */
    sweep("cos", f_cos, cos, 65536.f, 1099511627776.0f);
    sweep("exp", f_exp, exp, -INFINITY, INFINITY);
    sweep("exp2", f_exp2, exp2, -INFINITY, INFINITY);
    sweep("log", f_log, log, -INFINITY, INFINITY);
    sweep("log2", f_log2, log2, -INFINITY, INFINITY);
    sweep("asin", f_asin, asin, -INFINITY, INFINITY);
    sweep("acos", f_acos, acos, -INFINITY, INFINITY);
    sweep("atan", f_atan, atan, -INFINITY, INFINITY);
    long worst = 0, worst32 = 0, sat_bad = 0;
#pragma omp parallel for reduction(max : worst, worst32) reduction(+ : sat_bad)
    for (long i = 0; i < 400000000L; ++i) {
        const uint64_t h = mix((uint64_t)i);
        const float x = u2f((uint32_t)(h & 0x7fffffffu));
        float y = u2f((uint32_t)(h >> 32));
        if (x != x || y != y || isinf(x) || isinf(y)) continue;
        if (fabsf(y) > 1e4f || fabsf(y) < 1e-4f) y = (float)((int32_t)(h >> 40) % 20000) / 1000.0f;
        const float wf = (float)pow((double)x, (double)y), got = f_pow(x, y);
        if (wf != wf) continue;
        if (isinf(wf) || wf == 0.0f) { if (got != wf && !(isinf(wf) ? got > 3.0e38f : got < 3e-45f)) sat_bad++; continue; }   /* (one step from the overflow / underflow boundary) */
        long d = ulp(got, wf);
        if (fabsf(wf) < 1.2e-38f) d = d > 2 ? d : 0;   /* subnormal results: an ulp there is not a relative statement */
        if (d > worst) worst = d;
        if (fabs((double)y * log2((double)x)) <= 32.0 && d > worst32) worst32 = d;
    }
    printf("pow    4e8 pseudo-random (x, y): max %ld ulp; max %ld ulp where |y log2 x| <= 32; saturation mismatches %ld\n", worst, worst32, sat_bad);
    worst = 0;
#pragma omp parallel for reduction(max : worst)
    for (long i = 0; i < 400000000L; ++i) {
        const uint64_t h = mix((uint64_t)i);
        const float x = u2f((uint32_t)h), y = u2f((uint32_t)(h >> 32));
        if (x != x || y != y) continue;
        const float wf = (float)atan2((double)y, (double)x);
        if (fabsf(wf) < 1.2e-38f) continue;
        const long d = ulp(f_atan2(y, x), wf);
        if (d > worst) worst = d;
    }
    printf("atan2  4e8 pseudo-random (y, x): max %ld ulp\n", worst);
    return 0;
}

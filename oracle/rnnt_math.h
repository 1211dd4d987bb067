/*
 * rnnt_math.h — the exact-order float32 arithmetic shared by rnnt_greedy.c and rnnt_alsd.c.
 * TEST INFRASTRUCTURE (see oracle/__init__.py).  Mirrors reazonspeech_amd/csrc/k_rnnt_common.h
 * operation for operation (only + - * / and fmaf, no libm transcendental), so results can be
 * compared with the HIP kernels bit for bit.  Compile with -ffp-contract=off.
 */
#pragma once
#include <math.h>
#include <stdint.h>

#define SPLITK_LSTM 16  /* K slices of the LSTM gate products */
#define SPLITK_TILE 8   /* K slices of the joint / prediction projections */
#define SPLITK_MAX 16

static inline float rs_expf(float x) {
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = fmaf(p, r2, r) + 1.0f;
    const int ni = (int)n;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)(ni + 127) << 23;
    return y * s.f;
}
static inline float rs_sigmoidf(float x) { return 1.0f / (1.0f + rs_expf(-x)); }
static inline float rs_tanhf(float x) { return 1.0f - 2.0f / (rs_expf(2.0f * x) + 1.0f); }

/* natural log of a positive normal float: x = m * 2^e, m in (sqrt(1/2), sqrt(2)], degree-9 polynomial in m - 1 */
static inline float rs_logf(float x) {
    union { uint32_t u; float f; } s;
    s.f = x;
    int e = (int)(s.u >> 23) - 127;
    s.u = (s.u & 0x007fffffu) | 0x3f800000u;
    float m = s.f;
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float fe = (float)e;
    const float r = m - 1.0f;
    const float z = r * r;
    float p = 7.0376836292e-2f;
    p = fmaf(p, r, -1.1514610310e-1f);
    p = fmaf(p, r, 1.1676998740e-1f);
    p = fmaf(p, r, -1.2420140846e-1f);
    p = fmaf(p, r, 1.4249322787e-1f);
    p = fmaf(p, r, -1.6668057665e-1f);
    p = fmaf(p, r, 2.0000714765e-1f);
    p = fmaf(p, r, -2.4999993993e-1f);
    p = fmaf(p, r, 3.3333331174e-1f);
    float y = (p * r) * z;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    return fmaf(fe, 0.693359375f, r + y);
}
/* log(exp(a) + exp(b)) */
static inline float rs_logaddexpf(float a, float b) {
    const float hi = a >= b ? a : b, lo = a >= b ? b : a;
    return hi + rs_logf(1.0f + rs_expf(lo - hi));
}

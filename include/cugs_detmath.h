/* cugs_detmath.h — deterministic fp32 transcendental functions.
 *
 * The reference calls CUDA's expf / rsqrtf / logf on paths that decide INTEGER
 * outputs (radius -> tile count -> sort keys: projection.cuh:31,68-70; the
 * alpha >= 1/255 and T < 1/255 tests: forward.cu:137-156, backward.cu:136-145).
 * Those CUDA functions are not correctly rounded and cannot be reproduced
 * off NVIDIA hardware, and glibc's / ocml's differ from them and from each
 * other in the last ulp.  A one-ulp difference flips a radius or an alpha test,
 * which is an O(1/255) change in a pixel, so "within 1e-4" needs the decisions
 * to be bit-identical between the HIP kernels and the CPU oracle.
 *
 * This header therefore defines ONE implementation, built only from IEEE-754
 * correctly rounded operations (+ - * / sqrt fma rint, integer bit moves), so
 * that it returns the same bits under gcc on x86-64 and under hipcc on gfx950
 * (both compiled with -ffp-contract=off; the fmaf calls below are the only
 * fused operations).  Accuracy: cugs_expf is within 1 ulp of exp() on
 * [-87, 88] (tests/test_detmath.py checks against libm on 10^6 points).
 *
 * It is included by the product kernels AND by oracle/ (the oracle depends on
 * the product's definition of exp, not the other way round).
 */
#ifndef CUGS_DETMATH_H
#define CUGS_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define CUGS_HD __host__ __device__ __forceinline__
#else
#define CUGS_HD static inline
#endif

CUGS_HD float cugs_bits_to_float(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
CUGS_HD uint32_t cugs_float_to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* 2^k as a float, k in [-126, 127]. */
CUGS_HD float cugs_pow2i(int k) { return cugs_bits_to_float((uint32_t)(k + 127) << 23); }

/* exp(x) for x in [-87.3, 88.7]: Cody-Waite reduction x = k ln2 + r,
 * |r| <= ln2/2, degree-5 minimax polynomial for (e^r - 1 - r)/r^2 (the Cephes
 * expf coefficients), result scaled by 2^k in two exact steps.
 * Precondition (callers guarantee it): x finite and inside the range. */
/* The reduction and the polynomial: returns e^r in [0.70, 1.42] and k (as a float) with x = k ln2 + r. */
CUGS_HD float cugs_expf_mant(float x, float* kf_out) {
    const float kLog2e = 1.44269504088896341f;
    const float kLn2Hi = 0.693359375f;          /* 8 significant bits: k*kLn2Hi is exact */
    const float kLn2Lo = -2.12194440e-4f;
    float kf = rintf(x * kLog2e);
    float r = fmaf(kf, -kLn2Hi, x);
    r = fmaf(kf, -kLn2Lo, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    *kf_out = kf;
    return fmaf(p, r2, r) + 1.0f;
}

CUGS_HD float cugs_expf_core(float x) {
    float kf;
    float e = cugs_expf_mant(x, &kf);
    int k = (int)kf;                            /* in [-126, 128] */
    int k1 = k >> 1;
    int k2 = k - k1;
    return (e * cugs_pow2i(k1)) * cugs_pow2i(k2);
}

/* The exponential of the blend kernels: exp(-q/2) for the quadratic form q >= 0 of a (pixel, Gaussian) pair,
 * clamped below at exp(-6) (alpha = opacity * that < 1/255 there: the pair is skipped whatever the value), q < 0
 * (rounding noise around the centre; the reference skips power > 0) evaluated as q = 0.  Base 2, ten PLAIN vector
 * instructions on gfx950 (no v_med3 / v_min: those cost 1.6 issue slots) where the Cody-Waite route above takes
 * fifteen - the blend loops are bound by instruction issue and evaluate this once per (pixel, Gaussian) step:
 *   u = sat(q / 12): the clamp of the argument to [0, 12] rides on the multiply (v_mul_f32 ... clamp; NaN -> 0);
 *   y = u * YS, YS = -(log2(e)/2) * 12 = -6 log2 e, is never formed:  t = fma(u, YS, 1.5*2^23) leaves round(y) = n
 *   in the low mantissa bits;  f = fma(u, YS, -n) = y - n in [-1/2, 1/2] (one rounding of the exact difference);
 *   2^f by a degree-5 minimax polynomial with constant term 1 (max rel. error 1.7e-7 in fp32 Horner form);  2^n
 *   enters as n added to the exponent field: bits(p) + (bits(t) << 23), the constant part of bits(t) << 23
 *   vanishing mod 2^32.
 * YS is the float nearest -(log2(e)/2) / CUGS_BLEND_KU for the FLOAT CUGS_BLEND_KU = RN(1/12), so that the
 * representation error of 1/12 cancels instead of adding up.
 * Every operation is a single correctly rounded fp32 operation (or integer), the same on the host and on the
 * device: identical bits, hence identical skip / clamp / termination decisions in the oracle and in the kernels.
 * Against exp(): <= 1e-6 relative over the whole range (tests/test_detmath.py); the reference's CUDA expf is a
 * 2-ulp approximation itself, and the bar on everything downstream is 1e-4. */
#define CUGS_BLEND_KU (0x1.555556p-4f)                 /* RN(1/12) = 0.0833333358168602 */
#define CUGS_BLEND_YS (-0x1.14ff58p+3f)                /* RN(-(log2(e)/2) / CUGS_BLEND_KU) = -8.656169891357422 */
CUGS_HD float cugs_blend_exp_q(float q) {
    float u = q * CUGS_BLEND_KU;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __builtin_amdgcn_fmed3f(u, 0.0f, 1.0f);        /* folds into the multiply's clamp bit */
#else
    u = (u > 0.0f) ? ((u < 1.0f) ? u : 1.0f) : 0.0f;   /* NaN -> 0, as the hardware clamp */
#endif
    const float magic = 12582912.0f;                   /* 1.5 * 2^23 */
    float t = fmaf(u, CUGS_BLEND_YS, magic);
    float nf = t - magic;
    float f = fmaf(u, CUGS_BLEND_YS, -nf);
    float p = 0x1.5c37d0p-10f;
    p = fmaf(p, f, 0x1.3d01dep-7f);
    p = fmaf(p, f, 0x1.c6b626p-5f);
    p = fmaf(p, f, 0x1.ebf96ap-3f);
    p = fmaf(p, f, 0x1.62e42ap-1f);
    p = fmaf(p, f, 1.0f);                              /* constant term exactly 1: exp(-0/2) == 1 */
    return cugs_bits_to_float(cugs_float_to_bits(p) + (cugs_float_to_bits(t) << 23));
}

/* exp(x), all inputs: NaN -> NaN, x > 88.72 -> +inf, x < -87.3 -> 0 (results
 * below FLT_MIN are flushed: nothing on this path distinguishes them). */
CUGS_HD float cugs_expf(float x) {
    if (!(x == x)) return x;
    if (x > 88.72f) return INFINITY;
    if (x < -87.3f) return 0.0f;
    return cugs_expf_core(x);
}

/* 1/sqrt(x) as two correctly rounded operations (the reference's rsqrtf,
 * projection.cuh:31, backward.cuh:175, is a 2-ulp approximation on CUDA). */
CUGS_HD float cugs_rsqrtf(float x) { return 1.0f / sqrtf(x); }

/* sigmoid as the reference writes it (projection.cu:121). */
CUGS_HD float cugs_sigmoidf(float x) { return 1.0f / (1.0f + cugs_expf(-x)); }

/* float -> int with CUDA's saturating semantics (C leaves out-of-range UB;
 * projection.cu:176-179 relies on the conversion for far-off-screen means). */
CUGS_HD int cugs_f2i(float x) {
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}

#endif /* CUGS_DETMATH_H */

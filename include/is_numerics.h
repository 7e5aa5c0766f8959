/*
 * is_numerics.h -- deterministic scalar numerics shared by the HIP column-DP core
 * and by the CPU oracle.
 *
 * The reference evaluates `__logf` / `logf` inside its device code
 * (/root/reference/InstanceStixels/src/StixelsKernels.cu:31-42, 88-199), built with
 * --use_fast_math, which is not reproducible off NVIDIA hardware.  The
 * canonical numerics of this project are IEEE fp32 with no contraction; for the logarithm we
 * use ONE implementation, written only with IEEE +,-,*,/ on binary64 and integer bit
 * manipulation, so that host gcc and gfx950 hipcc produce bit-identical results.
 *
 * Accuracy: the binary64 evaluation has relative error < 1e-13, so the fp32 result is the
 * correctly rounded logarithm except when the true value lies within ~1e-13 relative of a
 * rounding boundary; tests/test_numerics.py pins |is_logf - libm logf| <= 1 ulp.
 *
 * Plain C99 / C++ / HIP.  Compile every user with -ffp-contract=off.
 */
#ifndef IS_NUMERICS_H_
#define IS_NUMERICS_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define IS_HD __host__ __device__ __forceinline__
#else
#define IS_HD static inline
#endif

IS_HD uint64_t is_bits_f64(double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof(u));
    return u;
}
IS_HD double is_f64_bits(uint64_t u) {
    double x;
    memcpy(&x, &u, sizeof(x));
    return x;
}
IS_HD uint32_t is_bits_f32(float x) {
    uint32_t u;
    memcpy(&u, &x, sizeof(u));
    return u;
}
IS_HD float is_f32_bits(uint32_t u) {
    float x;
    memcpy(&x, &u, sizeof(x));
    return x;
}

/* Natural logarithm of an fp32 value, fp32 result.
 * log(x) with x = 2^k * m, m in [sqrt(1/2), sqrt(2)):
 *   t = (m-1)/(m+1),  log m = 2 t (1 + t^2/3 + t^4/5 + ... + t^16/17),  |t| <= 0.1716.
 * Special values follow C99 logf: log(+-0) = -inf, log(x<0) = NaN, log(+inf) = +inf,
 * log(NaN) = NaN, log(1) = +0. */
IS_HD float is_logf(float x) {
    const uint32_t ix = is_bits_f32(x);
    if ((ix & 0x7fffffffu) == 0u) return is_f32_bits(0xff800000u);          /* +-0 -> -inf */
    if ((ix & 0x7fffffffu) > 0x7f800000u) return is_f32_bits(0x7fc00000u);  /* NaN */
    if (ix & 0x80000000u) return is_f32_bits(0x7fc00000u);                  /* x < 0 -> NaN */
    if (ix == 0x7f800000u) return x;                                        /* +inf */

    /* exact widening: fp32 subnormals are normal binary64 numbers */
    const uint64_t dx = is_bits_f64((double)x);
    int k = (int)((dx >> 52) & 0x7ffu) - 1023;
    double m = is_f64_bits((dx & 0x000fffffffffffffull) | 0x3ff0000000000000ull); /* [1,2) */
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        k = k + 1;
    }
    const double t = (m - 1.0) / (m + 1.0);
    const double z = t * t;
    double p = 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    const double logm = 2.0 * t * p;
    const double r = (double)k * 0.6931471805599453 + logm;
    return (float)r;
}

#endif /* IS_NUMERICS_H_ */

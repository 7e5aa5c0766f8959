/*
 * is_numerics.h -- deterministic scalar numerics shared by the HIP column-DP core
 * and by the CPU oracle.
 *
 * The reference evaluates `__logf` / `logf` inside its device code
 * (/root/reference/InstanceStixels/src/StixelsKernels.cu:31-42, 88-199), built with
 * --use_fast_math, which is not reproducible off NVIDIA hardware.  The
 * canonical numerics of this project are IEEE fp32 with no contraction; for the logarithm we
 * use ONE implementation, written only with IEEE +,-,* on binary64, a 32-entry table of
 * binary64 literals and integer bit manipulation, so that host gcc and gfx950 hipcc produce
 * bit-identical results.  It sits on the serial critical path of the pairwise DP (two calls per
 * row), hence table + short polynomial and no division.
 *
 * Accuracy: the binary64 evaluation has relative error < 3e-10 of the result (worst case next
 * to x = 1, far below half an fp32 ulp = 6e-8), so the fp32 result is the correctly rounded
 * logarithm except when the true value lies that close to a rounding boundary;
 * tests/test_numerics.py pins |is_logf - libm logf| <= 1 ulp.
 *
 * Plain C99 / C++ / HIP.  Compile every user with -ffp-contract=off.
 */
#ifndef IS_NUMERICS_H_
#define IS_NUMERICS_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define IS_HD __host__ __device__ __forceinline__
#define IS_TABLE_QUAL __device__ __constant__
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

/* log(x), x = 2^k * m, m in [1, 2): i = top 5 mantissa bits selects the centre
 * c_i = 1 + (i + 0.5)/32 (c_0 = 1 so that nothing cancels next to x = 1);
 * r = m * (1/c_i) - 1, |r| < 1/32;  log m = log c_i + (r - r^2/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6).
 * Tables: IS_LOG_INVC[i] = RN(1/c_i), IS_LOG_LOGC[i] = log(1/IS_LOG_INVC[i]) (hex literals). */
#define IS_LOG_TABLE_BITS 5
/* The two 32-entry tables as literals.  `is_log_tables` copies them (e.g. into LDS, so that the
 * serial pairwise chain does not wait on global memory); `is_logf_t` evaluates with caller-
 * provided tables; `is_logf` uses the literals directly.  All three give identical bits. */
#define IS_LOG_TABLE_SIZE (1 << IS_LOG_TABLE_BITS)
IS_HD void is_log_tables(double* invc, double* logc) {
    const double INVC[32] = {
        0x1.0000000000000p+0,
        0x1.e9131abf0b767p-1,
        0x1.dae6076b981dbp-1,
        0x1.cd85689039b0bp-1,
        0x1.c0e070381c0e0p-1,
        0x1.b4e81b4e81b4fp-1,
        0x1.a98ef606a63bep-1,
        0x1.9ec8e951033d9p-1,
        0x1.948b0fcd6e9e0p-1,
        0x1.8acb90f6bf3aap-1,
        0x1.8181818181818p-1,
        0x1.78a4c8178a4c8p-1,
        0x1.702e05c0b8170p-1,
        0x1.6816816816817p-1,
        0x1.6058160581606p-1,
        0x1.58ed2308158edp-1,
        0x1.51d07eae2f815p-1,
        0x1.4afd6a052bf5bp-1,
        0x1.446f86562d9fbp-1,
        0x1.3e22cbce4a902p-1,
        0x1.3813813813814p-1,
        0x1.323e34a2b10bfp-1,
        0x1.2c9fb4d812ca0p-1,
        0x1.27350b8812735p-1,
        0x1.21fb78121fb78p-1,
        0x1.1cf06ada2811dp-1,
        0x1.1811811811812p-1,
        0x1.135c81135c811p-1,
        0x1.0ecf56be69c90p-1,
        0x1.0a6810a6810a7p-1,
        0x1.0624dd2f1a9fcp-1,
        0x1.0204081020408p-1};
    const double LOGC[32] = {
        0x0.0p+0,
        0x1.77458f632dcfcp-5,
        0x1.341d7961bd1d1p-4,
        0x1.a926d3a4ad563p-4,
        0x1.0d77e7cd08e59p-3,
        0x1.44d2b6ccb7d1ep-3,
        0x1.7ab890210d909p-3,
        0x1.af3c94e80bff3p-3,
        0x1.e27076e2af2e6p-3,
        0x1.0a324e27390e3p-2,
        0x1.22941fbcf7966p-2,
        0x1.3a64c556945eap-2,
        0x1.51aad872df82dp-2,
        0x1.686c81e9b14afp-2,
        0x1.7eaf83b82afc1p-2,
        0x1.947941c2116fbp-2,
        0x1.a9cec9a9a084ap-2,
        0x1.beb4d9da71b79p-2,
        0x1.d32fe7e00ebd5p-2,
        0x1.e744261d6878ap-2,
        0x1.faf588f78f31cp-2,
        0x1.0723e5c1cdf42p-1,
        0x1.109f39e2d4c97p-1,
        0x1.19ee6b467c96fp-1,
        0x1.23130d7bebf43p-1,
        0x1.2c0e9ed448e8cp-1,
        0x1.34e289d9ce1d2p-1,
        0x1.3d9026a7156fbp-1,
        0x1.4618bc21c5ec2p-1,
        0x1.4e7d811b75bb0p-1,
        0x1.56bf9d5b3f399p-1,
        0x1.5ee02a9241675p-1};
    for (int i = 0; i < IS_LOG_TABLE_SIZE; i++) {
        invc[i] = INVC[i];
        logc[i] = LOGC[i];
    }
}
IS_HD void is_log_table(int i, double* invc, double* logc) {
    const double INVC[32] = {
        0x1.0000000000000p+0,
        0x1.e9131abf0b767p-1,
        0x1.dae6076b981dbp-1,
        0x1.cd85689039b0bp-1,
        0x1.c0e070381c0e0p-1,
        0x1.b4e81b4e81b4fp-1,
        0x1.a98ef606a63bep-1,
        0x1.9ec8e951033d9p-1,
        0x1.948b0fcd6e9e0p-1,
        0x1.8acb90f6bf3aap-1,
        0x1.8181818181818p-1,
        0x1.78a4c8178a4c8p-1,
        0x1.702e05c0b8170p-1,
        0x1.6816816816817p-1,
        0x1.6058160581606p-1,
        0x1.58ed2308158edp-1,
        0x1.51d07eae2f815p-1,
        0x1.4afd6a052bf5bp-1,
        0x1.446f86562d9fbp-1,
        0x1.3e22cbce4a902p-1,
        0x1.3813813813814p-1,
        0x1.323e34a2b10bfp-1,
        0x1.2c9fb4d812ca0p-1,
        0x1.27350b8812735p-1,
        0x1.21fb78121fb78p-1,
        0x1.1cf06ada2811dp-1,
        0x1.1811811811812p-1,
        0x1.135c81135c811p-1,
        0x1.0ecf56be69c90p-1,
        0x1.0a6810a6810a7p-1,
        0x1.0624dd2f1a9fcp-1,
        0x1.0204081020408p-1};
    const double LOGC[32] = {
        0x0.0p+0,
        0x1.77458f632dcfcp-5,
        0x1.341d7961bd1d1p-4,
        0x1.a926d3a4ad563p-4,
        0x1.0d77e7cd08e59p-3,
        0x1.44d2b6ccb7d1ep-3,
        0x1.7ab890210d909p-3,
        0x1.af3c94e80bff3p-3,
        0x1.e27076e2af2e6p-3,
        0x1.0a324e27390e3p-2,
        0x1.22941fbcf7966p-2,
        0x1.3a64c556945eap-2,
        0x1.51aad872df82dp-2,
        0x1.686c81e9b14afp-2,
        0x1.7eaf83b82afc1p-2,
        0x1.947941c2116fbp-2,
        0x1.a9cec9a9a084ap-2,
        0x1.beb4d9da71b79p-2,
        0x1.d32fe7e00ebd5p-2,
        0x1.e744261d6878ap-2,
        0x1.faf588f78f31cp-2,
        0x1.0723e5c1cdf42p-1,
        0x1.109f39e2d4c97p-1,
        0x1.19ee6b467c96fp-1,
        0x1.23130d7bebf43p-1,
        0x1.2c0e9ed448e8cp-1,
        0x1.34e289d9ce1d2p-1,
        0x1.3d9026a7156fbp-1,
        0x1.4618bc21c5ec2p-1,
        0x1.4e7d811b75bb0p-1,
        0x1.56bf9d5b3f399p-1,
        0x1.5ee02a9241675p-1};
    *invc = INVC[i];
    *logc = LOGC[i];
}

/* Special values of C99 logf: log(+-0) = -inf, log(x<0) = NaN, log(+inf) = +inf,
 * log(NaN) = NaN, log(1) = +0.  Returns 1 and sets *out when x is one of them. */
IS_HD int is_logf_special(float x, float* out) {
    const uint32_t ix = is_bits_f32(x);
    if ((ix & 0x7fffffffu) == 0u) { *out = is_f32_bits(0xff800000u); return 1; }         /* +-0 */
    if ((ix & 0x7fffffffu) > 0x7f800000u) { *out = is_f32_bits(0x7fc00000u); return 1; } /* NaN */
    if (ix & 0x80000000u) { *out = is_f32_bits(0x7fc00000u); return 1; }                 /* x < 0 */
    if (ix == 0x7f800000u) { *out = x; return 1; }                                       /* +inf */
    if (ix == 0x3f800000u) { *out = 0.0f; return 1; }                                    /* 1 */
    return 0;
}
IS_HD int is_logf_index(float x) { /* table index of a positive finite x */
    const uint64_t dx = is_bits_f64((double)x);
    return (int)((dx >> (52 - IS_LOG_TABLE_BITS)) & ((1u << IS_LOG_TABLE_BITS) - 1u));
}
IS_HD float is_logf_eval(float x, double invc, double logc) { /* positive finite x, its table entry */
    /* exact widening: fp32 subnormals are normal binary64 numbers */
    const uint64_t dx = is_bits_f64((double)x);
    const int k = (int)((dx >> 52) & 0x7ffu) - 1023;
    const double m = is_f64_bits((dx & 0x000fffffffffffffull) | 0x3ff0000000000000ull); /* [1,2) */
    const double r = m * invc - 1.0;
    double p = -1.0 / 6.0;
    p = p * r + 1.0 / 5.0;
    p = p * r - 1.0 / 4.0;
    p = p * r + 1.0 / 3.0;
    p = p * r - 1.0 / 2.0;
    p = p * r + 1.0;
    p = p * r;
    const double res = ((double)k * 0x1.62e42fefa39efp-1 + logc) + p;
    return (float)res;
}

/* Natural logarithm with caller-provided copies of the tables. */
IS_HD float is_logf_t(float x, const double* invc_tab, const double* logc_tab) {
    float special;
    if (is_logf_special(x, &special)) return special;
    const int i = is_logf_index(x);
    return is_logf_eval(x, invc_tab[i], logc_tab[i]);
}

/* Natural logarithm of an fp32 value, fp32 result. */
IS_HD float is_logf(float x) {
    float special;
    if (is_logf_special(x, &special)) return special;
    double invc, logc;
    is_log_table(is_logf_index(x), &invc, &logc);
    return is_logf_eval(x, invc, logc);
}

#endif /* IS_NUMERICS_H_ */

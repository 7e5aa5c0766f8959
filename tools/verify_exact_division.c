/*
 * verify_exact_division.c -- exhaustive proof-by-enumeration that the FMA-corrected
 * reciprocal multiplication used by the DP kernels equals IEEE fp32 division a / h for every
 * integer divisor h in [1, HMAX] and EVERY fp32 mantissa of a (one binade suffices: all steps
 * are exact under scaling by powers of two as long as nothing under/overflows; the kernels take
 * the plain-division slow path outside [2^-100, 2^100], see is_kernels.hip `fast_div`).
 *
 *   r  = RN(1/h)                       (table, computed with a true division)
 *   q0 = RN(a*r); e0 = fma(-q0,h,a); q1 = fma(e0,r,q0)            -- variant 1
 *                 e1 = fma(-q1,h,a); q2 = fma(e1,r,q1)            -- variant 2
 *
 * Build: gcc -O3 -march=native -fopenmp -ffp-contract=off tools/verify_exact_division.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
    const int hmax = argc > 1 ? atoi(argv[1]) : 2048;
    long long bad1 = 0, bad2 = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : bad1, bad2)
    for (int h = 1; h <= hmax; h++) {
        const float hf = (float)h;
        const float r = 1.0f / hf;
        long long b1 = 0, b2 = 0;
        for (uint32_t m = 0; m < (1u << 23); m++) {
            const uint32_t bits = 0x3f800000u | m; /* a in [1, 2) */
            float a;
            memcpy(&a, &bits, 4);
            const float want = a / hf;
            const float q0 = a * r;
            const float e0 = fmaf(-q0, hf, a);
            const float q1 = fmaf(e0, r, q0);
            const float e1 = fmaf(-q1, hf, a);
            const float q2 = fmaf(e1, r, q1);
            b1 += (q1 != want);
            b2 += (q2 != want);
        }
        bad1 += b1;
        bad2 += b2;
    }
    printf("h in [1,%d] x 2^23 mantissas: variant1 mismatches %lld, variant2 mismatches %lld\n",
           hmax, bad1, bad2);
    return (bad2 != 0);
}

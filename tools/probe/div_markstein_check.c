/* Exhaustive check of the division the dJPEG kernels use for the IJG tables (csrc/djpeg.hip, div_q<false>):
 *     rc = RN(1 / q);  y0 = RN(x rc);  r = RN(x - q y0) (one fma, exact);  y = RN(y0 + r rc)
 * against the IEEE quotient RN(x / q), for EVERY integer divisor q in 1 .. 255 (quantisation table entries, and the 255 of the
 * final colour scaling) and EVERY float mantissa of x.  The sequence is homogeneous in x, so one binade covers all x whose
 * quotient does not underflow; the sign is symmetric.
 *     gcc -O2 -fopenmp -ffp-contract=off div_markstein_check.c -lm -o chk && ./chk [mantissa stride, default 1]
 * prints the number of mismatches (0) - 255 x 2^23 divisions, ~15 core-seconds.  tests/test_host_logic.py runs it strided. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
    long bad_total = 0, plain_total = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : bad_total, plain_total)
    for (int qi = 1; qi <= 255; ++qi) {
        volatile float qv = (float)qi;
        const float q = qv, rc = 1.0f / q;
        long bad = 0, plain = 0;
        for (uint32_t m = 0; m < (1u << 23); m += stride) {
            const uint32_t bits = 0x3f800000u | m;
            float x;
            memcpy(&x, &bits, 4);
            const float ref = x / q;
            const float y0 = x * rc;
            const float r = fmaf(-y0, q, x);
            const float y = fmaf(r, rc, y0);
            bad += y != ref;
            plain += y0 != ref;
        }
        bad_total += bad;
        plain_total += plain;
    }
    printf("mismatches %ld (x * rc alone: %ld)\n", bad_total, plain_total);
    return bad_total != 0;
}

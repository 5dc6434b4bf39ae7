/* asan_check.c -- exercises oracle/sdsp_oracle.c under AddressSanitizer + UBSan (CPU only; GPU
 * ASan is not available on the pool).  TEST INFRASTRUCTURE ONLY.  Built and run by
 * tests/test_oracle_sanitizers.py:  gcc -fsanitize=address,undefined asan_check.c sdsp_oracle.c */
#include <stdio.h>
#include <stdlib.h>

#include "sdsp_oracle.h"

int main(void)
{
    const unsigned sizes[] = { 2, 4, 8, 64, 256, 1024, 4096 };
    for (unsigned si = 0; si < sizeof(sizes) / sizeof(sizes[0]); si++) {
        const unsigned n = sizes[si];
        double *d = (double *)malloc(sizeof(double) * 2 * n * 3);
        for (unsigned i = 0; i < 2 * n * 3; i++)
            d[i] = (double)((i * 2654435761u) % 1000) / 500.0 - 1.0;
        for (int radix = 2; radix <= 4; radix += 2) {
            if (radix == 4 && !sdsp_oracle_is_power_of_4(n))
                continue;
            for (int rev = 0; rev < 2; rev++) {
                sdsp_oracle_fft_plan *p = sdsp_oracle_fft_plan_create(n, radix, rev);
                if (!p)
                    return 2;
                sdsp_oracle_fft_exec(p, d, 3);
                sdsp_oracle_fft_plan_destroy(p);
            }
        }
        unsigned *lut = (unsigned *)malloc(sizeof(unsigned) * n);
        sdsp_oracle_calc_swap_lookup(n, 2, lut);
        free(lut);
        free(d);
    }
    double *w = (double *)malloc(sizeof(double) * 2 * 64 * 6);
    sdsp_oracle_calc_wcoeffs(64, 1, w);
    free(w);
    for (unsigned m = 2; m <= SDSP_ORACLE_MAX_SECTIONS; m += 2) {
        sdsp_oracle_iir f;
        if (sdsp_oracle_iir_init(&f, m))
            return 3;
        double x[100];
        for (int kind = 0; kind < 4; kind++) {
            if (kind == 2)
                sdsp_oracle_iir_set_hp_coeff(&f, 2e3, 39e3, 1.0);
            else if (kind == 3)
                sdsp_oracle_iir_set_bp_coeff(&f, 2e3, 39e3, 0.8, 1.0);
            else
                sdsp_oracle_iir_set_lp_coeff(&f, 2e3, 39e3, 1.0);
            sdsp_oracle_iir_preload_filter(&f, 3.0);
            for (int i = 0; i < 100; i++)
                x[i] = i == 0;
            sdsp_oracle_iir_process(&f, kind, x, 100);
            sdsp_oracle_iir_process(&f, kind, x, 0);
        }
    }
    if (sdsp_oracle_fft_plan_create(96, 2, 0) || sdsp_oracle_fft_plan_create(2048, 4, 0))
        return 4;
    puts("sanitizers clean");
    return 0;
}

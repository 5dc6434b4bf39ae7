/*
 * sdsp_oracle.h -- CPU restatement of the simpledsp FFT / cascaded-biquad hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker (or as the reported CPU baseline), never as the thing shipped or measured as
 * the GPU path.  The product (simpledsp_amd/, include/) never links or imports it.
 *
 * Every function cites the reference lines it restates (paths relative to the reference
 * checkout, include/sdsp/...).  Parity of this restatement is PINNED: tests/test_oracle_*.py
 * check it bit-for-bit against outputs of the real reference (built by oracle/Makefile into
 * oracle/_ref/ when /root/reference is present; stored as tests/golden/ fixtures otherwise)
 * and against the reference's own known-answer tests and Octave CSV fixtures.
 */
#ifndef SDSP_ORACLE_H
#define SDSP_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDSP_ORACLE_MAX_SECTIONS 32

/* ---- size helpers: fft.h:12-43 ---- */
unsigned sdsp_oracle_log2(unsigned num);
unsigned sdsp_oracle_log4(unsigned num);
int sdsp_oracle_is_power_of_2(unsigned num);
int sdsp_oracle_is_power_of_4(unsigned num);

/* ---- tables: fft.h:148-256 ---- */
/* one row of calc_trigs<N,T>: row i (period 2^(i+1)), full length n. is_cos: 1 cosine, 0 sine */
int sdsp_oracle_calc_trig_row(unsigned n, unsigned i, int is_cos, double *out);
/* calc_wCoeffs<N,T>: log2(n) rows of n interleaved complex doubles */
int sdsp_oracle_calc_wcoeffs(unsigned n, int reverse, double *out);
unsigned sdsp_oracle_digit_reverse(unsigned n, unsigned base, unsigned x);
int sdsp_oracle_calc_swap_lookup(unsigned n, unsigned base, unsigned *out);

/* ---- FFT: fft.h:258-360 ---- */
typedef struct sdsp_oracle_fft_plan sdsp_oracle_fft_plan;
/* radix 2|4; reverse 0 = forward_fft, 1 = reverse_fft (conjugate twiddles, 1/N scale) */
sdsp_oracle_fft_plan *sdsp_oracle_fft_plan_create(unsigned n, int radix, int reverse);
void sdsp_oracle_fft_plan_destroy(sdsp_oracle_fft_plan *plan);
/* in place over `batch` contiguous transforms of n interleaved complex doubles */
int sdsp_oracle_fft_exec(const sdsp_oracle_fft_plan *plan, double *data, size_t batch);

/* ---- cascaded biquads: casc_2o_iir.h ---- */
typedef struct {
    unsigned m;       /* sections (m_t), even */
    int pos;          /* m_pos   casc_2o_iir.h:11 */
    double gain;      /* m_gain  :13 */
    int f_type;       /* m_f_type :20 (filter_type.h:6 values) */
    double mem[(SDSP_ORACLE_MAX_SECTIONS + 1) * 3]; /* m_mem[m+1][3] :15 */
    double b[SDSP_ORACLE_MAX_SECTIONS * 3];         /* m_b_coeff[m][3] :17 */
    double a[SDSP_ORACLE_MAX_SECTIONS * 3];         /* m_a_coeff[m][3] :18 */
} sdsp_oracle_iir;

int sdsp_oracle_iir_init(sdsp_oracle_iir *f, unsigned m);
void sdsp_oracle_iir_copy_coeff_from(sdsp_oracle_iir *f, const sdsp_oracle_iir *other);
int sdsp_oracle_iir_set_lp_coeff(sdsp_oracle_iir *f, double f0, double fs, double gain_in);
int sdsp_oracle_iir_set_hp_coeff(sdsp_oracle_iir *f, double f0, double fs, double gain_in);
int sdsp_oracle_iir_set_bp_coeff(sdsp_oracle_iir *f, double f0, double fs, double q, double gain_in);
/* band-stop: README.md:15 TODO, no reference code -- pinned to scipy, see the .c file */
int sdsp_oracle_iir_set_bs_coeff(sdsp_oracle_iir *f, double f0, double fs, double q, double gain_in);
void sdsp_oracle_iir_preload_filter(sdsp_oracle_iir *f, double value);
/* kind 0: casc_2o_iir::process (reads b); 1/2/3: casc_2o_iir_{lp,hp,bp}::process (b folded) */
int sdsp_oracle_iir_process(sdsp_oracle_iir *f, int kind, double *data, size_t n);

/* ---- FIR filter: README.md:16 TODO, NO reference code -> parity with the reference unpinned; pinned to
 * scipy.signal.firwin / lfilter in tests/test_oracle_fir.py ---- */
int sdsp_oracle_fir_design(unsigned taps, int filter_type, double f0, double fs, double q, double gain_in,
                           double *h);
/* y[n] = sum_k h[k] x[n-k], ascending k, in place; mem[j] = x[n-1-j] (taps-1 values), updated */
void sdsp_oracle_fir_process(unsigned taps, const double *h, double *mem, double *data, size_t n);

#ifdef __cplusplus
}
#endif
#endif

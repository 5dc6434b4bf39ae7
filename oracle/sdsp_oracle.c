/*
 * sdsp_oracle.c -- plain-C restatement of simpledsp's FFT and cascaded-biquad algorithms.
 *
 * TEST INFRASTRUCTURE ONLY (see sdsp_oracle.h).  Double precision, single threaded, same loop
 * structure and operation order as the reference so that results are bit-identical to the
 * reference built with g++ on x86-64 (no FMA contraction: compile with -ffp-contract=off).
 *
 * The reference builds its twiddle tables at COMPILE time, where GCC folds std::sin/std::cos
 * with correctly-rounded arithmetic (fft.h:79,106 via :169).  To reproduce those table entries
 * bit-for-bit at run time the first-quadrant values are evaluated in binary128 (libquadmath)
 * and rounded once to double; everything else is mirror symmetry exactly as fft.h:174-189.
 */
#include "sdsp_oracle.h"

#include <complex.h>
#include <math.h>
#include <quadmath.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ size helpers */

/* fft.h:12-19 */
unsigned sdsp_oracle_log2(unsigned num)
{
    unsigned ret = 0;
    while ((num = num >> 1) > 0u)
        ret++;
    return ret;
}

/* fft.h:21-28 */
unsigned sdsp_oracle_log4(unsigned num)
{
    unsigned ret = 0;
    while ((num = num >> 2) > 0u)
        ret++;
    return ret;
}

/* fft.h:31-37 */
int sdsp_oracle_is_power_of_2(unsigned num)
{
    if (num == 0)
        return 0;
    return (num & (num - 1)) == 0;
}

/* fft.h:40-43 */
int sdsp_oracle_is_power_of_4(unsigned num)
{
    return sdsp_oracle_is_power_of_2(num) && (sdsp_oracle_log2(num) % 2 == 0);
}

/* ------------------------------------------------------------------ tables */

/* sine_calculator / cosine_calculator, fft.h:67-119 */
static double trig_value0(int is_cos) { return is_cos ? 1.0 : 0.0; }
static double trig_value90(int is_cos) { return is_cos ? 0.0 : 1.0; }
static double trig_sym0(int is_cos) { return is_cos ? 1.0 : -1.0; }
static double trig_sym90(int is_cos) { return is_cos ? -1.0 : 1.0; }
static double trig_value(int is_cos, double rad)
{
    /* correctly rounded, as GCC's constant folder does for fft.h:79 / :106 */
    return is_cos ? (double)cosq((__float128)rad) : (double)sinq((__float128)rad);
}

/* calc_trigs<N,T>, one row: fft.h:148-194 (loop body for row i) */
int sdsp_oracle_calc_trig_row(unsigned n, unsigned i, int is_cos, double *value)
{
    if (!sdsp_oracle_is_power_of_2(n) || n < 2 || i >= sdsp_oracle_log2(n))
        return -1;
    unsigned pow2 = 1u << (i + 1u);

    value[0] = trig_value0(is_cos); /* :157 */
    if (i == 0) {                   /* :161-164 */
        for (size_t j = 1; j < n; j++)
            value[j] = value[j - 1] * -1.0;
        return 0;
    }
    unsigned num = 1u << (i - 1u); /* :167 */
    for (unsigned j = 1; j < num; j++)
        value[j] = trig_value(is_cos, 2 * M_PI * j / pow2); /* :169 */
    value[num] = trig_value90(is_cos);                      /* :172 */

    int dir = -1; /* :175-189 */
    double sign = trig_sym90(is_cos);
    unsigned bouncy = num;
    for (size_t j = (size_t)num + 1; j < n; j++) {
        bouncy += (unsigned)dir;
        value[j] = value[bouncy] * sign;
        if (bouncy == 0) {
            dir = 1;
            sign *= trig_sym0(is_cos);
        } else if (bouncy == num) {
            dir = -1;
            sign *= trig_sym90(is_cos);
        }
    }
    return 0;
}

/* calc_wCoeffs<N,T>: fft.h:197-214.  out = log2(n) rows x n x (re,im) */
int sdsp_oracle_calc_wcoeffs(unsigned n, int reverse, double *out)
{
    if (!sdsp_oracle_is_power_of_2(n) || n < 2)
        return -1;
    double sign = reverse ? -1.0 : 1.0; /* fft.h:123-126,137-140 */
    double *c = (double *)malloc(sizeof(double) * n);
    double *s = (double *)malloc(sizeof(double) * n);
    if (!c || !s) {
        free(c);
        free(s);
        return -2;
    }
    unsigned rows = sdsp_oracle_log2(n);
    for (unsigned i = 0; i < rows; i++) {
        sdsp_oracle_calc_trig_row(n, i, 1, c);
        sdsp_oracle_calc_trig_row(n, i, 0, s);
        for (size_t j = 0; j < n; j++) {
            out[2 * ((size_t)i * n + j)] = c[j];
            out[2 * ((size_t)i * n + j) + 1] = sign * -1.0 * s[j]; /* :210 */
        }
    }
    free(c);
    free(s);
    return 0;
}

/* digit_reverse<N,base>: fft.h:217-236 */
unsigned sdsp_oracle_digit_reverse(unsigned n, unsigned base, unsigned x)
{
    unsigned num_bits = sdsp_oracle_log2(base);
    unsigned ret = 0;
    unsigned shift = sdsp_oracle_log2(n) - num_bits;
    unsigned upper = (base - 1) << shift;
    unsigned lower = base - 1;
    while (upper > lower) {
        ret |= (x & upper) >> shift;
        ret |= (x & lower) << shift;
        upper = upper >> num_bits;
        lower = lower << num_bits;
        shift -= num_bits * 2;
    }
    if (upper == lower)
        ret |= x & upper;
    return ret;
}

/* calc_swap_lookup<N,base>: fft.h:238-256 */
int sdsp_oracle_calc_swap_lookup(unsigned n, unsigned base, unsigned *lut)
{
    if (base == 2 ? !sdsp_oracle_is_power_of_2(n) : !sdsp_oracle_is_power_of_4(n))
        return -1;
    if (n < base)
        return -1;
    for (size_t i = 0; i < n; i++)
        lut[i] = sdsp_oracle_digit_reverse(n, base, (unsigned)i);
    for (size_t i = 1; i + 1 < n; i++) {
        unsigned i2 = lut[i];
        if (i2 != i)
            lut[i2] = i2;
    }
    return 0;
}

/* ------------------------------------------------------------------ FFT */

struct sdsp_oracle_fft_plan {
    unsigned n;
    int radix;
    int reverse;
    unsigned rows;  /* log2(n) */
    double *w;      /* radix 2: rows x n complex (the whole coeff_array); radix 4: last row only */
    unsigned *swap; /* calc_swap_lookup */
};

sdsp_oracle_fft_plan *sdsp_oracle_fft_plan_create(unsigned n, int radix, int reverse)
{
    if (radix == 2) {
        if (!sdsp_oracle_is_power_of_2(n) || n < 2) /* static_assert fft.h:261 */
            return NULL;
    } else if (radix == 4) {
        if (!sdsp_oracle_is_power_of_4(n) || n < 4) /* static_assert fft.h:304 */
            return NULL;
    } else {
        return NULL;
    }
    sdsp_oracle_fft_plan *p = (sdsp_oracle_fft_plan *)calloc(1, sizeof(*p));
    if (!p)
        return NULL;
    p->n = n;
    p->radix = radix;
    p->reverse = reverse ? 1 : 0;
    p->rows = sdsp_oracle_log2(n);
    p->swap = (unsigned *)malloc(sizeof(unsigned) * n);
    size_t wrows = radix == 2 ? p->rows : 1;
    p->w = (double *)malloc(sizeof(double) * 2 * wrows * n);
    if (!p->swap || !p->w) {
        sdsp_oracle_fft_plan_destroy(p);
        return NULL;
    }
    sdsp_oracle_calc_swap_lookup(n, (unsigned)radix, p->swap);

    double sign = reverse ? -1.0 : 1.0;
    double *c = (double *)malloc(sizeof(double) * n);
    double *s = (double *)malloc(sizeof(double) * n);
    for (size_t r = 0; r < wrows; r++) {
        unsigned i = radix == 2 ? (unsigned)r : p->rows - 1; /* coeff_subscript fft.h:309 */
        sdsp_oracle_calc_trig_row(n, i, 1, c);
        sdsp_oracle_calc_trig_row(n, i, 0, s);
        for (size_t j = 0; j < n; j++) {
            p->w[2 * (r * n + j)] = c[j];
            p->w[2 * (r * n + j) + 1] = sign * -1.0 * s[j]; /* fft.h:210 */
        }
    }
    free(c);
    free(s);
    return p;
}

void sdsp_oracle_fft_plan_destroy(sdsp_oracle_fft_plan *p)
{
    if (!p)
        return;
    free(p->w);
    free(p->swap);
    free(p);
}

/* std::complex<double> product as libstdc++ evaluates it for finite operands */
#define CMUL(rr, ri, ar, ai, br, bi)        \
    do {                                    \
        double _ar = (ar), _ai = (ai);      \
        double _br = (br), _bi = (bi);      \
        (rr) = _ar * _br - _ai * _bi;       \
        (ri) = _ar * _bi + _ai * _br;       \
    } while (0)

/* fft_radix2<T,N>: fft.h:258-299 */
static void fft_radix2_one(const sdsp_oracle_fft_plan *p, double *d)
{
    const unsigned n = p->n;
    /* decimation in time: bit-reversal swap of the inputs, :269-273 */
    for (unsigned i = 1; i + 1 < n; i++) {
        unsigned i2 = p->swap[i];
        if (i2 != i) {
            double tr = d[2 * i], ti = d[2 * i + 1];
            d[2 * i] = d[2 * i2];
            d[2 * i + 1] = d[2 * i2 + 1];
            d[2 * i2] = tr;
            d[2 * i2 + 1] = ti;
        }
    }
    /* stages, :276-294 */
    for (unsigned i = 0; i < p->rows; i++) {
        unsigned two_i = 1u << i;
        const double *w = p->w + 2 * (size_t)i * n;
        for (unsigned j = 0; j < n; j += (two_i << 1)) {
            for (unsigned k = 0; k < two_i; k++) {
                unsigned i1 = j + k;
                unsigned i2 = j + k + two_i;
                double tr, ti;
                CMUL(tr, ti, d[2 * i2], d[2 * i2 + 1], w[2 * i1], w[2 * i1 + 1]); /* :286 */
                double v1r = d[2 * i1] + tr, v1i = d[2 * i1 + 1] + ti;            /* :287 */
                double v2r = d[2 * i1] - tr, v2i = d[2 * i1 + 1] - ti;            /* :288 */
                d[2 * i1] = v1r;
                d[2 * i1 + 1] = v1i;
                d[2 * i2] = v2r;
                d[2 * i2 + 1] = v2i;
            }
        }
    }
    if (p->reverse) { /* reverse_fft::ScaleValues :128-132 */
        double sc = 1.0 / n;
        for (unsigned i = 0; i < n; i++) {
            d[2 * i] *= sc;
            d[2 * i + 1] *= sc;
        }
    }
}

/* fft_radix4<T,N>: fft.h:301-360 */
static void fft_radix4_one(const sdsp_oracle_fft_plan *p, double *d)
{
    const unsigned n = p->n;
    const double *w = p->w; /* row coeff_subscript = log2(N)-1, :309 */
    const double sign = p->reverse ? -1.0 : 1.0;
    const unsigned stages = sdsp_oracle_log4(n);

    for (unsigned i = 0; i < stages; i++) {
        unsigned group = n / (4u << (2u * i));                  /* :312 */
        unsigned four_n = i > 0 ? 1u << ((i - 1u) * 2u) : 0;    /* :313 */
        unsigned j2 = 0;
        for (unsigned j = 0; j < n; j += (group << 2)) {
            for (unsigned k = 0; k < group; k++) {
                unsigned idx[4] = { k + j, k + group + j, k + 2 * group + j, k + 3 * group + j };
                double tr[4], ti[4];
                for (int q = 0; q < 4; q++) {
                    unsigned ci = (idx[q] - j) * four_n * (j2 % 4); /* :322-325 */
                    if (ci > 0)                                     /* :327-338 */
                        CMUL(tr[q], ti[q], d[2 * idx[q]], d[2 * idx[q] + 1], w[2 * ci], w[2 * ci + 1]);
                    else {
                        tr[q] = d[2 * idx[q]];
                        ti[q] = d[2 * idx[q] + 1];
                    }
                }
                /* T::Sign() * complex(-imag, real), :339-340 */
                double t2ir = sign * -ti[1], t2ii = sign * tr[1];
                double t4ir = sign * -ti[3], t4ii = sign * tr[3];
                /* :342-345, complex sums evaluated left to right */
                double o1r = ((tr[0] + tr[1]) + tr[2]) + tr[3];
                double o1i = ((ti[0] + ti[1]) + ti[2]) + ti[3];
                double o2r = ((tr[0] - t2ir) - tr[2]) + t4ir;
                double o2i = ((ti[0] - t2ii) - ti[2]) + t4ii;
                double o3r = ((tr[0] - tr[1]) + tr[2]) - tr[3];
                double o3i = ((ti[0] - ti[1]) + ti[2]) - ti[3];
                double o4r = ((tr[0] + t2ir) - tr[2]) - t4ir;
                double o4i = ((ti[0] + t2ii) - ti[2]) - t4ii;
                d[2 * idx[0]] = o1r;
                d[2 * idx[0] + 1] = o1i;
                d[2 * idx[1]] = o2r;
                d[2 * idx[1] + 1] = o2i;
                d[2 * idx[2]] = o3r;
                d[2 * idx[2] + 1] = o3i;
                d[2 * idx[3]] = o4r;
                d[2 * idx[3] + 1] = o4i;
            }
            j2++;
        }
    }
    /* base-4 digit reversal after the stages, :351-355 */
    for (unsigned i = 1; i + 1 < n; i++) {
        unsigned i2 = p->swap[i];
        if (i2 != i) {
            double xr = d[2 * i], xi = d[2 * i + 1];
            d[2 * i] = d[2 * i2];
            d[2 * i + 1] = d[2 * i2 + 1];
            d[2 * i2] = xr;
            d[2 * i2 + 1] = xi;
        }
    }
    if (p->reverse) { /* :359 -> :128-132 */
        double sc = 1.0 / n;
        for (unsigned i = 0; i < n; i++) {
            d[2 * i] *= sc;
            d[2 * i + 1] *= sc;
        }
    }
}

int sdsp_oracle_fft_exec(const sdsp_oracle_fft_plan *p, double *data, size_t batch)
{
    if (!p || !data)
        return -1;
    for (size_t b = 0; b < batch; b++) {
        double *d = data + 2 * (size_t)p->n * b;
        if (p->radix == 2)
            fft_radix2_one(p, d);
        else
            fft_radix4_one(p, d);
    }
    return 0;
}

/* ------------------------------------------------------------------ cascaded biquads */

#define MEM(f, j, i) ((f)->mem[(j) * 3 + (i)])
#define BC(f, j, i) ((f)->b[(j) * 3 + (i)])
#define AC(f, j, i) ((f)->a[(j) * 3 + (i)])

/* casc_2o_iir<m_t>::casc_2o_iir and member initialisers, casc_2o_iir.h:11-26 */
int sdsp_oracle_iir_init(sdsp_oracle_iir *f, unsigned m)
{
    if (m == 0 || m % 2 != 0 || m > SDSP_ORACLE_MAX_SECTIONS) /* static_assert :25 */
        return -1;
    memset(f, 0, sizeof(*f));
    f->m = m;
    f->pos = 0;
    f->gain = 1.0;
    f->f_type = 0;
    return 0;
}

/* copy_coeff_from, casc_2o_iir.h:28-34 (specialised classes: intended meaning of :274-278) */
void sdsp_oracle_iir_copy_coeff_from(sdsp_oracle_iir *f, const sdsp_oracle_iir *o)
{
    f->gain = o->gain;
    memcpy(f->b, o->b, sizeof(f->b));
    memcpy(f->a, o->a, sizeof(f->a));
    f->f_type = o->f_type;
}

/*
 * Butterworth second-order-section design shared by low-pass and high-pass:
 * set_lp_coeff casc_2o_iir.h:168-194 (== :297-321), set_hp_coeff :140-166 (== :355-379).
 * The two differ only in the sign of gamma inside alpha and in the middle numerator tap.
 * Expression order is kept so the doubles come out identical to the reference's.
 */
static void design_lp_hp(sdsp_oracle_iir *f, double f0, double fs, double gain_in, int high)
{
    const double mid_tap = high ? -2.0 : 2.0;
    f->gain = gain_in;
    f->f_type = high ? 2 : 1;
    const double e0 = 2 * M_PI * f0 / fs;
    for (unsigned k = 0; k < f->m; k++) {
        const double dk = 2 * sin((2 * k + 1) * M_PI / (4.0 * f->m));
        const double t = dk * sin(e0) / 2;
        const double beta = (1 - t) / (1 + t) / 2;
        const double gamma = (0.5 + beta) * cos(e0);
        const double alpha = high ? (0.5 + beta + gamma) / 4 : (0.5 + beta - gamma) / 4;
        f->gain *= 2 * alpha;
        BC(f, k, 0) = 1.0;
        BC(f, k, 1) = mid_tap;
        BC(f, k, 2) = 1.0;
        AC(f, k, 0) = 1;
        AC(f, k, 1) = -2 * gamma;
        AC(f, k, 2) = 2 * beta;
    }
}

int sdsp_oracle_iir_set_lp_coeff(sdsp_oracle_iir *f, double f0, double fs, double gain_in)
{
    design_lp_hp(f, f0, fs, gain_in, 0);
    return 0;
}

int sdsp_oracle_iir_set_hp_coeff(sdsp_oracle_iir *f, double f0, double fs, double gain_in)
{
    design_lp_hp(f, f0, fs, gain_in, 1);
    return 0;
}

/* one band-pass pole-pair half: beta/gamma from the pair's centre angle, casc_2o_iir.h:107-116 */
static void bp_half(double dk, double e, double *beta, double *gamma)
{
    double t = dk * sin(e) / 2.0;
    *beta = (1 - t) / (1 + t) / 2.0;
    *gamma = (0.5 + *beta) * cos(e);
}

/* set_bp_coeff, casc_2o_iir.h:82-138 (== :413-467): m/2 pole pairs -> two sections each */
int sdsp_oracle_iir_set_bp_coeff(sdsp_oracle_iir *f, double f0, double fs, double q, double gain_in)
{
    f->gain = gain_in;
    f->f_type = 3;
    const double e0 = 2 * M_PI * f0 / fs;
    const double de = 2 * tan(e0 / (2 * q)) / sin(e0); /* :85,:91 */
    for (unsigned k = 0; k < f->m / 2; k++) {
        const double d = 2 * sin((2 * k + 1) * M_PI / (2.0 * f->m));
        const double aa = (1 + de * de / 4.0) * 2 / d / de;
        const double dk = sqrt(de * d / (aa + sqrt(aa * aa - 1)));
        const double bb = d * de / dk / 2.0;
        const double w = bb + sqrt(bb * bb - 1);
        const double th = tan(e0 / 2.0);
        const double e1 = 2.0 * atan(th / w);
        const double e2 = 2.0 * atan(w * th);
        double beta[2], gamma[2];
        bp_half(dk, e1, &beta[0], &gamma[0]);
        bp_half(dk, e2, &beta[1], &gamma[1]);
        const double sc = sqrt(1 + (w - 1 / w) / dk * (w - 1 / w) / dk); /* :118 */
        const double alpha1 = (0.5 - beta[0]) * sc / 2.0;
        const double alpha2 = (0.5 - beta[1]) * sc / 2.0;
        f->gain *= 4 * alpha1 * alpha2;
        for (unsigned h = 0; h < 2; h++) {
            BC(f, 2 * k + h, 0) = 1.0;
            BC(f, 2 * k + h, 1) = 0;
            BC(f, 2 * k + h, 2) = -1.0;
            AC(f, 2 * k + h, 0) = 1;
            AC(f, 2 * k + h, 1) = -2 * gamma[h];
            AC(f, 2 * k + h, 2) = 2 * beta[h];
        }
    }
    return 0;
}

/*
 * Band-stop design.  NOT a restatement of reference code: the reference only lists it as a TODO
 * (README.md:15), so parity with the reference is UNPINNED for this function; it is pinned instead to
 * scipy.signal.butter(btype='bandstop', output='sos') in tests/test_oracle_iir.py (<= 1e-12 on the
 * impulse responses, the bound testIIR.cpp:59 uses for the other three types).  Written with C99
 * complex arithmetic independently of the product's C++ version.  Parameters as set_bp_coeff:
 * centre f0, -3 dB width f0/q, Butterworth prototype of order m -> m sections.
 */
int sdsp_oracle_iir_set_bs_coeff(sdsp_oracle_iir *f, double f0, double fs, double q, double gain_in)
{
    const double e0 = 2 * M_PI * f0 / fs;
    const double c0 = cos(e0), bw = tan(e0 / (2 * q));
    f->gain = gain_in;
    f->f_type = 4;
    for (unsigned k = 0; k < f->m / 2; k++) {
        const double th = (2 * k + 1) * M_PI / (2.0 * f->m);
        const double complex p = -sin(th) + I * cos(th); /* prototype pole, |p| = 1 */
        /* s = bw (z^2-1)/(z^2-2 c0 z+1) = p  ->  (p-bw) z^2 - 2 p c0 z + (p+bw) = 0 */
        const double complex qa = p - bw, qb = -2.0 * p * c0, qc = p + bw;
        const double complex disc = csqrt(qb * qb - 4.0 * qa * qc);
        const double complex z[2] = { (-qb + disc) / (2.0 * qa), (-qb - disc) / (2.0 * qa) };
        for (unsigned h = 0; h < 2; h++) {
            const unsigned s = 2 * k + h;
            AC(f, s, 0) = 1.0;
            AC(f, s, 1) = -2 * creal(z[h]);
            AC(f, s, 2) = creal(z[h]) * creal(z[h]) + cimag(z[h]) * cimag(z[h]);
            BC(f, s, 0) = 1.0;
            BC(f, s, 1) = -2 * c0;
            BC(f, s, 2) = 1.0;
            f->gain *= (1 + AC(f, s, 1) + AC(f, s, 2)) / (2 - 2 * c0);
        }
    }
    return 0;
}

/* preload_filter, casc_2o_iir.h:197-214 */
void sdsp_oracle_iir_preload_filter(sdsp_oracle_iir *f, double value)
{
    double preload = value * f->gain;
    double mem_vals[(SDSP_ORACLE_MAX_SECTIONS + 1) * 3];
    memset(mem_vals, 0, sizeof(mem_vals));
    for (int i = 0; i < 3; i++)
        mem_vals[0 * 3 + i] = preload;
    if (f->f_type == 1 || f->f_type == 4) { /* low_pass only, :204 (+ band_stop, which also passes DC) */
        for (unsigned j = 1; j < f->m + 1; j++) {
            preload /= 1 + AC(f, j - 1, 1) + AC(f, j - 1, 2);
            preload *= BC(f, j - 1, 0) + BC(f, j - 1, 1) + BC(f, j - 1, 2);
            for (unsigned i = 0; i < 3; i++)
                mem_vals[j * 3 + i] = preload;
        }
    }
    memcpy(f->mem, mem_vals, sizeof(double) * (f->m + 1) * 3);
}

/*
 * process: casc_2o_iir.h:36-80 (kind 0) and process_base :228-263 with
 * process_spec :286-295 (lp, kind 1), :344-353 (hp, kind 2), :402-411 (bp, kind 3).
 */
int sdsp_oracle_iir_process(sdsp_oracle_iir *f, int kind, double *data, size_t n)
{
    if (kind < 0 || kind > 3)
        return -1;
    const int order = 2;
    const unsigned m = f->m;
    int p = f->pos;
    double y[(SDSP_ORACLE_MAX_SECTIONS + 1)][3];
    double b[SDSP_ORACLE_MAX_SECTIONS][3];
    double a[SDSP_ORACLE_MAX_SECTIONS][3];
    for (unsigned j = 0; j <= m; j++)
        for (int i = 0; i < 3; i++)
            y[j][i] = MEM(f, j, i);
    for (unsigned j = 0; j < m; j++)
        for (int i = 0; i < 3; i++) {
            b[j][i] = BC(f, j, i);
            a[j][i] = AC(f, j, i);
        }

    for (size_t s = 0; s < n; s++) {
        y[0][p] = data[s] * f->gain; /* :52 / :242 */
        int d1 = p - 1;
        if (d1 < 0)
            d1 += order + 1;
        int d2 = p - 2;
        if (d2 < 0)
            d2 += order + 1;
        for (unsigned j = 0; j < m; j++) {
            y[j + 1][p] = y[j][p];
            switch (kind) {
            case 0: /* :67-68 */
                y[j + 1][p] += y[j][d1] * b[j][1] - y[j + 1][d1] * a[j][1];
                y[j + 1][p] += y[j][d2] * b[j][2] - y[j + 1][d2] * a[j][2];
                break;
            case 1: /* :292-293 */
                y[j + 1][p] += y[j][d1] + y[j][d1] - y[j + 1][d1] * a[j][1];
                y[j + 1][p] += y[j][d2] - y[j + 1][d2] * a[j][2];
                break;
            case 2: /* :350-351 */
                y[j + 1][p] += -y[j][d1] - y[j][d1] - y[j + 1][d1] * a[j][1];
                y[j + 1][p] += y[j][d2] - y[j + 1][d2] * a[j][2];
                break;
            default: /* :408-409 */
                y[j + 1][p] += -y[j + 1][d1] * a[j][1];
                y[j + 1][p] += -y[j][d2] - y[j + 1][d2] * a[j][2];
                break;
            }
        }
        data[s] = y[m][p]; /* :71 / :254 */
        p++;
        if (p > order)
            p = 0;
    }
    f->pos = p;
    for (unsigned j = 0; j <= m; j++)
        for (int i = 0; i < 3; i++)
            MEM(f, j, i) = y[j][i];
    return 0;
}

/*
 * FIR filter.  NOT a restatement of reference code: the reference lists it as a TODO (README.md:16).
 * Parity with the reference is UNPINNED; the design is pinned to scipy.signal.firwin (Hamming window)
 * and the filter to scipy.signal.lfilter in tests/test_oracle_fir.py.
 * filter_type uses filter_type.h:6's values (1 lp, 2 hp, 3 bp) plus 4 = band stop.
 */
static double sinc_pi(double x)
{
    return x == 0 ? 1.0 : sin(M_PI * x) / (M_PI * x);
}

int sdsp_oracle_fir_design(unsigned taps, int filter_type, double f0, double fs, double q, double gain_in,
                           double *h)
{
    if (taps == 0 || !h || filter_type < 1 || filter_type > 4)
        return -1;
    if ((filter_type == 2 || filter_type == 4) && taps % 2 == 0)
        return -1; /* would need a zero at fs/2 */
    const double nyq = fs / 2;
    double edge[4];
    unsigned ne = 0;
    switch (filter_type) {
    case 1: edge[ne++] = 0; edge[ne++] = f0 / nyq; break;
    case 2: edge[ne++] = f0 / nyq; edge[ne++] = 1; break;
    case 3: edge[ne++] = (f0 - f0 / (2 * q)) / nyq; edge[ne++] = (f0 + f0 / (2 * q)) / nyq; break;
    default:
        edge[ne++] = 0; edge[ne++] = (f0 - f0 / (2 * q)) / nyq;
        edge[ne++] = (f0 + f0 / (2 * q)) / nyq; edge[ne++] = 1;
        break;
    }
    const double alpha = 0.5 * (taps - 1);
    for (unsigned i = 0; i < taps; i++) {
        const double m = i - alpha;
        double v = 0;
        for (unsigned e = 0; e < ne; e += 2)
            v += edge[e + 1] * sinc_pi(edge[e + 1] * m) - edge[e] * sinc_pi(edge[e] * m);
        h[i] = v * (taps == 1 ? 1.0 : 0.54 - 0.46 * cos(2 * M_PI * i / (taps - 1)));
    }
    const double at = edge[0] == 0 ? 0.0 : (edge[1] == 1 ? 1.0 : 0.5 * (edge[0] + edge[1]));
    double s = 0;
    for (unsigned i = 0; i < taps; i++)
        s += h[i] * cos(M_PI * (i - alpha) * at);
    for (unsigned i = 0; i < taps; i++)
        h[i] = h[i] / s * gain_in;
    return 0;
}

void sdsp_oracle_fir_process(unsigned taps, const double *h, double *mem, double *data, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        const double x = data[i];
        double acc = h[0] * x;
        for (unsigned k = 1; k < taps; k++)
            acc = acc + h[k] * mem[k - 1];
        for (unsigned k = taps - 1; k > 1; k--)
            mem[k - 1] = mem[k - 2];
        if (taps > 1)
            mem[0] = x;
        data[i] = acc;
    }
}

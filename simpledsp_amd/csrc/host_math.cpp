// host_math.cpp -- the cold, host-side part of the path: size helpers, the run-time twiddle
// precompute that replaces the reference's compile-time tables, Butterworth section design and
// filter preload.  Double precision, runs once per plan / per filter.
#include "sdsp_hip_internal.h"

#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

namespace sdsp_hip
{
thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

// W_n^j = cos(2 pi j / n) - i * sgn * sin(2 pi j / n), sgn = +1 forward, -1 reverse.
// Same construction idea as the reference's calc_trigs (fft.h:148-194): only the first quadrant
// is evaluated with libm, the other three are exact mirror images, so cos(90 deg) is exactly 0
// and conjugate-symmetric entries are bit-identical.  (The reference does this at compile time
// for every stage row; one row of n entries is all the GPU kernels need: row i of the
// reference table is this row sampled with stride n / 2^(i+1).)
void make_twiddles(uint32_t n, int direction, std::vector<double> &out)
{
    out.assign(2 * (size_t)n, 0.0);
    const double sgn = direction == SDSP_HIP_REVERSE ? -1.0 : 1.0;
    if (n == 1) {
        out[0] = 1.0;
        return;
    }
    if (n == 2) {
        out[0] = 1.0;
        out[2] = -1.0;
        return;
    }
    const uint32_t quarter = n / 4;
    std::vector<double> cq(quarter + 1), sq(quarter + 1);
    cq[0] = 1.0;
    sq[0] = 0.0;
    cq[quarter] = 0.0;
    sq[quarter] = 1.0;
    for (uint32_t j = 1; j < quarter; j++) {
        // same argument recipe as fft.h:169 (double), then long-double libm so that the value
        // rounded to double is the correctly rounded one GCC's constant folder produces
        const double rad = 2 * M_PI * j / n;
        cq[j] = (double)cosl((long double)rad);
        sq[j] = (double)sinl((long double)rad);
    }
    for (uint32_t m = 0; m < n; m++) {
        const uint32_t q = m / quarter, r = m % quarter;
        double c, s;
        switch (q) {
        case 0: c = cq[r]; s = sq[r]; break;
        case 1: c = -cq[quarter - r]; s = sq[quarter - r]; break;
        case 2: c = -cq[r]; s = -sq[r]; break;
        default: c = cq[quarter - r]; s = -sq[quarter - r]; break;
        }
        out[2 * (size_t)m] = c;
        out[2 * (size_t)m + 1] = -sgn * s;
    }
}

// set_lp_coeff / set_hp_coeff: casc_2o_iir.h:168-194 and :140-166.  Expression order kept so
// the coefficients equal the reference's doubles.
static int design_lp_hp(uint32_t m, double f0, double fs, double gain_in, bool high, double *a,
                        double *b, double *gain)
{
    if (m == 0 || m % 2 != 0 || m > SDSP_HIP_MAX_SECTIONS)
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "M must be even! (and <= SDSP_HIP_MAX_SECTIONS)");
    if (!a || !b || !gain)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null output pointer");
    double g = gain_in;
    const double e0 = 2 * M_PI * f0 / fs;
    for (uint32_t k = 0; k < m; k++) {
        const double dk = 2 * std::sin((2 * k + 1) * M_PI / (4.0 * m));
        const double t = dk * std::sin(e0) / 2;
        const double beta = (1 - t) / (1 + t) / 2;
        const double gamma = (0.5 + beta) * std::cos(e0);
        const double alpha = high ? (0.5 + beta + gamma) / 4 : (0.5 + beta - gamma) / 4;
        g *= 2 * alpha;
        b[3 * k + 0] = 1.0;
        b[3 * k + 1] = high ? -2.0 : 2.0;
        b[3 * k + 2] = 1.0;
        a[3 * k + 0] = 1.0;
        a[3 * k + 1] = -2 * gamma;
        a[3 * k + 2] = 2 * beta;
    }
    *gain = g;
    return SDSP_HIP_OK;
}

int design_lp(uint32_t m, double f0, double fs, double gain_in, double *a, double *b, double *gain)
{
    return design_lp_hp(m, f0, fs, gain_in, false, a, b, gain);
}

int design_hp(uint32_t m, double f0, double fs, double gain_in, double *a, double *b, double *gain)
{
    return design_lp_hp(m, f0, fs, gain_in, true, a, b, gain);
}

// set_bp_coeff: casc_2o_iir.h:82-138 -- m/2 pole pairs, two sections each, numerator [1,0,-1]
int design_bp(uint32_t m, double f0, double fs, double q, double gain_in, double *a, double *b,
              double *gain)
{
    if (m == 0 || m % 2 != 0 || m > SDSP_HIP_MAX_SECTIONS)
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "M must be even! (and <= SDSP_HIP_MAX_SECTIONS)");
    if (!a || !b || !gain)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null output pointer");
    double g = gain_in;
    const double e0 = 2 * M_PI * f0 / fs;
    const double de = 2 * std::tan(e0 / (2 * q)) / std::sin(e0);
    for (uint32_t k = 0; k < m / 2; k++) {
        const double d = 2 * std::sin((2 * k + 1) * M_PI / (2.0 * m));
        const double aa = (1 + de * de / 4.0) * 2 / d / de;
        const double dk = std::sqrt(de * d / (aa + std::sqrt(aa * aa - 1)));
        const double bb = d * de / dk / 2.0;
        const double w = bb + std::sqrt(bb * bb - 1);
        const double th = std::tan(e0 / 2.0);
        const double e[2] = { 2.0 * std::atan(th / w), 2.0 * std::atan(w * th) };
        const double sc = std::sqrt(1 + (w - 1 / w) / dk * (w - 1 / w) / dk);
        double alpha[2];
        for (int h = 0; h < 2; h++) {
            const double t = dk * std::sin(e[h]) / 2.0;
            const double beta = (1 - t) / (1 + t) / 2.0;
            const double gamma = (0.5 + beta) * std::cos(e[h]);
            alpha[h] = (0.5 - beta) * sc / 2.0;
            const uint32_t s = 2 * k + h;
            b[3 * s + 0] = 1.0;
            b[3 * s + 1] = 0.0;
            b[3 * s + 2] = -1.0;
            a[3 * s + 0] = 1.0;
            a[3 * s + 1] = -2 * gamma;
            a[3 * s + 2] = 2 * beta;
        }
        g *= 4 * alpha[0] * alpha[1];
    }
    *gain = g;
    return SDSP_HIP_OK;
}

// Band-stop design -- the reference's README lists it as TODO (README.md:15); no reference code exists,
// so this is a new design in the same parameterisation as set_bp_coeff (centre f0, quality q ->
// -3 dB width f0/q, Butterworth prototype of order m): the band-stop frequency transformation
// s = tan(e0/2q) (z^2 - 1) / (z^2 - 2 cos(e0) z + 1) applied to each prototype pole p gives the
// quadratic (p - B) z^2 - 2 p c z + (p + B) = 0; its two roots (and their conjugates, from p*) are
// the poles of two sections.  Every section has the zero pair e^{+-j e0} (numerator [1, -2c, 1]);
// the gain normalises each section to 1 at DC.  Checked against scipy.signal.butter(btype='bandstop').
int design_bs(uint32_t m, double f0, double fs, double q, double gain_in, double *a, double *b,
              double *gain)
{
    if (m == 0 || m % 2 != 0 || m > SDSP_HIP_MAX_SECTIONS)
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "M must be even! (and <= SDSP_HIP_MAX_SECTIONS)");
    if (!a || !b || !gain)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null output pointer");
    using cplx = std::complex<double>;
    double g = gain_in;
    const double e0 = 2 * M_PI * f0 / fs;
    const double c = std::cos(e0);
    const double bw = std::tan(e0 / (2 * q));
    for (uint32_t k = 0; k < m / 2; k++) {
        const double th = (2 * k + 1) * M_PI / (2.0 * m);
        const cplx p(-std::sin(th), std::cos(th));
        const cplx qa = p - bw, qb = -2.0 * p * c, qc = p + bw;
        const cplx disc = std::sqrt(qb * qb - 4.0 * qa * qc);
        const cplx z[2] = { (-qb + disc) / (2.0 * qa), (-qb - disc) / (2.0 * qa) };
        for (int h = 0; h < 2; h++) {
            const uint32_t s = 2 * k + h;
            a[3 * s + 0] = 1.0;
            a[3 * s + 1] = -2 * z[h].real();
            a[3 * s + 2] = std::norm(z[h]);
            b[3 * s + 0] = 1.0;
            b[3 * s + 1] = -2 * c;
            b[3 * s + 2] = 1.0;
            g *= (1 + a[3 * s + 1] + a[3 * s + 2]) / (2 - 2 * c);
        }
    }
    *gain = g;
    return SDSP_HIP_OK;
}

// FIR design -- the reference's README.md:16 TODO; no reference code.  Windowed-sinc (Hamming) design,
// the same construction as scipy.signal.firwin (which the tests pin it to): ideal band responses
// sum(right sinc(right m) - left sinc(left m)) over the pass bands, times the symmetric Hamming window,
// scaled to unit gain at DC / Nyquist / the band centre.  lp, hp take the cutoff f0; bp, bs take
// centre f0 and q with edges f0 -+ f0/(2q) (width f0/q as in set_bp_coeff, without frequency warping).
int design_fir(uint32_t taps, int filter_type, double f0, double fs, double q, double gain_in, double *h)
{
    if (taps == 0 || taps > SDSP_HIP_FIR_MAX_TAPS)
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "taps must be in [1, SDSP_HIP_FIR_MAX_TAPS]");
    if (!h)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null output pointer");
    const double nyq = fs / 2;
    double lo = 0, hi = 0; // the band that defines the filter, as fractions of Nyquist
    bool pass_zero = false;
    switch (filter_type) {
    case SDSP_HIP_FILTER_LOW_PASS: lo = 0; hi = f0 / nyq; pass_zero = true; break;
    case SDSP_HIP_FILTER_HIGH_PASS: lo = f0 / nyq; hi = 1; pass_zero = false; break;
    case SDSP_HIP_FILTER_BAND_PASS: lo = (f0 - f0 / (2 * q)) / nyq; hi = (f0 + f0 / (2 * q)) / nyq; pass_zero = false; break;
    case SDSP_HIP_FILTER_BAND_STOP: lo = (f0 - f0 / (2 * q)) / nyq; hi = (f0 + f0 / (2 * q)) / nyq; pass_zero = true; break;
    default: return fail(SDSP_HIP_ERR_INVALID_ARG, "filter_type must be low_pass, high_pass, band_pass or band_stop");
    }
    const bool band = filter_type == SDSP_HIP_FILTER_BAND_PASS || filter_type == SDSP_HIP_FILTER_BAND_STOP;
    if (!(fs > 0) || !(band ? (lo > 0 && hi < 1 && lo < hi) : (f0 > 0 && f0 < nyq)))
        return fail(SDSP_HIP_ERR_INVALID_ARG, "cutoff frequencies must lie strictly between 0 and fs/2");
    const bool pass_nyquist = filter_type == SDSP_HIP_FILTER_HIGH_PASS || filter_type == SDSP_HIP_FILTER_BAND_STOP;
    if (pass_nyquist && taps % 2 == 0)
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "a filter that passes fs/2 needs an odd number of taps");
    // pass bands as (left, right) pairs
    double bands[2][2];
    int nb = 0;
    if (filter_type == SDSP_HIP_FILTER_LOW_PASS) { bands[nb][0] = 0; bands[nb++][1] = hi; }
    else if (filter_type == SDSP_HIP_FILTER_HIGH_PASS) { bands[nb][0] = lo; bands[nb++][1] = 1; }
    else if (filter_type == SDSP_HIP_FILTER_BAND_PASS) { bands[nb][0] = lo; bands[nb++][1] = hi; }
    else { bands[nb][0] = 0; bands[nb++][1] = lo; bands[nb][0] = hi; bands[nb++][1] = 1; }
    (void)pass_zero;
    auto sinc = [](double x) { return x == 0 ? 1.0 : std::sin(M_PI * x) / (M_PI * x); };
    const double alpha = 0.5 * (taps - 1);
    for (uint32_t i = 0; i < taps; i++) {
        const double m = i - alpha;
        double v = 0;
        for (int bnd = 0; bnd < nb; bnd++)
            v += bands[bnd][1] * sinc(bands[bnd][1] * m) - bands[bnd][0] * sinc(bands[bnd][0] * m);
        const double w = taps == 1 ? 1.0 : 0.54 - 0.46 * std::cos(2 * M_PI * i / (taps - 1));
        h[i] = v * w;
    }
    const double left = bands[0][0], right = bands[0][1];
    const double scale_frequency = left == 0 ? 0.0 : (right == 1 ? 1.0 : 0.5 * (left + right));
    double s = 0;
    for (uint32_t i = 0; i < taps; i++)
        s += h[i] * std::cos(M_PI * (i - alpha) * scale_frequency);
    for (uint32_t i = 0; i < taps; i++)
        h[i] = h[i] / s * gain_in;
    return SDSP_HIP_OK;
}

// preload_filter: casc_2o_iir.h:197-214.  DC propagates section to section only for low_pass
// (and band_stop, which also passes DC; not in the reference).
int preload(uint32_t m, int filter_type, const double *a, const double *b, double gain,
            double value, double *mem)
{
    if (m == 0 || m % 2 != 0 || m > SDSP_HIP_MAX_SECTIONS)
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "M must be even! (and <= SDSP_HIP_MAX_SECTIONS)");
    if (!mem || ((filter_type == SDSP_HIP_FILTER_LOW_PASS || filter_type == SDSP_HIP_FILTER_BAND_STOP) && (!a || !b)))
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null pointer");
    double v = value * gain;
    std::memset(mem, 0, sizeof(double) * 3 * (m + 1));
    for (int i = 0; i < 3; i++)
        mem[i] = v;
    if (filter_type == SDSP_HIP_FILTER_LOW_PASS || filter_type == SDSP_HIP_FILTER_BAND_STOP) {
        for (uint32_t j = 1; j < m + 1; j++) {
            v /= 1 + a[3 * (j - 1) + 1] + a[3 * (j - 1) + 2];
            v *= b[3 * (j - 1) + 0] + b[3 * (j - 1) + 1] + b[3 * (j - 1) + 2];
            for (int i = 0; i < 3; i++)
                mem[3 * j + i] = v;
        }
    }
    return SDSP_HIP_OK;
}
} // namespace sdsp_hip

using namespace sdsp_hip;

extern "C" {

const char *sdsp_hip_last_error_string(void) { return g_last_error.c_str(); }
// SDSP_HIP_SOURCE_HASH: sha256 over csrc/ and include/ at build time (simpledsp_amd/build.py); the Python loader compares
// it with the sources beside the library and refuses a stale .so
#ifndef SDSP_HIP_SOURCE_HASH
#define SDSP_HIP_SOURCE_HASH "unhashed"
#endif
const char *sdsp_hip_version(void) { return "sdsp-hip 0.3 (gfx950) src:" SDSP_HIP_SOURCE_HASH; }

// fft.h:12-19
unsigned sdsp_hip_log2(unsigned num)
{
    unsigned r = 0;
    while ((num >>= 1) > 0u)
        r++;
    return r;
}
// fft.h:21-28
unsigned sdsp_hip_log4(unsigned num)
{
    unsigned r = 0;
    while ((num >>= 2) > 0u)
        r++;
    return r;
}
// fft.h:31-37
int sdsp_hip_is_power_of_2(unsigned num) { return num != 0 && (num & (num - 1)) == 0; }
// fft.h:40-43
int sdsp_hip_is_power_of_4(unsigned num)
{
    return sdsp_hip_is_power_of_2(num) && (sdsp_hip_log2(num) % 2 == 0);
}
// fft.h:217-236, as a digit loop
unsigned sdsp_hip_digit_reverse(unsigned n, unsigned base, unsigned x)
{
    const unsigned bits = sdsp_hip_log2(base);
    const unsigned digits = bits ? sdsp_hip_log2(n) / bits : 0;
    unsigned r = 0;
    for (unsigned d = 0; d < digits; d++) {
        r = (r << bits) | (x & (base - 1));
        x >>= bits;
    }
    return r;
}

int sdsp_hip_calc_twiddles(unsigned n, int direction, double *out)
{
    if (!sdsp_hip_is_power_of_2(n))
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "FFT size must be a power of 2!");
    if (!out || (direction != SDSP_HIP_FORWARD && direction != SDSP_HIP_REVERSE))
        return fail(SDSP_HIP_ERR_INVALID_ARG, "bad argument");
    std::vector<double> w;
    make_twiddles(n, direction, w);
    std::memcpy(out, w.data(), sizeof(double) * w.size());
    return SDSP_HIP_OK;
}

int sdsp_hip_iir_design_lp(uint32_t m, double f0, double fs, double gain_in, double *a, double *b, double *gain)
{
    return design_lp(m, f0, fs, gain_in, a, b, gain);
}
int sdsp_hip_iir_design_hp(uint32_t m, double f0, double fs, double gain_in, double *a, double *b, double *gain)
{
    return design_hp(m, f0, fs, gain_in, a, b, gain);
}
int sdsp_hip_iir_design_bp(uint32_t m, double f0, double fs, double q, double gain_in, double *a, double *b, double *gain)
{
    return design_bp(m, f0, fs, q, gain_in, a, b, gain);
}
int sdsp_hip_iir_design_bs(uint32_t m, double f0, double fs, double q, double gain_in, double *a, double *b, double *gain)
{
    return design_bs(m, f0, fs, q, gain_in, a, b, gain);
}
int sdsp_hip_fir_design(uint32_t taps, int filter_type, double f0, double fs, double q, double gain_in, double *h)
{
    return design_fir(taps, filter_type, f0, fs, q, gain_in, h);
}
int sdsp_hip_iir_preload(uint32_t m, int filter_type, const double *a, const double *b, double gain, double value, double *mem)
{
    return preload(m, filter_type, a, b, gain, value, mem);
}
}

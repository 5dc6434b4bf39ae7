// ref_shim.cpp -- C-callable wrapper around the REAL simpledsp headers.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile with g++ (the reference's fft.h needs
// GCC's constexpr std::sin/std::cos, README.md:20) against the reference sources where they lie
// (-I/root/reference/include); the output goes to oracle/_ref/ which is git-ignored.  No reference
// source is copied into this repository.  It exists to (1) pin oracle/sdsp_oracle.c bit-for-bit,
// (2) generate tests/golden/ fixtures (oracle/gen_golden.py) and (3) serve as the "reference"
// CPU baseline that bench.py times beside the GPU path.
#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

// the generic filter keeps its designed coefficients private; the shim reads them out so the
// fixtures can carry them.  Standard headers are included above, before the keyword is remapped.
#define private public
#include "sdsp/casc_2o_iir.h"
#undef private
#include "sdsp/fft.h"

namespace
{
template <size_t N, class T>
void run_fft(int radix, double *data, size_t batch)
{
    for (size_t b = 0; b < batch; b++) {
        auto &arr = *reinterpret_cast<sdsp::complex_array<N> *>(data + 2 * N * b);
        if (radix == 2)
            sdsp::fft_radix2<T>(arr);
        else if constexpr (sdsp::isPowerOf4(N))
            sdsp::fft_radix4<T>(arr);
    }
}

template <size_t N>
int fft_n(int radix, int reverse, double *data, size_t batch)
{
    if (radix == 4 && !sdsp::isPowerOf4(N))
        return -1;
    if (reverse)
        run_fft<N, sdsp::reverse_fft>(radix, data, batch);
    else
        run_fft<N, sdsp::forward_fft>(radix, data, batch);
    return 0;
}

template <size_t N>
int wcoeffs_n(int reverse, double *out)
{
    if (reverse) {
        static const auto w = sdsp::calc_wCoeffs<N, sdsp::reverse_fft>();
        std::memcpy(out, w.data(), sizeof(w));
    } else {
        static const auto w = sdsp::calc_wCoeffs<N, sdsp::forward_fft>();
        std::memcpy(out, w.data(), sizeof(w));
    }
    return 0;
}

template <size_t N>
int swap_n(unsigned base, unsigned *out)
{
    if (base == 2) {
        auto l = sdsp::calc_swap_lookup<N, 2>();
        std::copy(l.begin(), l.end(), out);
    } else if constexpr (sdsp::isPowerOf4(N)) {
        auto l = sdsp::calc_swap_lookup<N, 4>();
        std::copy(l.begin(), l.end(), out);
    } else {
        return -1;
    }
    return 0;
}

// type-erased filter: kind 0 = casc_2o_iir, 1/2/3 = casc_2o_iir_lp/hp/bp
struct filt {
    virtual ~filt() = default;
    virtual int set_lp(double, double, double) { return -1; }
    virtual int set_hp(double, double, double) { return -1; }
    virtual int set_bp(double, double, double, double) { return -1; }
    virtual int preload(double) { return -1; }
    virtual void process(double *d, size_t n) = 0;
    virtual filt *clone() const = 0;
    virtual int coeffs(double *, double *, double *) const { return -1; }
    virtual int state(double *, int *) const { return -1; }
};

template <size_t M>
struct filt_generic final : filt {
    sdsp::casc_2o_iir<M> f;
    int set_lp(double f0, double fs, double g) override { f.set_lp_coeff(f0, fs, g); return 0; }
    int set_hp(double f0, double fs, double g) override { f.set_hp_coeff(f0, fs, g); return 0; }
    int set_bp(double f0, double fs, double q, double g) override { f.set_bp_coeff(f0, fs, q, g); return 0; }
    int preload(double v) override { f.preload_filter(v); return 0; }
    void process(double *d, size_t n) override { f.process(d, d + n); }
    filt *clone() const override { return new filt_generic(*this); }
    int coeffs(double *a, double *b, double *gain) const override
    {
        for (size_t j = 0; j < M; j++)
            for (size_t i = 0; i < 3; i++) {
                a[j * 3 + i] = f.m_a_coeff.at(j).at(i);
                b[j * 3 + i] = f.m_b_coeff.at(j).at(i);
            }
        *gain = f.m_gain;
        return 0;
    }
    int state(double *mem, int *pos) const override
    {
        for (size_t j = 0; j < M + 1; j++)
            for (size_t i = 0; i < 3; i++)
                mem[j * 3 + i] = f.m_mem.at(j).at(i);
        *pos = f.m_pos;
        return 0;
    }
};

template <size_t M>
struct filt_lp final : filt {
    sdsp::casc_2o_iir_lp<M> f;
    int set_lp(double f0, double fs, double g) override { f.set_lp_coeff(f0, fs, g); return 0; }
    void process(double *d, size_t n) override { f.process(d, d + n); }
    filt *clone() const override { return new filt_lp(*this); }
};
template <size_t M>
struct filt_hp final : filt {
    sdsp::casc_2o_iir_hp<M> f;
    int set_hp(double f0, double fs, double g) override { f.set_hp_coeff(f0, fs, g); return 0; }
    void process(double *d, size_t n) override { f.process(d, d + n); }
    filt *clone() const override { return new filt_hp(*this); }
};
template <size_t M>
struct filt_bp final : filt {
    sdsp::casc_2o_iir_bp<M> f;
    int set_bp(double f0, double fs, double q, double g) override { f.set_bp_coeff(f0, fs, q, g); return 0; }
    void process(double *d, size_t n) override { f.process(d, d + n); }
    filt *clone() const override { return new filt_bp(*this); }
};

template <size_t M>
filt *make(int kind)
{
    switch (kind) {
    case 0: return new filt_generic<M>();
    case 1: return new filt_lp<M>();
    case 2: return new filt_hp<M>();
    case 3: return new filt_bp<M>();
    default: return nullptr;
    }
}
} // namespace

#define FOR_SIZES(X) X(4) X(8) X(16) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096)

extern "C" {

int sdsp_ref_fft(unsigned n, int radix, int reverse, double *data, size_t batch)
{
    if (radix != 2 && radix != 4)
        return -1;
    switch (n) {
#define X(N) case N: return fft_n<N>(radix, reverse, data, batch);
        FOR_SIZES(X)
#undef X
    default: return -2;
    }
}

int sdsp_ref_wcoeffs(unsigned n, int reverse, double *out)
{
    switch (n) {
#define X(N) case N: return wcoeffs_n<N>(reverse, out);
        FOR_SIZES(X)
#undef X
    default: return -2;
    }
}

int sdsp_ref_swap_lookup(unsigned n, unsigned base, unsigned *out)
{
    switch (n) {
#define X(N) case N: return swap_n<N>(base, out);
        FOR_SIZES(X)
#undef X
    default: return -2;
    }
}

unsigned sdsp_ref_log2(unsigned v) { return sdsp::log2(v); }
unsigned sdsp_ref_log4(unsigned v) { return sdsp::log4(v); }
int sdsp_ref_is_power_of_2(unsigned v) { return sdsp::isPowerOf2(v); }
int sdsp_ref_is_power_of_4(unsigned v) { return sdsp::isPowerOf4(v); }

void *sdsp_ref_iir_create(unsigned m, int kind)
{
    switch (m) {
    case 2: return make<2>(kind);
    case 4: return make<4>(kind);
    case 6: return make<6>(kind);
    case 8: return make<8>(kind);
    default: return nullptr;
    }
}
void sdsp_ref_iir_destroy(void *h) { delete static_cast<filt *>(h); }
void *sdsp_ref_iir_clone(const void *h) { return static_cast<const filt *>(h)->clone(); }
int sdsp_ref_iir_set_lp_coeff(void *h, double f0, double fs, double g) { return static_cast<filt *>(h)->set_lp(f0, fs, g); }
int sdsp_ref_iir_set_hp_coeff(void *h, double f0, double fs, double g) { return static_cast<filt *>(h)->set_hp(f0, fs, g); }
int sdsp_ref_iir_set_bp_coeff(void *h, double f0, double fs, double q, double g) { return static_cast<filt *>(h)->set_bp(f0, fs, q, g); }
int sdsp_ref_iir_preload_filter(void *h, double v) { return static_cast<filt *>(h)->preload(v); }
void sdsp_ref_iir_process(void *h, double *d, size_t n) { static_cast<filt *>(h)->process(d, n); }
int sdsp_ref_iir_coeffs(const void *h, double *a, double *b, double *gain) { return static_cast<const filt *>(h)->coeffs(a, b, gain); }
int sdsp_ref_iir_state(const void *h, double *mem, int *pos) { return static_cast<const filt *>(h)->state(mem, pos); }
}

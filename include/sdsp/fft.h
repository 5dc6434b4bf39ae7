// sdsp/fft.h -- MI355X-backed drop-in for simpledsp's include/sdsp/fft.h.
//
// A program that includes "sdsp/fft.h" and calls sdsp::fft_radix2 / sdsp::fft_radix4 on a
// sdsp::complex_array<N> keeps compiling and gets the same results (within the reference's own
// test tolerance 4*N*eps, testFFT.cpp:37): the call is carried out by the f64 HIP kernels behind
// the C ABI (sdsp_hip.h) -- H2D, transform, D2H.  Moving 16 bytes per point over PCIe for one
// small transform is of course slower than a CPU; the drop-in form exists for source
// compatibility and for the parity tests.  The form that is worth a GPU is the batched one below
// (sdsp::fft_batch / sdsp::fft_plan): data resident in HBM, thousands of transforms per launch.
//
// Differences from the reference, all at the edges:
//   * the twiddle / digit-reversal tables are built at run time (the reference needs GCC's
//     constexpr std::sin/std::cos, its README.md:20; this header also compiles with clang/hipcc),
//     so calc_trigs / calc_wCoeffs / calc_swap_lookup are ordinary functions, not constexpr;
//   * errors the reference cannot have (no device, HIP failure) throw sdsp::hip_error;
//   * there is no CPU implementation in this header.
#pragma once

#include <array>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "detail/hip_runtime.h"

namespace sdsp
{
using uint = unsigned int; // the reference leans on glibc's ::uint (its fft.h:12)

// ---- size helpers (reference fft.h:12-43) ------------------------------------------------------
constexpr uint log2(uint num)
{
    uint bits = 0;
    for (num >>= 1; num != 0; num >>= 1)
        ++bits;
    return bits;
}

constexpr uint log4(uint num)
{
    uint digits = 0;
    for (num >>= 2; num != 0; num >>= 2)
        ++digits;
    return digits;
}

constexpr bool isPowerOf2(uint num) { return num != 0 && (num & (num - 1)) == 0; }

constexpr bool isPowerOf4(uint num) { return isPowerOf2(num) && (log2(num) & 1u) == 0; }

// ---- table types (reference fft.h:45-52) -------------------------------------------------------
template <size_t N> using trig_array = std::array<std::array<double, N>, log2(N)>;
template <size_t N> using coeff_array = std::array<std::array<std::complex<double>, N>, log2(N)>;
template <size_t N> using complex_array = std::array<std::complex<double>, N>;

// ---- direction policies (reference fft.h:121-146) ----------------------------------------------
class forward_fft {
public:
    static constexpr int direction = SDSP_HIP_FORWARD;
    constexpr static double Sign() { return 1.0; }
    template <size_t N> constexpr static void ScaleValues(complex_array<N> &) {}
};

class reverse_fft {
public:
    static constexpr int direction = SDSP_HIP_REVERSE;
    constexpr static double Sign() { return -1.0; }
    template <size_t N> static void ScaleValues(complex_array<N> &data)
    {
        for (auto &v : data)
            v *= (1.0 / N);
    }
};

// ---- quarter-wave calculators (reference fft.h:67-119): names kept for source compatibility -----
class sine_calculator {
public:
    constexpr static bool is_cosine = false;
    constexpr static double Value0() { return 0.0; }
    constexpr static double Value90() { return 1.0; }
    static double Value(double rad) { return std::sin(rad); }
    constexpr static double Sym0() { return -1.0; }
    constexpr static double Sym90() { return 1.0; }
};

class cosine_calculator {
public:
    constexpr static bool is_cosine = true;
    constexpr static double Value0() { return 1.0; }
    constexpr static double Value90() { return 0.0; }
    static double Value(double rad) { return std::cos(rad); }
    constexpr static double Sym0() { return 1.0; }
    constexpr static double Sym90() { return -1.0; }
};

// trig_array<N> straight from the calculator, no symmetry tricks (reference fft.h:54-65): row i, column j =
// T::Value(2 pi j / 2^(i+1)).  The reference keeps it beside calc_trigs for comparison; so does this header.
template <size_t N, class T> trig_array<N> calc_trigs_naive()
{
    trig_array<N> trigs{};
    for (size_t i = 0; i < trigs.size(); ++i) {
        const double period = static_cast<double>(1u << (i + 1u));
        for (size_t j = 0; j < N; ++j)
            trigs[i][j] = T::Value(2 * M_PI * static_cast<double>(j) / period);
    }
    return trigs;
}

// ---- run-time tables ---------------------------------------------------------------------------
// One row exp(-/+ 2 pi i j / N), j < N -- the only row the GPU kernels keep resident; produced by
// the same host routine that fills the plans (sdsp_hip_calc_twiddles).
template <size_t N, class T> std::array<std::complex<double>, N> calc_twiddle_row()
{
    static_assert(isPowerOf2(N), "FFT size must be a power of 2!");
    std::array<std::complex<double>, N> row{};
    detail::check(sdsp_hip_calc_twiddles(static_cast<unsigned>(N), T::direction, reinterpret_cast<double *>(row.data())));
    return row;
}

// coeff_array<N> of the reference (fft.h:197-214): row i, column j = exp(-/+ 2 pi i j / 2^(i+1)),
// i.e. the twiddle row sampled with stride N / 2^(i+1).
template <size_t N, class T> coeff_array<N> calc_wCoeffs()
{
    const auto row = calc_twiddle_row<N, T>();
    coeff_array<N> w{};
    for (size_t i = 0; i < w.size(); ++i) {
        const size_t stride = N >> (i + 1);
        for (size_t j = 0; j < N; ++j)
            w[i][j] = row[(j * stride) & (N - 1)];
    }
    return w;
}

// trig_array<N> of the reference (fft.h:148-194): cosines / sines of the same angles
template <size_t N, class T> trig_array<N> calc_trigs()
{
    const auto w = calc_wCoeffs<N, forward_fft>();
    trig_array<N> t{};
    for (size_t i = 0; i < t.size(); ++i)
        for (size_t j = 0; j < N; ++j)
            t[i][j] = T::is_cosine ? w[i][j].real() : -w[i][j].imag();
    return t;
}

// digit reversal (reference fft.h:217-236), written as a digit loop
template <size_t N, uint base> constexpr uint digit_reverse(uint n)
{
    constexpr uint bits = log2(base);
    constexpr uint digits = log2(static_cast<uint>(N)) / bits;
    uint out = 0;
    for (uint d = 0; d < digits; ++d) {
        out = (out << bits) | (n & (base - 1u));
        n >>= bits;
    }
    return out;
}

// de-duplicated swap list (reference fft.h:238-256).  The GPU kernels fold the permutation into
// their addressing and never read such a table; provided for source compatibility.
template <size_t N, uint base> std::array<uint, N> calc_swap_lookup()
{
    std::array<uint, N> lut{};
    for (size_t i = 0; i < N; ++i)
        lut[i] = digit_reverse<N, base>(static_cast<uint>(i));
    for (size_t i = 1; i + 1 < N; ++i)
        if (lut[i] != i)
            lut[lut[i]] = lut[i];
    return lut;
}

// ---- batched plans: the form worth a GPU --------------------------------------------------------
// real_t = float (BASELINE configs) or double.  Data: batch x n interleaved complex, in place.
template <typename real_t> class fft_plan {
public:
    fft_plan(std::uint32_t n, int radix, int direction = SDSP_HIP_FORWARD, std::uint64_t max_batch = 1, int device = 0)
        : m_n(n), m_device(device),
          m_h(std::make_unique<detail::fft_plan_handle>(n, radix, direction, detail::precision_of<real_t>::value,
                                                        max_batch, device))
    {
    }
    std::uint32_t size() const noexcept { return m_n; }
    int device() const noexcept { return m_device; }
    // device pointer, asynchronous on `stream` (hipStream_t as void*)
    void exec(std::complex<real_t> *device_data, std::uint64_t batch, void *stream = nullptr)
    {
        detail::check(sdsp_hip_fft_exec(m_h->get(), device_data, batch, stream));
    }
    // host pointer: H2D, transform, D2H
    void exec_host(std::complex<real_t> *host_data, std::uint64_t batch)
    {
        std::lock_guard<std::mutex> lock(m_h->mutex());
        detail::check(sdsp_hip_fft_exec_host(m_h->get(), host_data, batch));
    }
    // fast convolution: data <- IFFT(FFT(data) .* h) per transform, in place; needs a
    // FORWARD plan; device pointers; h = n complex values (frequency response, natural order).
    // One fused kernel for float n = 16 .. 16384 (radix 2: .. 32768) and double n = 16 .. 8192; three launches beyond.
    void convolve(std::complex<real_t> *device_data, const std::complex<real_t> *device_h, std::uint64_t batch,
                  void *stream = nullptr)
    {
        detail::check(sdsp_hip_fft_convolve(m_h->get(), device_data, device_h, batch, stream));
    }
    sdsp_hip_fft_plan *native_handle() const noexcept { return m_h->get(); }

private:
    std::uint32_t m_n;
    int m_device;
    std::unique_ptr<detail::fft_plan_handle> m_h;
};

// real-input packing: n_real real samples <-> packed half spectrum (n_real/2 complex,
// element 0 = (X[0], X[n_real/2])), in place, half the bytes of the complex transform.  real_t = float
// (n_real = 32 .. 32768; radix 2: .. 65536) or double (32 .. 16384; radix 2: .. 32768 -- the reference's precision).
template <typename real_t> class rfft_plan_t {
public:
    rfft_plan_t(std::uint32_t n_real, int radix = 2, int direction = SDSP_HIP_FORWARD, std::uint64_t max_batch = 1, int device = 0)
        : m_n_real(n_real)
    {
        detail::check(sdsp_hip_rfft_plan_create_p(&m_plan, n_real, radix, direction, detail::precision_of<real_t>::value,
                                                  max_batch, device));
    }
    ~rfft_plan_t() { sdsp_hip_fft_plan_destroy(m_plan); }
    rfft_plan_t(const rfft_plan_t &) = delete;
    rfft_plan_t &operator=(const rfft_plan_t &) = delete;
    std::uint32_t size() const noexcept { return m_n_real; }
    void exec(real_t *device_data, std::uint64_t batch, void *stream = nullptr)
    {
        detail::check(sdsp_hip_fft_exec(m_plan, device_data, batch, stream));
    }
    void exec_host(real_t *host_data, std::uint64_t batch) { detail::check(sdsp_hip_fft_exec_host(m_plan, host_data, batch)); }

private:
    std::uint32_t m_n_real;
    sdsp_hip_fft_plan *m_plan{ nullptr };
};
using rfft_plan = rfft_plan_t<float>;

// batch of `batch` transforms of length n in host memory, in place
template <class T = forward_fft, typename real_t> void fft_batch(int radix, std::complex<real_t> *data, std::uint32_t n, std::uint64_t batch)
{
    fft_plan<real_t> plan(n, radix, T::direction, batch, 0);
    plan.exec_host(data, batch);
}

namespace detail
{
template <size_t N, int RADIX, class T> void run_single(std::complex<double> *data)
{
    // one resident plan per (N, radix, direction), built on first use
    static detail::fft_plan_handle plan(static_cast<std::uint32_t>(N), RADIX, T::direction, SDSP_HIP_F64, 1, 0);
    std::lock_guard<std::mutex> lock(plan.mutex());
    check(sdsp_hip_fft_exec_host(plan.get(), data, 1));
}
} // namespace detail

// ---- the reference's call surface (fft.h:258-259, :301-302) --------------------------------------
template <class T = forward_fft, size_t N> void fft_radix2(complex_array<N> &data)
{
    static_assert(isPowerOf2(N), "FFT size must be a power of 2!");
    detail::run_single<N, 2, T>(data.data());
}

template <class T = forward_fft, size_t N> void fft_radix4(complex_array<N> &data)
{
    static_assert(isPowerOf4(N), "FFT radix 4 size must be a power of 4!");
    detail::run_single<N, 4, T>(data.data());
}
} // namespace sdsp

// sdsp/fir.h -- FIR filter for the MI355X engine, in the style of sdsp::casc_2o_iir.
//
// The reference has NO FIR filter: it is a TODO in its README (README.md:16).  This
// header therefore mirrors the conventions of the reference's IIR class (casc_2o_iir.h:23-37,
// :82-214) -- value type, set_*_coeff(f0, fs[, q], gain_in), process(begin, end) in place and
// stateful across calls, copy_coeff_from, preload_filter -- instead of existing code.  The design is
// a Hamming-windowed sinc (== scipy.signal.firwin, which the tests pin it to).  Every call goes
// through the C ABI (sdsp_hip.h); there is no CPU path.
#ifndef SDSP_MI355X_FIR_H
#define SDSP_MI355X_FIR_H

#include <array>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <type_traits>
#include <vector>

#include "detail/hip_runtime.h"
#include "filter_type.h"

namespace sdsp
{
// ---- one stream, double samples, like the reference's filter classes ---------------------------
template <size_t n_taps> class fir_filter {
    static_assert(n_taps >= 1, "a FIR filter needs at least one tap");
    std::array<double, n_taps> m_coeff{};
    std::array<double, (n_taps > 1 ? n_taps - 1 : 1)> m_mem{}; // previous inputs, newest first
    filter_type m_f_type{ filter_type::none };

    void design(filter_type t, double f0, double fs, double q, double gain_in)
    {
        detail::check(sdsp_hip_fir_design(static_cast<std::uint32_t>(n_taps), static_cast<int>(t), f0, fs, q, gain_in, m_coeff.data()));
        m_f_type = t;
    }

public:
    fir_filter() = default;

    void copy_coeff_from(const fir_filter<n_taps> &other_filter)
    {
        m_coeff = other_filter.m_coeff;
        m_f_type = other_filter.m_f_type;
    }

    template <typename iter_t> void process(iter_t begin, iter_t end)
    {
        using value_t = typename std::iterator_traits<iter_t>::value_type;
        static_assert(std::is_same<value_t, double>::value, "the drop-in classes filter double samples, like the reference");
        const auto n = static_cast<std::uint64_t>(std::distance(begin, end));
        if (n == 0)
            return;
        sdsp_hip_fir_plan *plan = nullptr;
        detail::check(sdsp_hip_fir_plan_create(&plan, static_cast<std::uint32_t>(n_taps), m_coeff.data(), SDSP_HIP_F64, 0));
        const int rc = sdsp_hip_fir_process_host(plan, &*begin, 1, n, n, m_mem.data());
        sdsp_hip_fir_plan_destroy(plan);
        detail::check(rc);
    }

    void set_coeff(const std::array<double, n_taps> &h, double gain_in = 1.0)
    {
        for (size_t i = 0; i < n_taps; ++i)
            m_coeff[i] = h[i] * gain_in;
        m_f_type = filter_type::none;
    }
    void set_lp_coeff(double f0, double fs, double gain_in = 1.0) { design(filter_type::low_pass, f0, fs, 0.0, gain_in); }
    void set_hp_coeff(double f0, double fs, double gain_in = 1.0) { design(filter_type::high_pass, f0, fs, 0.0, gain_in); }
    void set_bp_coeff(double f0, double fs, double q, double gain_in = 1.0) { design(filter_type::band_pass, f0, fs, q, gain_in); }
    void set_bs_coeff(double f0, double fs, double q, double gain_in = 1.0) { design(filter_type::band_stop, f0, fs, q, gain_in); }

    // history of a steady input equal to `value`
    void preload_filter(double value) { m_mem.fill(value); }

    filter_type type() const { return m_f_type; }
    const std::array<double, n_taps> &coeff() const { return m_coeff; }
};

// ---- the batched entry: a bank of channels on the device ----------------------------------------
template <size_t n_taps, typename real_t = float> class fir_bank {
public:
    explicit fir_bank(std::uint64_t channels, int device = 0) : m_channels(channels), m_device(device) {}
    ~fir_bank()
    {
        if (m_plan)
            sdsp_hip_fir_plan_destroy(m_plan);
        if (m_state)
            sdsp_hip_free(m_state, m_device);
    }
    fir_bank(const fir_bank &) = delete;
    fir_bank &operator=(const fir_bank &) = delete;

    void set_coeff(const std::array<double, n_taps> &h)
    {
        m_coeff = h;
        redesign(filter_type::none);
    }
    void set_lp_coeff(double f0, double fs, double gain_in = 1.0) { design(filter_type::low_pass, f0, fs, 0.0, gain_in); }
    void set_hp_coeff(double f0, double fs, double gain_in = 1.0) { design(filter_type::high_pass, f0, fs, 0.0, gain_in); }
    void set_bp_coeff(double f0, double fs, double q, double gain_in = 1.0) { design(filter_type::band_pass, f0, fs, q, gain_in); }
    void set_bs_coeff(double f0, double fs, double q, double gain_in = 1.0) { design(filter_type::band_stop, f0, fs, q, gain_in); }

    void preload_filter(double value) { fill_state(static_cast<real_t>(value)); }
    void reset()
    {
        if (m_state)
            fill_state(real_t(0));
    }

    // device pointer, channel-major, asynchronous on `stream`; continues every channel's stream
    void process(real_t *device_data, std::uint64_t samples, std::uint64_t stride, void *stream = nullptr)
    {
        ensure_plan();
        ensure_state();
        detail::check(sdsp_hip_fir_process(m_plan, device_data, m_channels, samples, stride, m_state, stream));
    }
    // host pointer convenience: channels x samples, contiguous
    void process_host(real_t *host_data, std::uint64_t samples)
    {
        ensure_plan();
        ensure_state();
        const size_t bytes = static_cast<size_t>(m_channels * samples) * sizeof(real_t);
        void *d = nullptr;
        detail::check(sdsp_hip_malloc(&d, bytes, m_device));
        int rc = sdsp_hip_memcpy_h2d(d, host_data, bytes, m_device);
        if (!rc)
            rc = sdsp_hip_fir_process(m_plan, d, m_channels, samples, samples, m_state, nullptr);
        if (!rc)
            rc = sdsp_hip_memcpy_d2h(host_data, d, bytes, m_device);
        sdsp_hip_free(d, m_device);
        detail::check(rc);
    }
    std::uint64_t channels() const noexcept { return m_channels; }
    const std::array<double, n_taps> &coeff() const { return m_coeff; }

private:
    static constexpr size_t hist = n_taps > 1 ? n_taps - 1 : 1;
    void design(filter_type t, double f0, double fs, double q, double gain_in)
    {
        detail::check(sdsp_hip_fir_design(static_cast<std::uint32_t>(n_taps), static_cast<int>(t), f0, fs, q, gain_in, m_coeff.data()));
        redesign(t);
    }
    void redesign(filter_type t)
    {
        m_f_type = t;
        if (m_plan) {
            sdsp_hip_fir_plan_destroy(m_plan);
            m_plan = nullptr;
        }
    }
    void ensure_plan()
    {
        if (!m_plan)
            detail::check(sdsp_hip_fir_plan_create(&m_plan, static_cast<std::uint32_t>(n_taps), m_coeff.data(),
                                                   detail::precision_of<real_t>::value, m_device));
    }
    void fill_state(real_t v)
    {
        const bool fresh = !m_state;
        if (fresh)
            detail::check(sdsp_hip_malloc(&m_state, hist * m_channels * sizeof(real_t), m_device));
        std::vector<real_t> host(hist * m_channels, v);
        detail::check(sdsp_hip_memcpy_h2d(m_state, host.data(), host.size() * sizeof(real_t), m_device));
    }
    void ensure_state()
    {
        if (!m_state)
            fill_state(real_t(0));
    }

    std::uint64_t m_channels;
    int m_device;
    std::array<double, n_taps> m_coeff{};
    filter_type m_f_type{ filter_type::none };
    sdsp_hip_fir_plan *m_plan{ nullptr };
    void *m_state{ nullptr };
};
} // namespace sdsp

#endif // SDSP_MI355X_FIR_H

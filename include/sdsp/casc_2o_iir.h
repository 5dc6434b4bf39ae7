// sdsp/casc_2o_iir.h -- MI355X-backed drop-in for simpledsp's include/sdsp/casc_2o_iir.h.
//
// sdsp::casc_2o_iir<m_t> and sdsp::casc_2o_iir_{lp,hp,bp}<m_t> keep the reference's members and
// meaning: value-type filter objects (copy = coefficients AND state, testIIR.cpp:48), coefficient
// design on the host in double (set_*_coeff), streaming process(begin, end) in place.  process()
// runs the f64 HIP kernel behind the C ABI (sdsp_hip.h) with one channel; that kernel keeps the
// reference's operation order, so results are the reference's doubles bit for bit and feeding a
// stream block by block gives exactly what one long call gives (testIIR.cpp:61-75).
//
// One stream through PCIe is not what a GPU is for: the batched entry is
// sdsp::casc_2o_iir_bank<m_t, real_t> below -- N channels, shared coefficients, per-channel state
// resident on the device.
//
// Reference warts handled on purpose: the specialised classes' copy_coeff_from (reference lines
// 274-278, 332-336, 390-394) names members that do not exist and only compiles because nothing
// calls it; here it copies the gain and the denominator as intended.  There is no CPU
// implementation in this header; failures of the GPU path throw sdsp::hip_error.
#pragma once

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
namespace detail
{
template <size_t m_t> using mem_array = std::array<std::array<double, 3>, m_t + 1>;
template <size_t m_t> using coeff3_array = std::array<std::array<double, 3>, m_t>;

// Device-side context of ONE filter object: the plan and a staging buffer that survive between process() calls, so
// that a caller streaming short blocks (testIIR.cpp:61-75 feeds 32 samples at a time) pays one upload, one launch
// and one download per block instead of a plan build and two hipMalloc/hipFree pairs.  It is a cache, not state:
// copying a filter copies coefficients and history (the reference's value semantics, testIIR.cpp:48) and starts
// the copy with an empty context; a new design drops the plan.
class stream_ctx {
public:
    stream_ctx() = default;
    stream_ctx(const stream_ctx &) noexcept {}
    stream_ctx &operator=(const stream_ctx &) noexcept
    {
        drop_plan(); // the design may differ after an assignment
        return *this;
    }
    ~stream_ctx()
    {
        drop_plan();
        if (m_dev)
            sdsp_hip_free(m_dev, 0);
    }
    void drop_plan() noexcept
    {
        if (m_plan)
            sdsp_hip_iir_plan_destroy(m_plan);
        m_plan = nullptr;
    }
    sdsp_hip_iir_plan *&plan() noexcept { return m_plan; }
    // device buffer of at least `doubles` doubles (grow-only)
    double *device(std::size_t doubles)
    {
        if (doubles > m_cap) {
            if (m_dev)
                sdsp_hip_free(m_dev, 0);
            m_dev = nullptr;
            m_cap = 0;
            std::size_t want = 256;
            while (want < doubles)
                want *= 2;
            void *d = nullptr;
            check(sdsp_hip_malloc(&d, want * sizeof(double), 0));
            m_dev = static_cast<double *>(d);
            m_cap = want;
        }
        return m_dev;
    }
    std::vector<double> &host() noexcept { return m_host; }

private:
    sdsp_hip_iir_plan *m_plan{ nullptr };
    double *m_dev{ nullptr };
    std::size_t m_cap{ 0 };
    std::vector<double> m_host; // [state | samples] staging image
};

// Run the samples of [begin, end) of ONE stream through the f64 bank kernel.  The reference keeps its history in a
// 3-deep ring indexed by m_pos (casc_2o_iir.h:11-15, :54-60, :73-75); the device layout is the same
// ring rotated so that slot `age` is the value age+1 samples ago (sdsp_hip.h).  Like the reference's process(),
// this takes any iterator pair with ++, * and != over doubles (std::deque, std::list ...): the samples are staged
// through one contiguous [state | samples] image, which is also what makes the call one upload and one download.
template <size_t m_t, typename iter_t>
void process_single(stream_ctx &ctx, int kind, double gain, const coeff3_array<m_t> &a, const coeff3_array<m_t> *b,
                    mem_array<m_t> &mem, int &pos, iter_t begin, iter_t end)
{
    using value_t = typename std::iterator_traits<iter_t>::value_type;
    static_assert(std::is_same<value_t, double>::value, "the drop-in classes filter double samples, like the reference");
    static_assert(m_t <= SDSP_HIP_MAX_SECTIONS, "at most SDSP_HIP_MAX_SECTIONS (16) sections are compiled into libsdsp_hip");
    constexpr std::size_t kState = 3 * (m_t + 1);
    std::vector<double> &img = ctx.host();
    img.resize(kState);
    for (size_t j = 0; j <= m_t; ++j)
        for (int age = 0; age < 3; ++age)
            img[3 * j + static_cast<size_t>(age)] = mem[j][static_cast<size_t>((pos + 2 - age + 3) % 3)]; // (pos-1-age) mod 3
    for (iter_t it = begin; it != end; ++it)
        img.push_back(*it);
    const std::uint64_t n = img.size() - kState;
    if (n == 0)
        return;

    if (!ctx.plan()) {
        std::array<double, 3 * m_t> af{}, bf{};
        for (size_t j = 0; j < m_t; ++j)
            for (size_t i = 0; i < 3; ++i) {
                af[3 * j + i] = a[j][i];
                bf[3 * j + i] = b ? (*b)[j][i] : 0.0;
            }
        check(sdsp_hip_iir_plan_create(&ctx.plan(), static_cast<std::uint32_t>(m_t), kind, af.data(), b ? bf.data() : nullptr,
                                       gain, SDSP_HIP_F64, 0));
    }
    double *dev = ctx.device(img.size());
    check(sdsp_hip_memcpy_h2d(dev, img.data(), img.size() * sizeof(double), 0));
    check(sdsp_hip_iir_process(ctx.plan(), dev + kState, 1, n, n, dev, nullptr));
    check(sdsp_hip_memcpy_d2h(img.data(), dev, img.size() * sizeof(double), 0)); // synchronises with the launch

    std::size_t k = kState;
    for (iter_t it = begin; it != end; ++it)
        *it = img[k++];
    pos = static_cast<int>((static_cast<std::uint64_t>(pos) + n) % 3);
    for (size_t j = 0; j <= m_t; ++j)
        for (int age = 0; age < 3; ++age)
            mem[j][static_cast<size_t>((pos + 2 - age + 3) % 3)] = img[3 * j + static_cast<size_t>(age)];
}
} // namespace detail

// ---- run-time configurable cascade (reference casc_2o_iir.h:8-215) -------------------------------
template <size_t m_t> class casc_2o_iir {
private:
    int m_pos{ 0 };
    double m_gain{ 1.0 };
    detail::mem_array<m_t> m_mem{};
    detail::coeff3_array<m_t> m_b_coeff{};
    detail::coeff3_array<m_t> m_a_coeff{};
    filter_type m_f_type{ filter_type::none };
    detail::stream_ctx m_ctx; // device-side cache (plan + staging buffer), not part of the filter's value

    void store_design(const std::array<double, 3 * m_t> &a, const std::array<double, 3 * m_t> &b, double gain, filter_type t)
    {
        for (size_t j = 0; j < m_t; ++j)
            for (size_t i = 0; i < 3; ++i) {
                m_a_coeff[j][i] = a[3 * j + i];
                m_b_coeff[j][i] = b[3 * j + i];
            }
        m_gain = gain;
        m_f_type = t;
        m_ctx.drop_plan();
    }

public:
    casc_2o_iir() { static_assert(m_t % 2 == 0, "M must be even!"); }

    void copy_coeff_from(const casc_2o_iir<m_t> &other_filter)
    {
        m_gain = other_filter.m_gain;
        m_b_coeff = other_filter.m_b_coeff;
        m_a_coeff = other_filter.m_a_coeff;
        m_f_type = other_filter.m_f_type;
        m_ctx.drop_plan();
    }

    template <typename iter_t> void process(iter_t begin, iter_t end)
    {
        detail::process_single<m_t>(m_ctx, SDSP_HIP_IIR_GENERIC, m_gain, m_a_coeff, &m_b_coeff, m_mem, m_pos, begin, end);
    }

    void set_bp_coeff(double f0, double fs, double q, double gain_in = 1.0)
    {
        std::array<double, 3 * m_t> a{}, b{};
        double g = 0;
        detail::check(sdsp_hip_iir_design_bp(m_t, f0, fs, q, gain_in, a.data(), b.data(), &g));
        store_design(a, b, g, filter_type::band_pass);
    }

    // band-stop: the reference's README TODO (README.md:15); same parameters as set_bp_coeff
    void set_bs_coeff(double f0, double fs, double q, double gain_in = 1.0)
    {
        std::array<double, 3 * m_t> a{}, b{};
        double g = 0;
        detail::check(sdsp_hip_iir_design_bs(m_t, f0, fs, q, gain_in, a.data(), b.data(), &g));
        store_design(a, b, g, filter_type::band_stop);
    }

    void set_hp_coeff(double f0, double fs, double gain_in = 1.0)
    {
        std::array<double, 3 * m_t> a{}, b{};
        double g = 0;
        detail::check(sdsp_hip_iir_design_hp(m_t, f0, fs, gain_in, a.data(), b.data(), &g));
        store_design(a, b, g, filter_type::high_pass);
    }

    void set_lp_coeff(double f0, double fs, double gain_in = 1.0)
    {
        std::array<double, 3 * m_t> a{}, b{};
        double g = 0;
        detail::check(sdsp_hip_iir_design_lp(m_t, f0, fs, gain_in, a.data(), b.data(), &g));
        store_design(a, b, g, filter_type::low_pass);
    }

    // preload the filter memory for a steady-state input equal to `value` (reference :197-214)
    void preload_filter(double value)
    {
        std::array<double, 3 * m_t> a{}, b{};
        for (size_t j = 0; j < m_t; ++j)
            for (size_t i = 0; i < 3; ++i) {
                a[3 * j + i] = m_a_coeff[j][i];
                b[3 * j + i] = m_b_coeff[j][i];
            }
        std::array<double, 3 * (m_t + 1)> mem{};
        detail::check(sdsp_hip_iir_preload(m_t, static_cast<int>(m_f_type), a.data(), b.data(), m_gain, value, mem.data()));
        for (size_t j = 0; j <= m_t; ++j)
            for (size_t i = 0; i < 3; ++i)
                m_mem[j][i] = mem[3 * j + i];
    }

    // read-only views (not in the reference; handy for tests and for seeding a bank)
    double gain() const { return m_gain; }
    filter_type type() const { return m_f_type; }
    const detail::coeff3_array<m_t> &a_coeff() const { return m_a_coeff; }
    const detail::coeff3_array<m_t> &b_coeff() const { return m_b_coeff; }
};

// ---- numerator-folded cascades (reference casc_2o_iir.h:217-468) --------------------------------
template <size_t m_t> class casc_2o_iir_base {
protected:
    int m_pos{ 0 };
    double m_gain{ 1.0 };
    detail::mem_array<m_t> m_mem{};
    detail::coeff3_array<m_t> m_a_coeff{};
    detail::stream_ctx m_ctx; // device-side cache (plan + staging buffer), not part of the filter's value

    template <typename iter_t> void process_kind(int kind, iter_t begin, iter_t end)
    {
        detail::process_single<m_t>(m_ctx, kind, m_gain, m_a_coeff, nullptr, m_mem, m_pos, begin, end);
    }
    void store_design(const std::array<double, 3 * m_t> &a, double gain)
    {
        for (size_t j = 0; j < m_t; ++j)
            for (size_t i = 0; i < 3; ++i)
                m_a_coeff[j][i] = a[3 * j + i];
        m_gain = gain;
        m_ctx.drop_plan();
    }
    void copy_design(const casc_2o_iir_base &o)
    {
        m_gain = o.m_gain;
        m_a_coeff = o.m_a_coeff;
        m_ctx.drop_plan();
    }
};

template <size_t m_t> class casc_2o_iir_lp : casc_2o_iir_base<m_t> {
public:
    casc_2o_iir_lp() { static_assert(m_t % 2 == 0, "M must be even!"); }
    void copy_coeff_from(const casc_2o_iir_lp<m_t> &other_filter) { this->copy_design(other_filter); }
    template <typename iter_t> void process(iter_t begin, iter_t end) { this->process_kind(SDSP_HIP_IIR_LP, begin, end); }
    void set_lp_coeff(double f0, double fs, double gain_in = 1.0)
    {
        std::array<double, 3 * m_t> a{}, b{};
        double g = 0;
        detail::check(sdsp_hip_iir_design_lp(m_t, f0, fs, gain_in, a.data(), b.data(), &g));
        this->store_design(a, g);
    }
};

template <size_t m_t> class casc_2o_iir_hp : casc_2o_iir_base<m_t> {
public:
    casc_2o_iir_hp() { static_assert(m_t % 2 == 0, "M must be even!"); }
    void copy_coeff_from(const casc_2o_iir_hp<m_t> &other_filter) { this->copy_design(other_filter); }
    template <typename iter_t> void process(iter_t begin, iter_t end) { this->process_kind(SDSP_HIP_IIR_HP, begin, end); }
    void set_hp_coeff(double f0, double fs, double gain_in = 1.0)
    {
        std::array<double, 3 * m_t> a{}, b{};
        double g = 0;
        detail::check(sdsp_hip_iir_design_hp(m_t, f0, fs, gain_in, a.data(), b.data(), &g));
        this->store_design(a, g);
    }
};

template <size_t m_t> class casc_2o_iir_bp : casc_2o_iir_base<m_t> {
public:
    casc_2o_iir_bp() { static_assert(m_t % 2 == 0, "M must be even!"); }
    void copy_coeff_from(const casc_2o_iir_bp<m_t> &other_filter) { this->copy_design(other_filter); }
    template <typename iter_t> void process(iter_t begin, iter_t end) { this->process_kind(SDSP_HIP_IIR_BP, begin, end); }
    void set_bp_coeff(double f0, double fs, double q, double gain_in = 1.0)
    {
        std::array<double, 3 * m_t> a{}, b{};
        double g = 0;
        detail::check(sdsp_hip_iir_design_bp(m_t, f0, fs, q, gain_in, a.data(), b.data(), &g));
        this->store_design(a, g);
    }
};

// band-stop in the style of the classes above (not in the reference, README.md:15 TODO).  Its numerator
// [1, -2cos(w0), 1] has a per-design middle term, so it keeps b and runs the generic recurrence.
template <size_t m_t> class casc_2o_iir_bs : casc_2o_iir_base<m_t> {
    detail::coeff3_array<m_t> m_b_coeff{};

public:
    casc_2o_iir_bs() { static_assert(m_t % 2 == 0, "M must be even!"); }
    void copy_coeff_from(const casc_2o_iir_bs<m_t> &other_filter)
    {
        this->copy_design(other_filter);
        m_b_coeff = other_filter.m_b_coeff;
    }
    template <typename iter_t> void process(iter_t begin, iter_t end)
    {
        detail::process_single<m_t>(this->m_ctx, SDSP_HIP_IIR_GENERIC, this->m_gain, this->m_a_coeff, &m_b_coeff, this->m_mem,
                                    this->m_pos, begin, end);
    }
    void set_bs_coeff(double f0, double fs, double q, double gain_in = 1.0)
    {
        std::array<double, 3 * m_t> a{}, b{};
        double g = 0;
        detail::check(sdsp_hip_iir_design_bs(m_t, f0, fs, q, gain_in, a.data(), b.data(), &g));
        this->store_design(a, g);
        for (size_t j = 0; j < m_t; ++j)
            for (size_t i = 0; i < 3; ++i)
                m_b_coeff[j][i] = b[3 * j + i];
    }
};

// ---- the batched entry: a bank of channels on the device ----------------------------------------
// `channels` independent streams share one design; per-channel state stays resident in HBM between
// process() calls.  Data is channel-major: channel c's samples are data[c*stride .. c*stride+samples).
template <size_t m_t, typename real_t = float> class casc_2o_iir_bank {
public:
    explicit casc_2o_iir_bank(std::uint64_t channels, int kind = SDSP_HIP_IIR_GENERIC, int device = 0)
        : m_channels(channels), m_kind(kind), m_device(device)
    {
        static_assert(m_t % 2 == 0, "M must be even!");
    }
    ~casc_2o_iir_bank()
    {
        if (m_plan)
            sdsp_hip_iir_plan_destroy(m_plan);
        if (m_state)
            sdsp_hip_free(m_state, m_device);
    }
    casc_2o_iir_bank(const casc_2o_iir_bank &) = delete;
    casc_2o_iir_bank &operator=(const casc_2o_iir_bank &) = delete;

    void set_lp_coeff(double f0, double fs, double gain_in = 1.0)
    {
        detail::check(sdsp_hip_iir_design_lp(m_t, f0, fs, gain_in, m_a.data(), m_b.data(), &m_gain));
        redesign(filter_type::low_pass);
    }
    void set_hp_coeff(double f0, double fs, double gain_in = 1.0)
    {
        detail::check(sdsp_hip_iir_design_hp(m_t, f0, fs, gain_in, m_a.data(), m_b.data(), &m_gain));
        redesign(filter_type::high_pass);
    }
    void set_bp_coeff(double f0, double fs, double q, double gain_in = 1.0)
    {
        detail::check(sdsp_hip_iir_design_bp(m_t, f0, fs, q, gain_in, m_a.data(), m_b.data(), &m_gain));
        redesign(filter_type::band_pass);
    }
    // band-stop needs the generic recurrence: construct the bank with kind SDSP_HIP_IIR_GENERIC (the default)
    void set_bs_coeff(double f0, double fs, double q, double gain_in = 1.0)
    {
        detail::check(sdsp_hip_iir_design_bs(m_t, f0, fs, q, gain_in, m_a.data(), m_b.data(), &m_gain));
        redesign(filter_type::band_stop);
    }
    template <typename other_real_t> void copy_coeff_from(const casc_2o_iir_bank<m_t, other_real_t> &o)
    {
        m_a = o.m_a;
        m_b = o.m_b;
        m_gain = o.m_gain;
        redesign(o.m_f_type);
    }

    // every channel's memory preloaded for a steady input (reference preload_filter, :197-214)
    void preload_filter(double value)
    {
        std::array<double, 3 * (m_t + 1)> mem{};
        detail::check(sdsp_hip_iir_preload(m_t, static_cast<int>(m_f_type), m_a.data(), m_b.data(), m_gain, value, mem.data()));
        std::vector<real_t> host(3 * (m_t + 1) * m_channels);
        for (size_t r = 0; r < 3 * (m_t + 1); ++r)
            for (std::uint64_t c = 0; c < m_channels; ++c)
                host[r * m_channels + c] = static_cast<real_t>(mem[r]);
        ensure_state();
        detail::check(sdsp_hip_memcpy_h2d(m_state, host.data(), host.size() * sizeof(real_t), m_device));
    }
    // forget the history (zero state)
    void reset()
    {
        if (m_state) {
            std::vector<real_t> zeros(3 * (m_t + 1) * m_channels, real_t(0));
            detail::check(sdsp_hip_memcpy_h2d(m_state, zeros.data(), zeros.size() * sizeof(real_t), m_device));
        }
    }

    // device pointer, asynchronous on `stream`; continues every channel's stream
    void process(real_t *device_data, std::uint64_t samples, std::uint64_t stride, void *stream = nullptr)
    {
        ensure_plan();
        ensure_state();
        detail::check(sdsp_hip_iir_process(m_plan, device_data, m_channels, samples, stride, m_state, stream));
    }
    // the sample-major "wire" layout: sample s of channel c at device_data[s*stride + c];
    // no transpose, bit-identical to process() on the transposed data
    void process_interleaved(real_t *device_data, std::uint64_t samples, std::uint64_t stride, void *stream = nullptr)
    {
        ensure_plan();
        ensure_state();
        detail::check(sdsp_hip_iir_process_interleaved(m_plan, device_data, m_channels, samples, stride, m_state, stream));
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
            rc = sdsp_hip_iir_process(m_plan, d, m_channels, samples, samples, m_state, nullptr);
        if (!rc)
            rc = sdsp_hip_memcpy_d2h(host_data, d, bytes, m_device);
        sdsp_hip_free(d, m_device);
        detail::check(rc);
    }
    std::uint64_t channels() const noexcept { return m_channels; }

private:
    template <size_t, typename> friend class casc_2o_iir_bank;
    void redesign(filter_type t)
    {
        m_f_type = t;
        if (m_plan) {
            sdsp_hip_iir_plan_destroy(m_plan);
            m_plan = nullptr;
        }
    }
    void ensure_plan()
    {
        if (!m_plan)
            detail::check(sdsp_hip_iir_plan_create(&m_plan, m_t, m_kind, m_a.data(), m_b.data(), m_gain,
                                                   detail::precision_of<real_t>::value, m_device));
    }
    void ensure_state()
    {
        if (!m_state) {
            const size_t bytes = 3 * (m_t + 1) * m_channels * sizeof(real_t);
            detail::check(sdsp_hip_malloc(&m_state, bytes, m_device));
            std::vector<real_t> zeros(3 * (m_t + 1) * m_channels, real_t(0));
            detail::check(sdsp_hip_memcpy_h2d(m_state, zeros.data(), bytes, m_device));
        }
    }

    std::uint64_t m_channels;
    int m_kind;
    int m_device;
    std::array<double, 3 * m_t> m_a{}, m_b{};
    double m_gain{ 1.0 };
    filter_type m_f_type{ filter_type::none };
    sdsp_hip_iir_plan *m_plan{ nullptr };
    void *m_state{ nullptr };
};
} // namespace sdsp

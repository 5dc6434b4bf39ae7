// sdsp/detail/hip_runtime.h -- glue between the sdsp:: headers and the C ABI (sdsp_hip.h).
#pragma once

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>

#include "../../sdsp_hip.h"

namespace sdsp
{
// Thrown when the MI355X path cannot run (no device, HIP error, unsupported shape).  The
// reference has no run-time errors on this path; there is deliberately no CPU fallback here.
class hip_error : public std::runtime_error {
public:
    hip_error(int code, const std::string &what) : std::runtime_error(what), m_code(code) {}
    int code() const noexcept { return m_code; }

private:
    int m_code;
};

namespace detail
{
inline void check(int rc)
{
    if (rc != SDSP_HIP_OK)
        throw hip_error(rc, std::string("sdsp_hip: ") + sdsp_hip_last_error_string());
}

template <typename real_t> struct precision_of;
template <> struct precision_of<float> { static constexpr int value = SDSP_HIP_F32; };
template <> struct precision_of<double> { static constexpr int value = SDSP_HIP_F64; };

// RAII owner of an FFT plan
class fft_plan_handle {
public:
    fft_plan_handle(std::uint32_t n, int radix, int direction, int precision, std::uint64_t max_batch, int device)
    {
        check(sdsp_hip_fft_plan_create(&m_plan, n, radix, direction, precision, max_batch, device));
    }
    ~fft_plan_handle() { sdsp_hip_fft_plan_destroy(m_plan); }
    fft_plan_handle(const fft_plan_handle &) = delete;
    fft_plan_handle &operator=(const fft_plan_handle &) = delete;
    sdsp_hip_fft_plan *get() const noexcept { return m_plan; }
    std::mutex &mutex() noexcept { return m_mutex; } // the plan's host staging buffer is not re-entrant

private:
    sdsp_hip_fft_plan *m_plan{ nullptr };
    std::mutex m_mutex;
};
} // namespace detail
} // namespace sdsp

// fft1m.hip -- host side of the batched N = 2^20 radix-2 complex f32 FFT (BASELINE config 3); the kernels and their
// description live in fft1m_kernels.h.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "fft1m_kernels.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
using namespace fft1m;

// One pass over one chunk of `count` transforms: which = 1 columns (data -> workspace),
// which = 2 rows (workspace -> data).  The workspace holds `count` matrices.  (The plan's variant 1.)
int launch_fft1m_pass(const fft1m_args &a, int which, void *stream)
{
    if (a.count == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const uint64_t blocks = a.count * kTiles;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "chunk too large for one launch");
    const float2 *in = reinterpret_cast<const float2 *>(a.data);
    float2 *out = reinterpret_cast<float2 *>(a.data);
    float2 *ws = reinterpret_cast<float2 *>(a.workspace);
    const float2 *tw1k = reinterpret_cast<const float2 *>(a.tw_1024);
    static std::atomic<uint64_t> done[4];
    const void *kerns[4] = { reinterpret_cast<const void *>(sdsp_fft1m_cols<false, MODE_FFT, WS_BLOCKED>),
                             reinterpret_cast<const void *>(sdsp_fft1m_cols<true, MODE_FFT, WS_BLOCKED>),
                             reinterpret_cast<const void *>(sdsp_fft1m_rows<false, MODE_FFT, WS_BLOCKED>),
                             reinterpret_cast<const void *>(sdsp_fft1m_rows<true, MODE_FFT, WS_BLOCKED>) };
    const int idx = (which == 1 ? 0 : 2) + (a.reverse ? 1 : 0);
    if (int rc = ensure_dynamic_lds(kerns[idx], kLdsBytes, done[idx]))
        return rc;
    const dim3 grid((uint32_t)blocks), block(kThreads);
    if (which == 1) {
        if (a.reverse)
            hipLaunchKernelGGL((sdsp_fft1m_cols<true, MODE_FFT, WS_BLOCKED>), grid, block, kLdsBytes, s, in, ws, tw1k);
        else
            hipLaunchKernelGGL((sdsp_fft1m_cols<false, MODE_FFT, WS_BLOCKED>), grid, block, kLdsBytes, s, in, ws, tw1k);
    } else {
        if (a.reverse)
            hipLaunchKernelGGL((sdsp_fft1m_rows<true, MODE_FFT, WS_BLOCKED>), grid, block, kLdsBytes, s, ws, out, tw1k, a.scale);
        else
            hipLaunchKernelGGL((sdsp_fft1m_rows<false, MODE_FFT, WS_BLOCKED>), grid, block, kLdsBytes, s, ws, out, tw1k, a.scale);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft1m launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

namespace
{
int resident_grid()
{
    // two 512-thread workgroups per CU (76 KiB of LDS, <= 128 VGPRs); a larger grid would be correct too (tickets are
    // drawn by running workgroups only), a smaller one leaves CUs idle
    // per device: a node may mix parts with different CU counts, and a plan runs on its own device
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return 512;
    std::atomic<int> &slot = cached[dev & 63];
    int g = slot.load();
    if (g)
        return g;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return 512;
    g = 2 * prop.multiProcessorCount;
    slot.store(g);
    return g;
}

template <bool REV, int LAYOUT> int launch_fused_t(const fft1m_fused_args &a, hipStream_t s)
{
    static std::atomic<uint64_t> done{ 0 };
    auto kern = sdsp_fft1m_fused<REV, MODE_FFT, LAYOUT>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), kLdsBytes, done))
        return rc;
    fused_args k;
    k.data = reinterpret_cast<float2 *>(a.data);
    k.ws = reinterpret_cast<float2 *>(a.workspace);
    k.tw_1024 = reinterpret_cast<const float2 *>(a.tw_1024);
    k.sync = reinterpret_cast<unsigned *>(a.sync);
    k.count = (uint32_t)a.count;
    k.ring = a.ring;
    k.lag = a.lag;
    k.queues = a.queues;
    k.flags = a.spin_limit == 0 ? 8u : 0u; // a bound of zero ticks = fault injection: every hand-off wait gives up (tests)
    k.sleep = 0;
    k.scale = a.scale;
    k.sticky = reinterpret_cast<unsigned *>(a.sticky);
    k.spin_limit = a.spin_limit; // default 2 s at 100 MHz: far beyond any real wait; a lost hand-off aborts instead of hanging
    const uint32_t grid = (uint32_t)resident_grid();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), kLdsBytes, s, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft1m fused launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace

size_t fft1m_sync_bytes(uint64_t count, uint32_t queues) { return fused_sync_words((uint32_t)count, queues) * sizeof(unsigned); }

// One persistent launch over `count` transforms (the plan's default).  a.sync: fft1m_sync_bytes(count) bytes of device
// memory, zeroed here on the stream before the launch (Guideline 16: re-initialise every call).
int launch_fft1m_fused(const fft1m_fused_args &a, void *stream)
{
    if (a.count == 0)
        return SDSP_HIP_OK;
    if (a.ring == 0 || a.lag >= a.ring || a.queues == 0 || a.count > 0x00ffffffu)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "fft1m fused: need lag < ring, queues > 0 and a sane count");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(a.sync, 0, fft1m_sync_bytes(a.count, a.queues), s);
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft1m sync memset: ") + hipGetErrorString(e));
    constexpr int kLayout = WS_BLOCKED | WS_SC1_STORES;
    return a.reverse ? launch_fused_t<true, kLayout>(a, s) : launch_fused_t<false, kLayout>(a, s);
}
} // namespace sdsp_hip

// fft4096.hip -- host side of the batched N = 4096 complex f32 FFT kernels (BASELINE configs 2 and 5); the kernels and
// their description live in fft4096_kernels.h.
#include <hip/hip_runtime.h>

#include "fft4096_kernels.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
using namespace fft4096;

namespace
{
template <int BAR, int SORD, int LORD, int LDSB, int WAVES> void launch_r4(const fft4096_args &a, hipStream_t s)
{
    float2 *d = reinterpret_cast<float2 *>(a.data);
    const float2 *w = reinterpret_cast<const float2 *>(a.tw);
    const dim3 grid((uint32_t)a.batch);
    if (a.reverse)
        hipLaunchKernelGGL((sdsp_fft4096_r4_f32<true, BAR, SORD, LORD, LDSB, WAVES>), grid, dim3(256), 0, s, d, w, a.batch, a.scale);
    else
        hipLaunchKernelGGL((sdsp_fft4096_r4_f32<false, BAR, SORD, LORD, LDSB, WAVES>), grid, dim3(256), 0, s, d, w, a.batch, a.scale);
}
constexpr int kNumVariants = 3;
} // namespace

int launch_fft4096_r2_f32(const fft4096_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    if (a.batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float2 *d = reinterpret_cast<float2 *>(a.data);
    const float2 *w = reinterpret_cast<const float2 *>(a.tw);
    const dim3 grid((uint32_t)a.batch);
    if (a.reverse)
        hipLaunchKernelGGL((sdsp_fft4096_r2_f32<true>), grid, dim3(256), 0, s, d, w, a.batch, a.scale);
    else
        hipLaunchKernelGGL((sdsp_fft4096_r2_f32<false>), grid, dim3(256), 0, s, d, w, a.batch, a.scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft4096 r2 launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

int launch_fft4096_conv_f32(void *data, const void *tw, const void *h, uint64_t batch, void *stream)
{
    if (batch == 0)
        return SDSP_HIP_OK;
    if (batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sdsp_fft4096_conv_f32<true>, dim3((uint32_t)batch), dim3(256), 0, s, reinterpret_cast<float2 *>(data),
                       reinterpret_cast<const float2 *>(tw), reinterpret_cast<const float2 *>(h), batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft4096 conv launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

int fft4096_num_variants() { return kNumVariants; }

// Variants run the same arithmetic in the same order (bit-identical results) and differ in scheduling only
// (fft4096_kernels.h lists the knobs; DESIGN.md section 5.1 the measurements of the whole grid):
//   0  the default
//   1  round 1's default: no barrier after the last LDS read, stores in register order
//   2  barrier right before the stores
int launch_fft4096_r4_f32(const fft4096_args &a, int variant, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    if (variant < 0 || variant >= kNumVariants)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "unknown fft4096 variant");
    if (a.batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (variant) {
    case 0: launch_r4<1, 2, 0, 0, 3>(a, s); break;
    case 1: launch_r4<0, 0, 0, 0, 3>(a, s); break;
    default: launch_r4<2, 0, 0, 0, 3>(a, s); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft4096 launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace sdsp_hip

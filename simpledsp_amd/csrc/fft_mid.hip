// fft_mid.hip -- the two streaming passes of the three-pass schedule for N = 2^16 .. 2^23 (other than 2^20), f32,
// on gfx950.  (Above 2^19 the schedule nests: the rows are three-pass plans themselves.)
//
// These sizes exceed what one workgroup can hold, and the general four-step through the coverage kernel
// reaches only 7-15 % of HBM peak.  With N = 16 x N2 (N2 = 4096 .. 32768, i.e. a size one of the tuned
// single-pass kernels covers) the transform becomes
//
//   pass 1  sdsp_fft_col16_kernel   for every n2: the 16-point DFT down the column (row stride N2), times the
//                                   inter-pass twiddle W_N^(n2*k1)  (fft.h:286's factor for the combined
//                                   stages), data -> workspace.  One lane = two adjacent columns, every
//                                   access a 1 KiB contiguous piece of a row per wave instruction.
//   pass 2  (existing kernels)      16 contiguous length-N2 transforms per big transform, in place in the
//                                   workspace: fft4096.hip / fft_big.hip, 16 x batch of them in one launch
//   pass 3  sdsp_fft_untwist16      X[k1 + 16 k2] = Z[k1][k2]: a [16][N2] -> [N2][16] transpose through LDS,
//                                   workspace -> data; 2 KiB pieces in, 32 KiB contiguous out
//
// Three passes over HBM instead of one, each at 60-75 % of the peak: 20-25 % of the compulsory-bytes roofline
// (N = 65536: 24.6 % where the four-step through the coverage kernel gave 14.5 %; N = 262144: 23.5 % vs 7 %).
#include <hip/hip_runtime.h>

#include "fft_passes.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
typedef float v4f_t __attribute__((ext_vector_type(4)));

// pass 1: thread = columns (2c, 2c+1) of one transform; x[k] / y[k] = row k of those columns.
// Inter-pass twiddle W_N^m, m = n2*k1 < N = 2^L, as in fft1m.hip: coarse factor W_1024^(m >> (L-10)) from an LDS
// copy of the 1024-entry table, fine factor (angle below 2 pi / 1024) from two series terms -- exact to fp32
// rounding, and no gathers from the N-entry row (they cost more cache-line requests than the data).
template <bool REV>
__global__ __launch_bounds__(256) void sdsp_fft_col16_kernel(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                             const float2 *__restrict__ tw1024, uint32_t n2, float scale,
                                                             uint32_t fine_bits, float fine_step)
{
    __shared__ float2 w1k[1024];
    reinterpret_cast<float4 *>(w1k)[threadIdx.x] = reinterpret_cast<const float4 *>(tw1024)[threadIdx.x];
    reinterpret_cast<float4 *>(w1k)[threadIdx.x + 256] = reinterpret_cast<const float4 *>(tw1024)[threadIdx.x + 256];
    __syncthreads();
    auto twiddle = [&](uint32_t m) {
        const float th = (float)(m & ((1u << fine_bits) - 1)) * fine_step; // < 2 pi / 1024
        const float th2 = th * th;
        const float sn = th - th * th2 * 0.16666667f;
        const float2 fine = float2{ 1.0f - 0.5f * th2, REV ? sn : -sn };
        return passes::cmul(w1k[m >> fine_bits], fine);
    };
    const uint32_t pairs = n2 / 2;                                  // column pairs per transform
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x; // over transforms x column pairs (exact grid)
    const uint64_t xform = gid / pairs;
    const uint32_t c = (uint32_t)(gid % pairs) * 2;
    const uint64_t base = xform * 16ull * n2 + c;
    float2 x[16], y[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(in + base + (uint64_t)k * n2));
        x[k] = float2{ v.x, v.y };
        y[k] = float2{ v.z, v.w };
    }
    const float2 none[4] = {};
    passes::r2_pass<REV, false, 0>::run(x, none); // register k now holds output bit_reverse4(k)
    passes::r2_pass<REV, false, 0>::run(y, none);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t k1 = __brev((uint32_t)k) >> 28;
        float2 a = x[k], b = y[k];
        if (k1 != 0) { // W_N^(n2 * k1), N = 16 * n2: exponent < N
            a = passes::cmul(a, twiddle(c * k1));
            b = passes::cmul(b, twiddle((c + 1) * k1));
        }
        if constexpr (REV) { // the row transforms scale by 1/N2; the remaining 1/16 of reverse_fft::ScaleValues
            a.x *= scale; a.y *= scale;
            b.x *= scale; b.y *= scale;
        }
        // default cache policy: pass 2 reads this back at once
        *reinterpret_cast<float4 *>(out + base + (uint64_t)k1 * n2) = float4{ a.x, a.y, b.x, b.y };
    }
}

// pass 3: workgroup = 256 consecutive k2 of one transform: in[k1][k2] -> out[k2*16 + k1]
// float2 elements per LDS row.  The transposed ds_read_b64 of lane t touches row 2 (t & 7) (+1), column (t >> 3) + 32 j:
// within a 32-lane group the bank pair index is (2 (t & 7) P + (t >> 3)) mod 32, distinct for P = 2 (mod 32) (4 a + b);
// P = 257 measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 25 %.
constexpr int kPitch = 258;
__global__ __launch_bounds__(256) void sdsp_fft_untwist16(const float2 *__restrict__ in, float2 *__restrict__ out, uint32_t n2)
{
    __shared__ float2 tile[16 * kPitch];
    const uint32_t t = threadIdx.x;
    const uint32_t blocks_per_xform = n2 / 256;
    const uint64_t xform = blockIdx.x / blocks_per_xform;
    const uint32_t k2_0 = (blockIdx.x % blocks_per_xform) * 256;
    const float2 *src = in + xform * 16ull * n2 + k2_0 + t;
#pragma unroll
    for (int r = 0; r < 16; r++)
        tile[r * kPitch + t] = src[(uint64_t)r * n2]; // default policy: just written by pass 2
    __syncthreads();
    float2 *dst = out + xform * 16ull * n2 + (uint64_t)k2_0 * 16;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t e = 2 * (t + 256 * j); // output element within the 4096 of this workgroup
        const uint32_t k2 = e >> 4, k1 = e & 15;
        const float2 a = tile[k1 * kPitch + k2], b = tile[(k1 + 1) * kPitch + k2];
        const v4f_t v = { a.x, a.y, b.x, b.y };
        __builtin_nontemporal_store(v, reinterpret_cast<v4f_t *>(dst + e));
    }
}
// ---- the same two passes in double (the reference's own precision): one column per lane (a complex double is
// already 16 bytes), four series terms for the fine twiddle factor (t^8/8! < 1e-21 at t < 2 pi / 1024)
typedef double v2d_t __attribute__((ext_vector_type(2)));
template <bool REV>
__global__ __launch_bounds__(256) void sdsp_fft_col16_f64_kernel(const double2 *__restrict__ in, double2 *__restrict__ out,
                                                                 const double2 *__restrict__ tw1024, uint32_t n2, double scale,
                                                                 uint32_t fine_bits, double fine_step)
{
    __shared__ double2 w1k[1024];
#pragma unroll
    for (int i = 0; i < 4; i++)
        w1k[threadIdx.x + 256 * i] = tw1024[threadIdx.x + 256 * i];
    __syncthreads();
    auto twiddle = [&](uint32_t m) {
        const double th = (double)(m & ((1u << fine_bits) - 1)) * fine_step;
        const double t2 = th * th;
        const double cs = 1.0 - t2 * (0.5 - t2 * (1.0 / 24 - t2 * (1.0 / 720)));
        const double sn = th * (1.0 - t2 * (1.0 / 6 - t2 * (1.0 / 120 - t2 * (1.0 / 5040))));
        return passes::cmul(w1k[m >> fine_bits], double2{ cs, REV ? sn : -sn });
    };
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x; // over transforms x columns (exact grid)
    const uint64_t xform = gid / n2;
    const uint32_t c = (uint32_t)(gid % n2);
    const uint64_t base = xform * 16ull * n2 + c;
    double2 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(in + base + (uint64_t)k * n2));
        x[k] = double2{ v.x, v.y };
    }
    const double2 none[4] = {};
    passes::r2_pass<REV, false, 0>::run(x, none);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t k1 = __brev((uint32_t)k) >> 28;
        double2 a = x[k];
        if (k1 != 0)
            a = passes::cmul(a, twiddle(c * k1));
        if constexpr (REV) {
            a.x *= scale;
            a.y *= scale;
        }
        out[base + (uint64_t)k1 * n2] = a;
    }
}

constexpr int kPitch64 = 129; // double2 elements per LDS row
__global__ __launch_bounds__(128) void sdsp_fft_untwist16_f64(const double2 *__restrict__ in, double2 *__restrict__ out, uint32_t n2)
{
    __shared__ double2 tile[16 * kPitch64];
    const uint32_t t = threadIdx.x;
    const uint32_t blocks_per_xform = n2 / 128;
    const uint64_t xform = blockIdx.x / blocks_per_xform;
    const uint32_t k2_0 = (blockIdx.x % blocks_per_xform) * 128;
    const double2 *src = in + xform * 16ull * n2 + k2_0 + t;
#pragma unroll
    for (int r = 0; r < 16; r++)
        tile[r * kPitch64 + t] = src[(uint64_t)r * n2];
    __syncthreads();
    double2 *dst = out + xform * 16ull * n2 + (uint64_t)k2_0 * 16;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t e = t + 128 * j; // output element within the 2048 of this workgroup
        const double2 a = tile[(e & 15) * kPitch64 + (e >> 4)];
        const v2d_t v = { a.x, a.y };
        __builtin_nontemporal_store(v, reinterpret_cast<v2d_t *>(dst + e));
    }
}
} // namespace

int launch_fft_mid_cols(int precision, const void *in, void *out, const void *tw /* W_1024^j */, uint32_t n2, uint64_t batch,
                        int reverse, void *stream)
{
    if (precision == SDSP_HIP_F64) {
        const uint64_t blocks64 = batch * n2 / 256;
        if (blocks64 == 0 || blocks64 > 0x7fffffffull)
            return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
        hipStream_t s64 = reinterpret_cast<hipStream_t>(stream);
        const uint32_t fb = sdsp_hip_log2(n2) + 4 - 10;
        const double step = 6.283185307179586476925 / (double)(16ull * n2);
        const double2 *i64 = reinterpret_cast<const double2 *>(in);
        double2 *o64 = reinterpret_cast<double2 *>(out);
        const double2 *w64 = reinterpret_cast<const double2 *>(tw);
        if (reverse)
            hipLaunchKernelGGL(sdsp_fft_col16_f64_kernel<true>, dim3((uint32_t)blocks64), dim3(256), 0, s64, i64, o64, w64, n2, 1.0 / 16.0, fb, step);
        else
            hipLaunchKernelGGL(sdsp_fft_col16_f64_kernel<false>, dim3((uint32_t)blocks64), dim3(256), 0, s64, i64, o64, w64, n2, 1.0, fb, step);
        hipError_t e64 = hipGetLastError();
        if (e64 != hipSuccess)
            return fail(SDSP_HIP_ERR_HIP, std::string("fft_mid cols launch: ") + hipGetErrorString(e64));
        return SDSP_HIP_OK;
    }
    const uint64_t threads = batch * (n2 / 2);
    const uint64_t blocks = threads / 256; // n2 >= 4096: exact
    if (blocks == 0 || blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float2 *i = reinterpret_cast<const float2 *>(in);
    float2 *o = reinterpret_cast<float2 *>(out);
    const float2 *w = reinterpret_cast<const float2 *>(tw);
    const uint32_t log2n = sdsp_hip_log2(n2) + 4, fine_bits = log2n - 10;
    const float fine_step = (float)(6.283185307179586476925 / (double)(16ull * n2)); // 2 pi / N
    if (reverse)
        hipLaunchKernelGGL(sdsp_fft_col16_kernel<true>, dim3((uint32_t)blocks), dim3(256), 0, s, i, o, w, n2, 1.0f / 16.0f,
                           fine_bits, fine_step);
    else
        hipLaunchKernelGGL(sdsp_fft_col16_kernel<false>, dim3((uint32_t)blocks), dim3(256), 0, s, i, o, w, n2, 1.0f, fine_bits,
                           fine_step);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_mid cols launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

int launch_fft_mid_untwist(int precision, const void *in, void *out, uint32_t n2, uint64_t batch, void *stream)
{
    if (precision == SDSP_HIP_F64) {
        const uint64_t blocks64 = batch * (n2 / 128);
        if (blocks64 == 0 || blocks64 > 0x7fffffffull)
            return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
        hipLaunchKernelGGL(sdsp_fft_untwist16_f64, dim3((uint32_t)blocks64), dim3(128), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const double2 *>(in), reinterpret_cast<double2 *>(out), n2);
        hipError_t e64 = hipGetLastError();
        if (e64 != hipSuccess)
            return fail(SDSP_HIP_ERR_HIP, std::string("fft_mid untwist launch: ") + hipGetErrorString(e64));
        return SDSP_HIP_OK;
    }
    const uint64_t blocks = batch * (n2 / 256);
    if (blocks == 0 || blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sdsp_fft_untwist16, dim3((uint32_t)blocks), dim3(256), 0, s, reinterpret_cast<const float2 *>(in),
                       reinterpret_cast<float2 *>(out), n2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_mid untwist launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace sdsp_hip

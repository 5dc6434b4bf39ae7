// fft_reg64.hip -- the register-pass FFT family in double precision: N = 16 .. 8192, radix-2 stages
// (sdsp::fft_radix2, fft.h:258-299) or radix-4 stages (sdsp::fft_radix4, fft.h:301-360).  The
// reference computes in double; this is the batched fast path for callers who keep that precision
// (sdsp::fft_plan<double>).  Same structure as fft_reg.hip: coalesced 16-byte copies HBM <-> LDS,
// ceil(log2 N / 4) in-LDS passes of four radix-2 (two radix-4) DIF stages on 16 registers, bit / digit
// reversal folded into the final LDS read.  A workgroup owns max(N, 2048) points (34 KiB of LDS at
// 2048).  Results agree with the reference to its own bound 4*N*eps (tests/test_gpu_fft.py).
#include <hip/hip_runtime.h>

#include "fft_passes.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load(const double2 *p)
{
    const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(p));
    return double2{ v.x, v.y };
}
__device__ __forceinline__ void nt_store(double2 *p, double2 a)
{
    const v2d_t v = { a.x, a.y };
    __builtin_nontemporal_store(v, reinterpret_cast<v2d_t *>(p));
}

constexpr int points64_for(int log2n) { return log2n > 11 ? (1 << log2n) : 2048; }
__device__ __forceinline__ uint32_t slot(uint32_t p) { return p + (p >> 4); }
template <int RADIX, int LOG2N> __device__ __forceinline__ uint32_t reversed(uint32_t q)
{
    uint32_t r = __brev(q) >> (32 - LOG2N);
    if constexpr (RADIX == 4)
        r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
    return r;
}

template <int RADIX, int LOG2N, bool REV>
__global__ __launch_bounds__(points64_for(LOG2N) / 16) void sdsp_fft_reg_f64_kernel(double2 *__restrict__ data,
                                                                                  const double2 *__restrict__ tw,
                                                                                  uint64_t batch, double scale)
{
    constexpr int N = 1 << LOG2N;
    constexpr int kPoints = points64_for(LOG2N);
    constexpr int THREADS = kPoints / 16;
    constexpr int T = N / 16;
    constexpr int G = kPoints / N;
    constexpr int P = (LOG2N + 3) / 4;
    constexpr int LAST = LOG2N - 4 * (P - 1);
    static_assert(RADIX == 2 || (LOG2N % 2 == 0), "radix 4 needs a power of 4");
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft_reg64_smem[];
    double2 *lds = reinterpret_cast<double2 *>(sdsp_fft_reg64_smem);

    const uint32_t tid = threadIdx.x;
    const uint64_t first = (uint64_t)blockIdx.x * G;
    const uint64_t have = batch - first < (uint64_t)G ? batch - first : (uint64_t)G;
    const uint32_t live = (uint32_t)have * N;
    double2 *base = data + first * N;

    // 1. HBM -> LDS, one point (16 bytes) per lane, linear; full workgroups unpredicated
    if (have == (uint64_t)G) {
        double2 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++)
            v[k] = nt_load(base + tid + THREADS * k);
#pragma unroll
        for (int k = 0; k < 16; k++)
            lds[slot(tid + THREADS * k)] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t e = tid + THREADS * k;
            if (e < live)
                lds[slot(e)] = nt_load(base + e);
        }
    }
    __syncthreads();

    // 2. register passes, in place in LDS
    const uint32_t g = tid / T, t = tid % T;
    const uint32_t gbase = g * N;
    double2 x[16];
    auto run_pass = [&](auto pass_tag) {
        constexpr int I = decltype(pass_tag)::value;
        constexpr bool is_last = I == P - 1;
        constexpr int S = is_last ? 1 : (N >> (4 * (I + 1)));
        const uint32_t b = t / S, r = t % S;
        const uint32_t p0 = gbase + b * 16 * S + r;
#pragma unroll
        for (int k = 0; k < 16; k++)
            x[k] = lds[slot(p0 + S * k)];
        constexpr bool TW = S > 1;
        if constexpr (RADIX == 2) {
            double2 w[4];
            if constexpr (TW) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    w[j] = tw[(6 * I + j) * T + t]; // thread-twiddle table, see fft_reg.hip
            }
            passes::r2_pass<REV, TW, is_last ? 4 - LAST : 0>::run(x, w);
        } else {
            double2 w1[3], w2[3];
            if constexpr (TW) {
#pragma unroll
                for (int q = 1; q < 4; q++) {
                    w1[q - 1] = tw[(6 * I + q - 1) * T + t];
                    w2[q - 1] = tw[(6 * I + q + 2) * T + t];
                }
            }
            passes::r4_pass<REV, TW, !(is_last && LAST == 2)>(x, w1, w2);
        }
#pragma unroll
        for (int k = 0; k < 16; k++)
            lds[slot(p0 + S * k)] = x[k];
        __syncthreads();
    };
    run_pass(std::integral_constant<int, 0>{});
    if constexpr (P > 1)
        run_pass(std::integral_constant<int, 1>{});
    if constexpr (P > 2)
        run_pass(std::integral_constant<int, 2>{});
    if constexpr (P > 3)
        run_pass(std::integral_constant<int, 3>{});

    // 3. LDS -> HBM: X[q] sits at position reversed(q) of its transform
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t e = tid + THREADS * k;
        if (e < live) {
            const uint32_t tb = e & ~(uint32_t)(N - 1), q = e & (N - 1);
            double2 a = lds[slot(tb + reversed<RADIX, LOG2N>(q))];
            if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
                a.x *= scale;
                a.y *= scale;
            }
            nt_store(base + e, a);
        }
    }
}

template <int RADIX, int LOG2N> int launch_n(const fft_reg_args &a, hipStream_t s)
{
    constexpr int kPoints = points64_for(LOG2N);
    constexpr int G = kPoints >> LOG2N;
    constexpr size_t lds = (size_t)(kPoints + kPoints / 16) * sizeof(double2);
    const uint64_t blocks = (a.batch + G - 1) / G;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    auto launch = [&](auto kern) {
        if constexpr (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(kPoints / 16), lds, s, reinterpret_cast<double2 *>(a.data),
                           reinterpret_cast<const double2 *>(a.tw), a.batch, a.scale_d);
    };
    if (a.reverse)
        launch(sdsp_fft_reg_f64_kernel<RADIX, LOG2N, true>);
    else
        launch(sdsp_fft_reg_f64_kernel<RADIX, LOG2N, false>);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_reg f64 launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace

bool fft_reg64_supports(uint32_t n, int radix)
{
    if (n < 16 || n > 8192 || !sdsp_hip_is_power_of_2(n))
        return false;
    return radix == 2 || (radix == 4 && sdsp_hip_is_power_of_4(n));
}

int launch_fft_reg_f64(const fft_reg_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const uint32_t l = sdsp_hip_log2(a.n);
    if (a.radix == 2) {
        switch (l) {
        case 4: return launch_n<2, 4>(a, s);
        case 5: return launch_n<2, 5>(a, s);
        case 6: return launch_n<2, 6>(a, s);
        case 7: return launch_n<2, 7>(a, s);
        case 8: return launch_n<2, 8>(a, s);
        case 9: return launch_n<2, 9>(a, s);
        case 10: return launch_n<2, 10>(a, s);
        case 11: return launch_n<2, 11>(a, s);
        case 12: return launch_n<2, 12>(a, s);
        case 13: return launch_n<2, 13>(a, s);
        default: break;
        }
    } else if (a.radix == 4) {
        switch (l) {
        case 4: return launch_n<4, 4>(a, s);
        case 6: return launch_n<4, 6>(a, s);
        case 8: return launch_n<4, 8>(a, s);
        case 10: return launch_n<4, 10>(a, s);
        case 12: return launch_n<4, 12>(a, s);
        default: break;
        }
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the f64 register-pass kernels");
}
} // namespace sdsp_hip

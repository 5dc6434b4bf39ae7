// fft_reg64.hip -- the register-pass FFT family in double precision: N = 16 .. 8192, radix-2 stages
// (sdsp::fft_radix2, fft.h:258-299) or radix-4 stages (sdsp::fft_radix4, fft.h:301-360).  The
// reference computes in double; this is the batched fast path for callers who keep that precision
// (sdsp::fft_plan<double>).  Same structure as fft_reg.hip: coalesced 16-byte copies HBM <-> LDS,
// ceil(log2 N / 4) in-LDS passes of four radix-2 (two radix-4) DIF stages on 16 registers, bit / digit
// reversal folded into the final LDS read.  A workgroup owns max(N, 1024) points (17 KiB of LDS at
// 1024; N = 2048: 34 KiB).  Results agree with the reference to its own bound 4*N*eps (tests/test_gpu_fft.py).
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

// Round 3: 1024-point tiles (one wave, 17 KiB of LDS: nine workgroups per CU) up to N = 1024 -- with 2048-point tiles (four 128-thread
// workgroups per CU) the family measured 68-71 % of HBM peak at N = 16 .. 1024, with these 75.3-77.2 % (the f32 family's finding, fft_reg.hip)
constexpr int points64_for(int log2n) { return log2n > 11 ? (1 << log2n) : (log2n <= 10 ? 1024 : 2048); }
__device__ __forceinline__ uint32_t slot(uint32_t p) { return p + (p >> 4); }
template <int RADIX, int LOG2N> __device__ __forceinline__ uint32_t reversed(uint32_t q)
{
    uint32_t r = __brev(q) >> (32 - LOG2N);
    if constexpr (RADIX == 4)
        r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
    return r;
}

// MODE as in fft_reg.hip: 0 complex transform; 1 / 2 real-input packing (the buffer holds 2N REAL doubles per
// transform, reinterpreted as N complex; forward split after the passes / inverse merge before them; tw2 = W_2N^k,
// direction-folded); 3 fused fast convolution data <- IFFT(FFT(data) .* h) (tw2 = h, natural order).
template <int RADIX, int LOG2N, bool REV, int MODE>
__global__ __launch_bounds__(points64_for(LOG2N) / 16) void sdsp_fft_reg_f64_kernel(double2 *__restrict__ data,
                                                                                  const double2 *__restrict__ tw,
                                                                                  const double2 *__restrict__ tw2,
                                                                                  uint64_t batch, double scale)
{
    using passes::cmul;
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

    if constexpr (MODE == 2) {
        // merge: natural-order packed half-spectrum -> Z (natural order), pairs (k, N-k)
        for (uint32_t idx = tid; idx < (uint32_t)(kPoints / 2); idx += THREADS) {
            const uint32_t tb = (idx / (N / 2)) * N, k = idx % (N / 2);
            if (k == 0) {
                const double2 x0 = lds[slot(tb)]; // (X[0], X[N]) both real
                lds[slot(tb)] = double2{ 0.5 * (x0.x + x0.y), 0.5 * (x0.x - x0.y) };
                const double2 xm = lds[slot(tb + N / 2)]; // X[N/2] = conj(Z[N/2])
                lds[slot(tb + N / 2)] = double2{ xm.x, -xm.y };
            } else {
                const double2 xa = lds[slot(tb + k)], xb = lds[slot(tb + N - k)];
                const double2 e = double2{ 0.5 * (xa.x + xb.x), 0.5 * (xa.y - xb.y) };  // (Xa + conj Xb)/2
                const double2 wo = double2{ 0.5 * (xa.x - xb.x), 0.5 * (xa.y + xb.y) }; // (Xa - conj Xb)/2
                const double2 o = cmul(wo, tw2[k]);                                     // tw2 is reverse-folded
                lds[slot(tb + k)] = double2{ e.x - o.y, e.y + o.x };                    // E + i O
                lds[slot(tb + N - k)] = double2{ e.x + o.y, o.x - e.y };                // conj(E - i O)
            }
        }
        __syncthreads();
    }

    // 2. register passes, in place in LDS
    const uint32_t g = tid / T, t = tid % T;
    const uint32_t gbase = g * N;
    double2 x[16];
    // RV: direction of this pass (differs from REV only in MODE 3, whose second half runs the reverse transform with the
    // conjugates of the forward twiddles)
    auto run_pass = [&](auto pass_tag, auto rev_tag) {
        constexpr int I = decltype(pass_tag)::value;
        constexpr bool RV = decltype(rev_tag)::value;
        constexpr bool CONJ = RV != REV;
        constexpr bool is_last = I == P - 1;
        constexpr int S = is_last ? 1 : (N >> (4 * (I + 1)));
        const uint32_t b = t / S, r = t % S;
        const uint32_t p0 = gbase + b * 16 * S + r;
#pragma unroll
        for (int k = 0; k < 16; k++)
            x[k] = lds[slot(p0 + S * k)];
        constexpr bool TW = S > 1;
        auto twl = [&](uint32_t v) { // thread-twiddle table, see fft_reg.hip
            double2 w = tw[(6 * I + v) * T + t];
            if constexpr (CONJ)
                w.y = -w.y;
            return w;
        };
        if constexpr (RADIX == 2) {
            double2 w[4];
            if constexpr (TW) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    w[j] = twl(j);
            }
            passes::r2_pass<RV, TW, is_last ? 4 - LAST : 0>::run(x, w);
        } else {
            double2 w1[3], w2[3];
            if constexpr (TW) {
#pragma unroll
                for (int q = 1; q < 4; q++) {
                    w1[q - 1] = twl(q - 1);
                    w2[q - 1] = twl(q + 2);
                }
            }
            passes::r4_pass<RV, TW, !(is_last && LAST == 2)>(x, w1, w2);
        }
#pragma unroll
        for (int k = 0; k < 16; k++)
            lds[slot(p0 + S * k)] = x[k];
        __syncthreads();
    };
    auto all_passes = [&](auto rev_tag) {
        run_pass(std::integral_constant<int, 0>{}, rev_tag);
        if constexpr (P > 1)
            run_pass(std::integral_constant<int, 1>{}, rev_tag);
        if constexpr (P > 2)
            run_pass(std::integral_constant<int, 2>{}, rev_tag);
        if constexpr (P > 3)
            run_pass(std::integral_constant<int, 3>{}, rev_tag);
    };
    all_passes(std::integral_constant<bool, REV>{});

    if constexpr (MODE == 3) {
        // the forward result sits in reversed order (position p holds Z[reversed(p)]); swap it back to natural order while
        // multiplying by H (tw2 = h, natural order), then run the reverse transform in place
        for (uint32_t idx = tid; idx < (uint32_t)kPoints; idx += THREADS) {
            const uint32_t tb = idx & ~(uint32_t)(N - 1), pp = idx & (N - 1);
            const uint32_t qq = reversed<RADIX, LOG2N>(pp);
            if (pp < qq) {
                const double2 a = lds[slot(tb + pp)], c = lds[slot(tb + qq)]; // a = Z[qq], c = Z[pp]
                lds[slot(tb + pp)] = cmul(c, tw2[pp]);
                lds[slot(tb + qq)] = cmul(a, tw2[qq]);
            } else if (pp == qq) {
                lds[slot(tb + pp)] = cmul(lds[slot(tb + pp)], tw2[pp]);
            }
        }
        __syncthreads();
        all_passes(std::integral_constant<bool, true>{});
    }

    if constexpr (MODE == 1) {
        // split: Z[k] sits at position reversed(k); pairs (k, N-k) are rewritten in place
        for (uint32_t idx = tid; idx < (uint32_t)(kPoints / 2); idx += THREADS) {
            const uint32_t tb = (idx / (N / 2)) * N, k = idx % (N / 2);
            if (k == 0) {
                const double2 z0 = lds[slot(tb)];
                lds[slot(tb)] = double2{ z0.x + z0.y, z0.x - z0.y }; // (X[0], X[N])
                const uint32_t pm = slot(tb + reversed<RADIX, LOG2N>(N / 2));
                const double2 zm = lds[pm];
                lds[pm] = double2{ zm.x, -zm.y }; // X[N/2] = conj(Z[N/2])
            } else {
                const uint32_t pa = slot(tb + reversed<RADIX, LOG2N>(k)), pb = slot(tb + reversed<RADIX, LOG2N>(N - k));
                const double2 za = lds[pa], zb = lds[pb];
                const double2 e = double2{ 0.5 * (za.x + zb.x), 0.5 * (za.y - zb.y) }; // (Za + conj Zb)/2
                const double2 d = double2{ 0.5 * (za.x - zb.x), 0.5 * (za.y + zb.y) }; // (Za - conj Zb)/2
                const double2 wd = cmul(d, tw2[k]);
                const double2 tt = double2{ wd.y, -wd.x };           // -i W D
                lds[pa] = double2{ e.x + tt.x, e.y + tt.y };         // X[k] = E + T
                lds[pb] = double2{ e.x - tt.x, tt.y - e.y };         // X[N-k] = conj(E - T)
            }
        }
        __syncthreads();
    }

    // 3. LDS -> HBM: X[q] sits at position reversed(q) of its transform
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t e = tid + THREADS * k;
        if (e < live) {
            const uint32_t tb = e & ~(uint32_t)(N - 1), q = e & (N - 1);
            double2 a = lds[slot(tb + reversed<RADIX, LOG2N>(q))];
            if constexpr (REV || MODE == 3) { // reverse_fft::ScaleValues, fft.h:128-132
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
                           reinterpret_cast<const double2 *>(a.tw), reinterpret_cast<const double2 *>(a.tw2), a.batch, a.scale_d);
    };
    if (a.real_mode == 1) // real forward
        launch(sdsp_fft_reg_f64_kernel<RADIX, LOG2N, false, 1>);
    else if (a.real_mode == 2) // real inverse
        launch(sdsp_fft_reg_f64_kernel<RADIX, LOG2N, true, 2>);
    else if (a.real_mode == 3) // fused convolution: forward plan, tw2 = h
        launch(sdsp_fft_reg_f64_kernel<RADIX, LOG2N, false, 3>);
    else if (a.reverse)
        launch(sdsp_fft_reg_f64_kernel<RADIX, LOG2N, true, 0>);
    else
        launch(sdsp_fft_reg_f64_kernel<RADIX, LOG2N, false, 0>);
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

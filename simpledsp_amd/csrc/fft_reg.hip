// fft_reg.hip -- register-pass FFT family for gfx950: f32, N = 16 .. 16384, radix-2 stages
// (sdsp::fft_radix2, fft.h:258-299) or radix-4 stages (sdsp::fft_radix4, fft.h:301-360), forward
// or reverse.  Fast path for every batched size the tuned N=4096 radix-4 kernel does not cover.
//
// A 128-thread workgroup owns 2048 complex points = 2048/N consecutive transforms (16 KiB of the
// batch, contiguous in HBM; N = 4096: 256 threads, one transform); for N = 8192 / 16384 a 512- / 1024-thread workgroup owns one transform:
//   1. the 32 KiB are copied HBM -> LDS with 16-byte lanes (fully coalesced, any N);
//   2. ceil(log2 N / 4) passes: a thread pulls 16 points (stride N/16^(i+1)) of one transform from
//      LDS into registers, runs four radix-2 DIF stages -- or two radix-4 DIF stages -- on them and
//      puts them back in place; the last pass runs whatever stages remain (1..4);
//   3. the result leaves LDS -> HBM coalesced again, the bit reversal (fft.h:269-273) or base-4 digit
//      reversal (fft.h:351-355) being folded into the LDS read address of that copy.
// Stage twiddles (fft.h:286 / :322-338) factor into a per-thread value read from the plan's
// HBM-resident row W_N^j (L1/L2 resident: <= 32 KiB) and a compile-time W_16 constant.
// LDS rows are padded by one slot per 16 so the strided pass accesses spread over the banks.
// DIF ordering: natural input, reversed output, so no permutation pass is needed up front; results
// agree with the reference to rounding (tests/test_gpu_fft.py, f32 tolerance 1e-6).
#include <hip/hip_runtime.h>

#include "fft_passes.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
using passes::cmul;
using passes::mul_w16;
using passes::r2_pass;
using passes::r4_pass;
using passes::rot90;
__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return passes::cadd(a, b); }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return passes::csub(a, b); }

typedef float v4f_t __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 gload16(const float4 *p)
{
    if constexpr (NT) {
        const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p));
        return float4{ v.x, v.y, v.z, v.w };
    } else {
        return *p;
    }
}
template <bool NT> __device__ __forceinline__ void gstore16(float4 *p, float4 a)
{
    if constexpr (NT) {
        const v4f_t v = { a.x, a.y, a.z, a.w };
        __builtin_nontemporal_store(v, reinterpret_cast<v4f_t *>(p));
    } else {
        *p = a;
    }
}

// complex points per workgroup: 2048 (128 threads) up to N = 2048, one whole transform above (256 / 512 / 1024 threads); LDS holds them
// with one padding slot per 16.  Round 3: the tile was 4096 points for every N <= 4096 -- four 256-thread workgroups per CU, whose load /
// compute / store phases are too coarse to keep the memory pipeline full: 68-74 % of HBM peak at N = 16 .. 2048.  With 2048-point tiles
// (eight workgroups of 128 per CU, the same bytes in LDS) the same kernel measures 75.8-78.9 % at every one of those sizes, either radix,
// one call (1024-point tiles: 77.1 % at N = 16, 74.5 % at 64 / 128, 76.4 % at 512: 2048 is the better compromise)
constexpr int points_for(int log2n) { return log2n > 12 ? (1 << log2n) : (log2n <= 11 ? 2048 : 4096); }
__device__ __forceinline__ uint32_t slot(uint32_t p) { return p + (p >> 4); }

// reversed index of q within an N-point transform: bit reversal (radix 2) or base-4 digit reversal
template <int RADIX, int LOG2N> __device__ __forceinline__ uint32_t reversed(uint32_t q)
{
    uint32_t r = __brev(q) >> (32 - LOG2N);
    if constexpr (RADIX == 4)
        r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
    return r;
}

// MODE 0: complex transform.  MODE 1 / 2 (real-input packing): the buffer holds 2N REAL
// samples per transform, reinterpreted as N complex z[m] = x[2m] + i x[2m+1].
//   MODE 1 (forward): after the passes an in-LDS split turns Z = FFT_N(z) into the first half of the
//     2N-point spectrum of x:  X[k] = E + T, X[N-k] = conj(E - T), E = (Z[k] + conj(Z[N-k]))/2,
//     T = -i W_2N^k (Z[k] - conj(Z[N-k]))/2; packed in place: out[0] = (X[0], X[N]) (both real).
//   MODE 2 (inverse): the packed half-spectrum is merged back (the same algebra inverted) before the
//     passes; the reverse transform then yields z, i.e. the real samples.
// Half the HBM bytes of pushing a real signal through the complex transform (what every reference
// test does, testFFT.cpp:23-25,84-90).  tw2 = W_2N^k, direction-folded like tw.
// MODE 3: fused fast convolution data <- IFFT(FFT(data) .* h), tw2 = h (see below).
template <int RADIX, int LOG2N, bool REV, bool NT, int MODE>
__global__ __launch_bounds__(points_for(LOG2N) / 16) void sdsp_fft_reg_kernel(float2 *__restrict__ data,
                                                                             const float2 *__restrict__ tw,
                                                                             const float2 *__restrict__ tw2,
                                                                             uint64_t batch, float scale)
{
    // `tw` is the plan's THREAD-TWIDDLE table (capi.hip: upload_thread_twiddles_reg): for pass I and value
    // slot v, entry [(6 I + v) * T + t] is the twiddle thread t of a transform needs -- the same rounded
    // values as the row W_N^j, but read with coalesced loads instead of gathers at strides of 8..96 B per lane
    // (which cost about as many cache-line requests as the data itself; see fft4096.hip).
    constexpr int N = 1 << LOG2N;
    constexpr int kPoints = points_for(LOG2N);
    constexpr int THREADS = kPoints / 16;
    constexpr int T = N / 16;             // threads per transform
    constexpr int G = kPoints / N;        // transforms per workgroup
    constexpr int P = (LOG2N + 3) / 4;    // register passes
    constexpr int LAST = LOG2N - 4 * (P - 1); // radix-2 stages left for the last pass (1..4)
    static_assert(RADIX == 2 || (LOG2N % 2 == 0), "radix 4 needs a power of 4");
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft_reg_smem[];
    float2 *lds = reinterpret_cast<float2 *>(sdsp_fft_reg_smem); // kPoints + kPoints/16 slots

    const uint32_t tid = threadIdx.x;
    const uint64_t first = (uint64_t)blockIdx.x * G;             // first transform of this workgroup
    const uint64_t have = batch - first < (uint64_t)G ? batch - first : (uint64_t)G;
    const uint32_t live = (uint32_t)have * N;                    // valid points (ragged last workgroup)
    float2 *base = data + first * N;

    // 1. HBM -> LDS, 16 bytes per lane (two points), linear.  Full workgroups (all but possibly the
    // last) take the unpredicated path: eight back-to-back loads with immediate offsets.
    const bool whole = have == (uint64_t)G;
    if (whole) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = gload16<NT>(reinterpret_cast<const float4 *>(base) + tid + THREADS * k);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t e = 2 * (tid + THREADS * k);
            lds[slot(e)] = float2{ v[k].x, v[k].y };
            lds[slot(e + 1)] = float2{ v[k].z, v[k].w };
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t e = 2 * (tid + THREADS * k);
            if (e < live) {
                const float4 v = gload16<NT>(reinterpret_cast<const float4 *>(base + e));
                lds[slot(e)] = float2{ v.x, v.y };
                lds[slot(e + 1)] = float2{ v.z, v.w };
            }
        }
    }
    __syncthreads();

    if constexpr (MODE == 2) {
        // merge: natural-order packed half-spectrum -> Z (natural order), pairs (k, N-k)
        for (uint32_t idx = tid; idx < (uint32_t)(kPoints / 2); idx += THREADS) {
            const uint32_t tb = (idx / (N / 2)) * N, k = idx % (N / 2);
            if (k == 0) {
                const float2 x0 = lds[slot(tb)];            // (X[0], X[N]) both real
                lds[slot(tb)] = float2{ 0.5f * (x0.x + x0.y), 0.5f * (x0.x - x0.y) };
                const float2 xm = lds[slot(tb + N / 2)];    // X[N/2] = conj(Z[N/2])
                lds[slot(tb + N / 2)] = float2{ xm.x, -xm.y };
            } else {
                const float2 xa = lds[slot(tb + k)], xb = lds[slot(tb + N - k)];
                const float2 e = float2{ 0.5f * (xa.x + xb.x), 0.5f * (xa.y - xb.y) };  // (Xa + conj Xb)/2
                const float2 wo = float2{ 0.5f * (xa.x - xb.x), 0.5f * (xa.y + xb.y) }; // (Xa - conj Xb)/2
                const float2 o = cmul(wo, tw2[k]);                                      // conj(W)·(W·O): tw2 is reverse-folded
                lds[slot(tb + k)] = float2{ e.x - o.y, e.y + o.x };                     // E + i O
                lds[slot(tb + N - k)] = float2{ e.x + o.y, o.x - e.y };                 // conj(E - i O)
            }
        }
        __syncthreads();
    }

    // 2. register passes, in place in LDS
    const uint32_t g = tid / T, t = tid % T; // transform within the workgroup, thread within it
    const uint32_t gbase = g * N;
    float2 x[16];
    // one register pass; RV = direction of this pass (differs from REV only in MODE 3, whose second
    // half runs the reverse transform with the conjugates of the forward twiddles)
    auto run_pass = [&](auto pass_tag, auto rev_tag) {
        constexpr int I = decltype(pass_tag)::value;
        constexpr bool RV = decltype(rev_tag)::value;
        constexpr bool CONJ = RV != REV;
        constexpr bool is_last = I == P - 1;
        constexpr int S = is_last ? 1 : (N >> (4 * (I + 1))); // point stride of this pass
        // thread (b, r): positions b*16*S + r + S*k
        const uint32_t b = t / S, r = t % S;
        const uint32_t p0 = gbase + b * 16 * S + r;
#pragma unroll
        for (int k = 0; k < 16; k++)
            x[k] = lds[slot(p0 + S * k)];
        constexpr bool TW = S > 1;          // r == 0 in the stride-1 pass: all thread twiddles are 1
        // slot v of pass I: radix 2: W_N^(unit << v), v < 4; radix 4: W_N^(unit (v+1)), v < 3, then
        // W_N^(4 unit (v-2)), v = 3..5; unit = r * 16^I
        auto twl = [&](uint32_t v) {
            float2 w = tw[(6 * I + v) * T + t];
            if constexpr (CONJ)
                w.y = -w.y;
            return w;
        };
        if constexpr (RADIX == 2) {
            float2 w[4];
            if constexpr (TW) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    w[j] = twl(j);
            }
            r2_pass<RV, TW, is_last ? 4 - LAST : 0>::run(x, w);
        } else {
            float2 w1[3], w2[3];
            if constexpr (TW) {
#pragma unroll
                for (int q = 1; q < 4; q++) {
                    w1[q - 1] = twl(q - 1);
                    w2[q - 1] = twl(q + 2);
                }
            }
            r4_pass<RV, TW, !(is_last && LAST == 2)>(x, w1, w2);
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
        // fused fast convolution for the whole family: the forward result sits in
        // reversed order (position p holds Z[reversed(p)]); swap it back to natural order while
        // multiplying by H (tw2 = h, natural order), then run the reverse transform in place.
        for (uint32_t idx = tid; idx < (uint32_t)kPoints; idx += THREADS) {
            const uint32_t tb = idx & ~(uint32_t)(N - 1), pp = idx & (N - 1);
            const uint32_t qq = reversed<RADIX, LOG2N>(pp);
            if (pp < qq) {
                const float2 a = lds[slot(tb + pp)], c = lds[slot(tb + qq)]; // a = Z[qq], c = Z[pp]
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
                const float2 z0 = lds[slot(tb)];
                lds[slot(tb)] = float2{ z0.x + z0.y, z0.x - z0.y }; // (X[0], X[N])
                const uint32_t pm = slot(tb + reversed<RADIX, LOG2N>(N / 2));
                const float2 zm = lds[pm];
                lds[pm] = float2{ zm.x, -zm.y };                    // X[N/2] = conj(Z[N/2])
            } else {
                const uint32_t pa = slot(tb + reversed<RADIX, LOG2N>(k)), pb = slot(tb + reversed<RADIX, LOG2N>(N - k));
                const float2 za = lds[pa], zb = lds[pb];
                const float2 e = float2{ 0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y) }; // (Za + conj Zb)/2
                const float2 d = float2{ 0.5f * (za.x - zb.x), 0.5f * (za.y + zb.y) }; // (Za - conj Zb)/2
                const float2 wd = cmul(d, tw2[k]);
                const float2 tt = float2{ wd.y, -wd.x };                               // -i W D
                lds[pa] = float2{ e.x + tt.x, e.y + tt.y };                            // X[k] = E + T
                lds[pb] = float2{ e.x - tt.x, tt.y - e.y };                            // X[N-k] = conj(E - T)
            }
        }
        __syncthreads();
    }

    // 3. LDS -> HBM: X[q] sits at position reversed(q) of its transform
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t e = 2 * (tid + THREADS * k);
        if (whole || e < live) {
            const uint32_t tb = e & ~(uint32_t)(N - 1), q = e & (N - 1);
            float2 a = lds[slot(tb + reversed<RADIX, LOG2N>(q))];
            float2 c = lds[slot(tb + reversed<RADIX, LOG2N>(q + 1))];
            if constexpr (REV || MODE == 3) { // reverse_fft::ScaleValues, fft.h:128-132
                a.x *= scale;
                a.y *= scale;
                c.x *= scale;
                c.y *= scale;
            }
            gstore16<NT>(reinterpret_cast<float4 *>(base + e), float4{ a.x, a.y, c.x, c.y });
        }
    }
}

template <int RADIX, int LOG2N, bool REV, bool NT, int MODE>
int launch_one(const fft_reg_args &a, dim3 grid, hipStream_t s)
{
    constexpr int kPoints = points_for(LOG2N);
    constexpr size_t lds = (size_t)(kPoints + kPoints / 16) * sizeof(float2);
    auto kern = sdsp_fft_reg_kernel<RADIX, LOG2N, REV, NT, MODE>;
    if constexpr (lds > 64 * 1024) {
        static std::atomic<uint64_t> attr_done{ 0 };
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done))
            return rc;
    }
    hipLaunchKernelGGL(kern, grid, dim3(kPoints / 16), lds, s, reinterpret_cast<float2 *>(a.data),
                       reinterpret_cast<const float2 *>(a.tw), reinterpret_cast<const float2 *>(a.tw2), a.batch, a.scale);
    return SDSP_HIP_OK;
}

template <int RADIX, int LOG2N> int launch_n(const fft_reg_args &a, hipStream_t s)
{
    constexpr int G = points_for(LOG2N) >> LOG2N;
    const uint64_t blocks = (a.batch + G - 1) / G;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    const dim3 grid((uint32_t)blocks);
    int rc;
    if (a.real_mode == 1) { // real forward: always the forward direction, streaming accesses
        rc = launch_one<RADIX, LOG2N, false, true, 1>(a, grid, s);
    } else if (a.real_mode == 2) { // real inverse
        rc = launch_one<RADIX, LOG2N, true, true, 2>(a, grid, s);
    } else if (a.real_mode == 3) { // fused convolution: forward plan, tw2 = h
        rc = launch_one<RADIX, LOG2N, false, true, 3>(a, grid, s);
    } else if (a.nontemporal) {
        rc = a.reverse ? launch_one<RADIX, LOG2N, true, true, 0>(a, grid, s) : launch_one<RADIX, LOG2N, false, true, 0>(a, grid, s);
    } else {
        rc = a.reverse ? launch_one<RADIX, LOG2N, true, false, 0>(a, grid, s) : launch_one<RADIX, LOG2N, false, false, 0>(a, grid, s);
    }
    if (rc)
        return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_reg launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace

bool fft_reg_supports(uint32_t n, int radix)
{
    if (n < 16 || n > 16384 || !sdsp_hip_is_power_of_2(n))
        return false;
    return radix == 2 || (radix == 4 && sdsp_hip_is_power_of_4(n));
}

int launch_fft_reg_f32(const fft_reg_args &a, void *stream)
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
        case 14: return launch_n<2, 14>(a, s);
        default: break;
        }
    } else if (a.radix == 4) {
        switch (l) {
        case 4: return launch_n<4, 4>(a, s);
        case 6: return launch_n<4, 6>(a, s);
        case 8: return launch_n<4, 8>(a, s);
        case 10: return launch_n<4, 10>(a, s);
        case 12: return launch_n<4, 12>(a, s);
        case 14: return launch_n<4, 14>(a, s);
        default: break;
        }
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the register-pass kernels");
}
} // namespace sdsp_hip

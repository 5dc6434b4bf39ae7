// fft_wave.hip -- batched N = 1024 complex f32 FFT, radix-2 (fft.h:258-299) or radix-4 (fft.h:301-360) stages, one
// transform per WAVE, for gfx950.
//
// The register-pass family (fft_reg.hip) stages a 4096-point block through LDS with a 16-byte copy phase on either side
// of its in-LDS passes and three workgroup barriers; at N = 1024 it reads 68.7-71.4 % of HBM peak.  N = 1024 = 64 lanes x
// 16 points is exactly what ONE wave holds in registers, so here a wave is autonomous:
//
//   load     x[k] = data[t + 64 k]                     512 contiguous bytes per wave instruction, straight into registers
//   pass A   stages of pair distances 512 .. 64        (radix 2: four stages; radix 4: two), thread twiddles W_N^(t ..)
//   exchange position t + 64 k  ->  64 b + v + 4 j     b = t >> 2, v = t & 3: through the wave's own 8 KiB of LDS
//   pass B   distances 32 .. 4 inside 64-point blocks  thread twiddles W_64^(v ..)
//   exchange position 64 b + v + 4 j  ->  16 w + i     w = reversed(t)
//   pass C   the last two radix-2 stages / the last radix-4 stage on 16 contiguous positions, constants only
//   store    X[t + 64 reversed(i)] = x[i]              512 contiguous bytes per wave instruction: the bit / digit
//                                                      reversal (fft.h:269-273, :351-355) is folded into the choice of w
//
// No workgroup barrier anywhere: LDS serves a wave's instructions in order, so a wave's reads see its own earlier writes;
// the four waves of a workgroup share nothing but the launch.  Arithmetic, stage order and thread twiddles (the plan's
// [pass][value][thread] table, capi.hip: upload_thread_twiddles_reg, 64 threads per transform) are those of the
// register-pass family: the results are bit-identical to it (tests/test_gpu_fft.py holds both to the oracle and to
// each other).
//
// LDS slots (8-byte units) are p ^ X(p >> 5) with X linear over GF(2), chosen by search (tools/model_fft_wave.py) so that
// all three access patterns of both radices hit 32 distinct 8-byte bank pairs in each half-wave.
#include <hip/hip_runtime.h>

#include "fft_passes.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 gload(const float2 *p)
{
    const v2f_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(p));
    return float2{ v.x, v.y };
}
__device__ __forceinline__ void gstore(float2 *p, float2 a)
{
    __builtin_nontemporal_store(v2f_t{ a.x, a.y }, reinterpret_cast<v2f_t *>(p));
}

// X(h) for h = p >> 5 (five bits): XOR of the rows of the bits set in h
constexpr uint32_t kRow[5] = { 2, 30, 15, 25, 26 };
constexpr uint32_t xterm(uint32_t h)
{
    uint32_t x = 0;
    for (int b = 0; b < 5; b++)
        if ((h >> b) & 1)
            x ^= kRow[b];
    return x;
}
__device__ __forceinline__ uint32_t xterm_dev(uint32_t h)
{
    uint32_t x = 0;
#pragma unroll
    for (int b = 0; b < 5; b++)
        x ^= ((h >> b) & 1) ? kRow[b] : 0u;
    return x;
}

template <int RADIX, bool REV>
__global__ __launch_bounds__(256) void sdsp_fft1024_wave_f32(float2 *__restrict__ data, const float2 *__restrict__ tw, uint64_t batch,
                                                             float scale)
{
    __shared__ __attribute__((aligned(16))) float2 lds_all[4][1024];
    const uint32_t t = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float2 *lds = lds_all[wave];
    const uint64_t f = static_cast<uint64_t>(blockIdx.x) * 4 + wave;
    if (f >= batch)
        return; // wave-uniform; the kernel has no barrier
    // thread twiddles: [pass][value][thread], 64 threads per transform (fft_reg.hip: twl)
    auto twl = [&](int pass, int v) { return tw[(6 * pass + v) * 64 + t]; };

    float2 x[16];
    const float2 *src = data + f * 1024 + t;
#pragma unroll
    for (int k = 0; k < 16; k++)
        x[k] = gload(src + 64 * k);

    auto run_pass = [&](auto pass_tag) {
        constexpr int I = decltype(pass_tag)::value;
        if constexpr (RADIX == 2) {
            float2 w[4];
            if constexpr (I < 2) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    w[j] = twl(I, j);
            }
            passes::r2_pass<REV, (I < 2), (I == 2 ? 2 : 0)>::run(x, w);
        } else {
            float2 w1[3], w2[3];
            if constexpr (I < 2) {
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    w1[q] = twl(I, q);
                    w2[q] = twl(I, q + 3);
                }
            }
            passes::r4_pass<REV, (I < 2), (I < 2)>(x, w1, w2);
        }
    };

    run_pass(std::integral_constant<int, 0>{});

    // ---- exchange A -> B.  A: p = t + 64 k, p >> 5 = (t >> 5) + 2 k
    {
        uint32_t ta = t ^ ((t >> 5) ? kRow[0] : 0u);
        asm volatile("" : "+v"(ta)); // one v_xor per access instead of sixteen live addresses
#pragma unroll
        for (int k = 0; k < 16; k++)
            lds[64 * k + (ta ^ xterm(2u * k))] = x[k];
    }
    // B: p = 64 b + v + 4 j, p >> 5 = 2 b + (j >> 3)
    uint32_t tb = 64u * (t >> 2) + ((t & 3u) ^ xterm_dev(2u * (t >> 2)));
    {
        uint32_t a = tb;
        asm volatile("" : "+v"(a));
#pragma unroll
        for (int j = 0; j < 16; j++)
            x[j] = lds[a ^ ((4u * j) ^ xterm(j >> 3))];
    }

    run_pass(std::integral_constant<int, 1>{});

    // ---- exchange B -> C (pass B's slots are written by the thread that read them)
    {
        uint32_t a = tb;
        asm volatile("" : "+v"(a));
#pragma unroll
        for (int j = 0; j < 16; j++)
            lds[a ^ ((4u * j) ^ xterm(j >> 3))] = x[j];
    }
    // C: p = 16 w + i, p >> 5 = w >> 1; w = reversed(t) so that the outputs land at t + 64 * reversed(i)
    const uint32_t w = RADIX == 2 ? (__brev(t) >> 26) : (((t & 3u) << 4) | (t & 12u) | (t >> 4));
    {
        uint32_t a = (16u * w) ^ xterm_dev(w >> 1);
        asm volatile("" : "+v"(a));
#pragma unroll
        for (int i = 0; i < 16; i++)
            x[i] = lds[a ^ (uint32_t)i];
    }

    run_pass(std::integral_constant<int, 2>{});

    float2 *dst = data + f * 1024 + t;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        float2 v = x[i];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            v.x *= scale;
            v.y *= scale;
        }
        const int row = RADIX == 2 ? (int)(__brev((uint32_t)i) >> 28) : 4 * (i & 3) + (i >> 2);
        gstore(dst + 64 * row, v);
    }
}

template <int RADIX, bool REV> int launch_t(const fft_reg_args &a, hipStream_t s)
{
    const uint64_t blocks = (a.batch + 3) / 4;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL((sdsp_fft1024_wave_f32<RADIX, REV>), dim3((uint32_t)blocks), dim3(256), 0, s, reinterpret_cast<float2 *>(a.data),
                       reinterpret_cast<const float2 *>(a.tw), a.batch, a.scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_wave launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace

bool fft_wave_supports(uint32_t n, int radix)
{
    return n == 1024 && (radix == 2 || radix == 4);
}

// a.tw: the plan's register-pass thread-twiddle table (twt_reg)
int launch_fft_wave_f32(const fft_reg_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    if (!fft_wave_supports(a.n, a.radix))
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the one-wave kernels");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a.radix == 2)
        return a.reverse ? launch_t<2, true>(a, s) : launch_t<2, false>(a, s);
    return a.reverse ? launch_t<4, true>(a, s) : launch_t<4, false>(a, s);
}
} // namespace sdsp_hip

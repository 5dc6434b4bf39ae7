// fft_big.hip -- batched N = 8192 / 16384 / 32768 complex f32 FFT with radix-2 butterfly stages
// (sdsp::fft_radix2, fft.h:258-299) in ONE pass over HBM, for gfx950.
//
// These sizes are too large for the register-pass family's "whole tile in LDS" scheme to keep more
// than one workgroup on a CU (N = 16384 is 128 KiB of complex f32), so nothing overlapped the load and
// store phases there (44-51 % of HBM peak; this kernel: 67 % at 8192, 60 % at 16384, 41 % at 32768 --
// where the previous path was the two-pass four-step).  In-place read+write traffic plateaus at 74-77 % of
// 8 TB/s on this part whatever the shape (tools/delaybench.hip), so what is left here is on-chip work.  Here the transform lives in REGISTERS, 32 points per
// thread (N/32 threads per transform, one transform per workgroup), and LDS is only the exchange medium
// between register passes -- moved one plane (real, then imaginary) at a time, so a transform needs
// 4*N bytes of LDS and TWO workgroups of N = 16384 (four of N = 8192) share a CU:
//
//   load     x[k] = data[t + T*k]            T = N/32; a wave reads 512 contiguous bytes per instruction
//   pass A   five DIF stages, strides N/2 .. N/32      thread twiddles W_N^(t << s)
//   exchange position k*M + t  ->  blk*M + v + (j << R)            M = N/32, R = log2(N) - 10
//   pass B   five DIF stages inside the 32 blocks of M points     thread twiddles W_N^(32 v << s)
//   exchange position blk*M + v + (j << R)  ->  32*w + i,  w = bit_reverse(t)
//   pass C   the last R stages (3, 4 or 5) on 32 contiguous positions, constants only
//   store    X[t + T*bit_reverse5(i)] = x[i]           coalesced: the bit reversal (fft.h:269-273) is
//                                                      folded into pass C's choice of w
//
// Stage twiddles come from a per-plan thread-twiddle table ([pass][stage][thread]: the values W_N^(t << s)
// and W_N^(32 v << s) of the row W_N^j, laid out for coalesced loads) when the stage starts and are combined
// with compile-time W_32 constants (fft32.h).  LDS positions are XOR-swizzled
// (sw() below) so that all three access patterns are bank-conflict free.
#include <hip/hip_runtime.h>

#include "fft32.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
using namespace fft32;

// ds_read_b32 / ds_write_b32 are serviced in two groups of 32 lanes, bank = position mod 32.  Pattern B (a
// half-wave = 2^(5-R) blocks x 2^R consecutive v) needs the low block bits p[5+R .. 9] in bank bits R .. 4;
// pattern C (a half-wave = the top five bits of w = the five block bits p[5+R .. L-1]) needs those five bits
// to reach all five bank bits: the remaining p[10 .. L-1] go to bank bits 0 .. R-1.  The XOR term depends only
// on p >> (5+R), so 2^(5+R) consecutive positions stay consecutive (pattern A is untouched).
// (A first version spread 64 lanes over 64 banks -- the wrong model for b32 accesses -- and left pattern C with
// 2-way conflicts: 69.2 / 60.9 / 41.2 % where this one gives 70.3 / 61.5 / 43.1 % at N = 8192 / 16384 / 32768.)
template <int L> __device__ __forceinline__ uint32_t sw(uint32_t p)
{
    constexpr int R = L - 10;
    return p ^ ((((p >> (5 + R)) & ((1u << (5 - R)) - 1)) << R) | ((p >> 10) & ((1u << R) - 1)));
}

// G transforms per workgroup (consecutive in memory), each on its own N/32 threads and LDS plane.  G = 1 is what
// ships: at N = 4096 larger
// workgroups measured slower -- 69.9 % (G = 1), 66.9 % (G = 2), 63.2 % (G = 4) -- the barriers span more waves.
template <int L, bool REV, bool NT, int G = 1>
// four waves per SIMD for every size: at N = 8192 that costs 20-36 B/lane of scratch, but three workgroups
// per CU without scratch measured 61 % against 67 %
__global__ __launch_bounds__(G * (1 << L) / 32, 4) void sdsp_fft_big_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                                           float scale, uint64_t batch)
{
    constexpr int R = L - 10;
    constexpr uint32_t N = 1u << L, T = N / 32, M = N / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft_big_smem[];
    const uint32_t g = threadIdx.x / T;
    float *plane = reinterpret_cast<float *>(sdsp_fft_big_smem) + g * N; // N floats per transform

    const uint32_t t = threadIdx.x % T;
    const uint64_t xform = static_cast<uint64_t>(blockIdx.x) * G + g;
    const bool live = xform < batch; // ragged last workgroup: idle threads still meet the barriers
    float2 *base = data + (live ? xform : 0) * N;
    const uint32_t toff = t * 8u;

    float2 x[32];
    if (G == 1 || live) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k] = NT ? nt_load(at(base + T * k, toff)) : *at(base + T * k, toff);
    }

    fft32_dif<REV, true, 0, true>(x, tw + t, T); // tw: thread-twiddle table [pass][stage][thread], see capi.hip

    const uint32_t blk = t >> R, v = t & ((1u << R) - 1);
    const uint32_t pb = blk * M + v; // pass-B position of register j: pb + (j << R)
    // ---- exchange A -> B, one plane at a time
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t ta = t, tb = pb;
        asm volatile("" : "+v"(ta), "+v"(tb)); // keep the 64 swizzled addresses out of long-lived registers
#pragma unroll
        for (int k = 0; k < 32; k++)
            plane[sw<L>(k * M + ta)] = half ? x[k].y : x[k].x;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const float f = plane[sw<L>(tb + (j << R))];
            if (half)
                x[j].y = f;
            else
                x[j].x = f;
        }
        __syncthreads();
    }

    fft32_dif<REV, true, 0, true>(x, tw + 5 * T + t, T);

    // ---- exchange B -> C
    const uint32_t w = __brev(t) >> (32 - (L - 5));
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t tb = pb, tc = 32u * w;
        asm volatile("" : "+v"(tb), "+v"(tc));
#pragma unroll
        for (int j = 0; j < 32; j++)
            plane[sw<L>(tb + (j << R))] = half ? x[j].y : x[j].x;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const float f = plane[sw<L>(tc + i)];
            if (half)
                x[i].y = f;
            else
                x[i].x = f;
        }
        if (half == 0)
            __syncthreads();
    }

    fft32_dif<REV, false, 5 - R>(x, tw, 0);

    // ---- store: position 32w + i holds X[bit_reverse_L(32w + i)] = X[t + T * bit_reverse5(i)]
    if (G > 1 && !live)
        return;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        float2 o = x[i];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            o.x *= scale;
            o.y *= scale;
        }
        float2 *dst = at(base + T * (int)(__brev((uint32_t)i) >> 27), toff);
        if constexpr (NT)
            nt_store(dst, o);
        else
            *dst = o;
    }
}

template <int L, bool REV, bool NT, int G = 1> int launch_l(const fft_reg_args &a, hipStream_t s)
{
    constexpr size_t lds = (sizeof(float) << L) * G;
    auto kern = sdsp_fft_big_kernel<L, REV, NT, G>;
    if constexpr (lds > 64 * 1024) {
        static std::atomic<uint64_t> attr_done{ 0 };
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done))
            return rc;
    }
    const uint64_t blocks = (a.batch + G - 1) / G;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(G * (1u << L) / 32), lds, s, reinterpret_cast<float2 *>(a.data),
                       reinterpret_cast<const float2 *>(a.tw), a.scale, a.batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_big launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

template <int L> int launch_dir(const fft_reg_args &a, hipStream_t s)
{
    if (a.nontemporal)
        return a.reverse ? launch_l<L, true, true>(a, s) : launch_l<L, false, true>(a, s);
    return a.reverse ? launch_l<L, true, false>(a, s) : launch_l<L, false, false>(a, s);
}
} // namespace

// Radix-4 plans of N = 16384 run here too: a radix-4 DIF stage (fft.h:311-349) is two fused radix-2 DIF
// stages (the inner twiddle is -+i), so the 5 + 5 + 4 radix-2 stages of this kernel are the same dataflow as
// the reference's seven radix-4 stages run unfused; the natural-order result is the same DFT to rounding
// (tests hold it to the same 1e-6 against the oracle's radix-4 algorithm).  The register-pass family's
// genuine two-stage radix-4 passes remain as variants 1 / 2 of such plans.
bool fft_big_supports(uint32_t n, int radix)
{
    if (radix == 4)
        return n == 16384;
    return radix == 2 && (n == 8192 || n == 16384 || n == 32768);
}

int launch_fft_big_f32(const fft_reg_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (a.n) {
    case 8192: return launch_dir<13>(a, s);
    case 16384: return launch_dir<14>(a, s);
    case 32768: return launch_dir<15>(a, s);
    default: break;
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the large single-pass kernels");
}
} // namespace sdsp_hip

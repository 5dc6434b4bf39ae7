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

// sw's XOR term for any position whose 32-block index (position >> (5 + R)) is k: k's five bits rotated by R
template <int R> __device__ __forceinline__ constexpr uint32_t rot5(uint32_t k)
{
    return ((k & ((1u << (5 - R)) - 1)) << R) | ((k >> (5 - R)) & ((1u << R) - 1));
}

// G transforms per workgroup (consecutive in memory), each on its own N/32 threads and LDS plane.  G = 1 is what
// ships: at N = 4096 larger
// workgroups measured slower -- 69.9 % (G = 1), 66.9 % (G = 2), 63.2 % (G = 4) -- the barriers span more waves.
template <int L, bool REV, bool NT, int G = 1, int LAB = 0, bool PERSIST = false>
// four waves per SIMD for every size: at N = 8192 that costs 20-36 B/lane of scratch, but three workgroups
// per CU without scratch measured 61 % against 67 %
__global__ __launch_bounds__(G * (1 << L) / 32, 4) void sdsp_fft_big_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                                           float scale, uint64_t batch, uint32_t stag_first,
                                                                           uint32_t stag_n, uint32_t stag_ticks)
{
    static_assert(G == 1, "one transform per workgroup (larger workgroups measured slower; the row buffer needs a uniform base)");
    constexpr int R = L - 10;
    constexpr uint32_t N = 1u << L, T = N / 32, M = N / 32;
    if (stag_n && blockIdx.x < stag_first) { // LAB: staggered start of the first round of workgroups
        const uint64_t t0 = wall_clock64();
        const uint64_t want = (uint64_t)((blockIdx.x >> 3) % stag_n) * stag_ticks;
        while (wall_clock64() - t0 < want)
            __builtin_amdgcn_s_sleep(16);
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft_big_smem[];
    const uint32_t g = G == 1 ? 0u : threadIdx.x / T;
    const uint32_t plane_off = g * N * 4u; // N floats per transform
    auto lds_f32 = [&](uint32_t byte) -> float & { return *reinterpret_cast<float *>(sdsp_fft_big_smem + byte); };

    const uint32_t t = G == 1 ? threadIdx.x : threadIdx.x % T;
    const uint32_t toff = t * 8u;
    // PERSIST (G == 1): the workgroup walks transforms blockIdx.x, + gridDim.x, ...; the loads of the next transform are issued
    // right behind the stores of this one, so the two memory phases of a CU that holds ONE workgroup overlap
    for (uint64_t xform = static_cast<uint64_t>(blockIdx.x) * G + g; PERSIST ? xform < batch : true; xform += gridDim.x) {
    const bool live = xform < batch; // ragged last workgroup: idle threads still meet the barriers
    const __amdgpu_buffer_rsrc_t rows = make_rows(data + (live ? xform : 0) * N, N * sizeof(float2)); // fft32.h: why a buffer

    float2 x[32];
    if constexpr (LAB == 1) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k] = float2{ (float)(t + k) * scale, (float)(t ^ k) * scale };
    } else if (G == 1 || live) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k] = row_load<NT>(rows, toff, T * k * sizeof(float2));
    }
    if constexpr (LAB != 2) {
    fft32_dif<REV, true, 0, true>(x, tw + t, T); // tw: thread-twiddle table [pass][stage][thread], see capi.hip

    // LDS byte addresses of the three access patterns.  sw<L>() only ever XORs a 5-bit term into the low five bits of a
    // position, and in every pattern that term is a compile-time constant or a per-thread constant, so an access costs
    // at most ONE v_xor (the generic expression cost 4-6 integer operations per access: 1100-1550 of the kernel's
    // 3000-3350 vector instructions were address arithmetic).  With rot(k) = sw's term for a position in 32-block k:
    //   pattern A  position k*M + t          ->  4*k*M + (4t ^ 4*rot(k))                 rot(k) is a literal
    //   pattern B  position pb + (j << R)    ->  baseB[j mod 2^(5-R)] + 128*(j >> (5-R))  no arithmetic per access
    //   pattern C  position 32*w + i         ->  (128*w + 4*xc) ^ 4*i                     xc per thread
    // (checked against sw<L> for every thread and register of the three sizes: tools/model_fft_big_lds.py)
    const uint32_t blk = t >> R, v = t & ((1u << R) - 1);
    constexpr int JL = 1 << (5 - R); // pattern B: the low 5 - R bits of j meet the thread's XOR term
    const uint32_t xb = rot5<R>(blk);
    uint32_t base_b[JL];
#pragma unroll
    for (int jl = 0; jl < JL; jl++)
        base_b[jl] = plane_off + 4u * (blk * M + (v ^ (xb & ((1u << R) - 1))) + (((uint32_t)jl ^ (xb >> R)) << R));
    const uint32_t w = __brev(t) >> (32 - (L - 5));
    const uint32_t base_c = plane_off + ((128u * w) | (4u * ((((w >> R) & ((1u << (5 - R)) - 1)) << R) | ((w >> 5) & ((1u << R) - 1)))));
    const uint32_t base_a = plane_off + 4u * t;
    // ---- exchange A -> B, one plane at a time
    if constexpr (PERSIST)
        __syncthreads(); // the previous transform's last plane has been read by every wave
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t ta = base_a;
        asm volatile("" : "+v"(ta)); // keep the 32 addresses of a plane out of long-lived registers
#pragma unroll
        for (int k = 0; k < 32; k++)
            lds_f32(4u * k * M + (ta ^ (4u * rot5<R>(k)))) = half ? x[k].y : x[k].x;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const float f = lds_f32(base_b[j % JL] + 128u * (j / JL));
            if (half)
                x[j].y = f;
            else
                x[j].x = f;
        }
        __syncthreads();
    }

    fft32_dif<REV, true, 0, true>(x, tw + 5 * T + t, T);

    // ---- exchange B -> C
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t tc = base_c;
        asm volatile("" : "+v"(tc));
#pragma unroll
        for (int j = 0; j < 32; j++)
            lds_f32(base_b[j % JL] + 128u * (j / JL)) = half ? x[j].y : x[j].x;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const float f = lds_f32(tc ^ (4u * i));
            if (half)
                x[i].y = f;
            else
                x[i].x = f;
        }
        if (half == 0)
            __syncthreads();
    }

    fft32_dif<REV, false, 5 - R>(x, tw, 0);
    } // LAB != 2
    if constexpr (LAB == 1) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 32; i++)
            acc += x[i].x * x[i].y;
        if (acc != 12345.678f) {
            if constexpr (PERSIST)
                continue;
            else
                return;
        }
    }

    // ---- store: position 32w + i holds X[bit_reverse_L(32w + i)] = X[t + T * bit_reverse5(i)]
    if constexpr (G > 1) {
        if (!live)
            return;
    }
#pragma unroll
    for (int i = 0; i < 32; i++) {
        float2 o = x[i];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            o.x *= scale;
            o.y *= scale;
        }
        row_store<NT>(rows, toff, T * (__brev((uint32_t)i) >> 27) * sizeof(float2), o);
    }
    if constexpr (!PERSIST)
        break;
    } // transforms of this workgroup
}

struct lab_knobs {
    int lab = 0;
    uint32_t first = 0, n = 0, ticks = 0;
    int persist = -1, per_cu = 0;
};
inline lab_knobs read_lab()
{
    lab_knobs k;
    if (const char *e = getenv("SDSP_LAB_BIG"))
        sscanf(e, "%d,%u,%u,%u,%d,%d", &k.lab, &k.first, &k.n, &k.ticks, &k.persist, &k.per_cu);
    return k;
}

int cu_count()
{
    static std::atomic<int> cached{ 0 };
    int g = cached.load();
    if (g)
        return g;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return 256;
    g = prop.multiProcessorCount;
    cached.store(g);
    return g;
}

template <int L, bool REV, bool NT, int G = 1, int LAB = 0, bool PERSIST = false> int launch_l(const fft_reg_args &a, hipStream_t s)
{
    constexpr size_t lds = (sizeof(float) << L) * G;
    auto kern = sdsp_fft_big_kernel<L, REV, NT, G, LAB, PERSIST>;
    const lab_knobs kn = read_lab();
    if constexpr (lds > 64 * 1024) {
        static std::atomic<uint64_t> attr_done{ 0 };
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done))
            return rc;
    }
    uint64_t blocks = (a.batch + G - 1) / G;
    if constexpr (PERSIST) {
        constexpr int kPerCu = L == 15 ? 1 : L == 14 ? 2 : 4;
        const uint64_t resident = (uint64_t)cu_count() * (kn.per_cu > 0 ? kn.per_cu : kPerCu);
        if (blocks > resident)
            blocks = resident;
    }
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(G * (1u << L) / 32), lds, s, reinterpret_cast<float2 *>(a.data),
                       reinterpret_cast<const float2 *>(a.tw), a.scale, a.batch, kn.first, kn.n, kn.ticks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_big launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

template <int L> int launch_dir(const fft_reg_args &a, hipStream_t s)
{
    if constexpr (L >= 14) {
        const lab_knobs kn = read_lab();
        if (kn.lab == 1)
            return kn.persist == 1 ? launch_l<L, true, true, 1, 1, true>(a, s) : launch_l<L, true, true, 1, 1>(a, s);
        if (kn.lab == 2)
            return kn.persist == 1 ? launch_l<L, true, true, 1, 2, true>(a, s) : launch_l<L, true, true, 1, 2>(a, s);
        if (kn.persist == 1 && a.nontemporal)
            return a.reverse ? launch_l<L, true, true, 1, 0, true>(a, s) : launch_l<L, false, true, 1, 0, true>(a, s);
    }
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

// fft_big.hip -- the registers-resident single-pass FFT kernel for gfx950, f32:
//   * batched N = 8192 / 16384 / 32768 complex transforms with radix-2 butterfly stages (sdsp::fft_radix2, fft.h:258-299), and
//     N = 16384 = 4^7 with seven radix-4 stages (sdsp::fft_radix4, fft.h:301-360; fft32_r4.h);
//   * the fused fast convolution of SURVEY 8(f)-1 (CONV) and the real-input packing of SURVEY 8(f)-3 (REAL) for N = 2048 .. 32768.
//
// These sizes are too large for the register-pass family's "whole tile in LDS" scheme to keep more
// than one workgroup on a CU (N = 16384 is 128 KiB of complex f32), so nothing overlapped the load and
// store phases there (44-51 % of HBM peak; this kernel: 71-77 % at 8192, 69-71 % at 16384, 56 % at 32768 --
// where the previous path was the two-pass four-step).  In-place read+write traffic plateaus at 74-77 % of
// 8 TB/s on this part whatever the shape (tools/delaybench.hip), and a CU that holds two workgroups of 128 KiB / one of
// 256 KiB caps at ~70 % / ~65 % even with no arithmetic between the loads and the stores (tools/cucap.hip).
// Here the transform lives in REGISTERS, 32 points per
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
#include "fft32_r4.h"
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
template <int L> [[maybe_unused]] __device__ __forceinline__ uint32_t sw(uint32_t p) // the layout's definition; the kernel uses the forms below
{
    constexpr int R = L - 10;
    return p ^ ((((p >> (5 + R)) & ((1u << (5 - R)) - 1)) << R) | ((p >> 10) & ((1u << R) - 1)));
}

// sw's XOR term for any position whose 32-block index (position >> (5 + R)) is k: k's five bits rotated by R
template <int R> __device__ __forceinline__ constexpr uint32_t rot5(uint32_t k)
{
    return ((k & ((1u << (5 - R)) - 1)) << R) | ((k >> (5 - R)) & ((1u << R) - 1));
}

// One transform per workgroup (two / four consecutive transforms on a larger workgroup measured slower at N = 4096:
// 69.9 % (1), 66.9 % (2), 63.2 % (4) -- the barriers span more waves).
// Four waves per SIMD for every size (<= 128 VGPRs: 124 / 124 / 127 at N = 8192 / 16384 / 32768, no scratch).
// R4 (N = 16384 = 4^7; N = 4096 = 4^6 for real-input plans): the layers run as radix-4 DIF stages (fft32_r4.h) -- two stages, then half
// of the third in pass A; its other half and two stages in pass B; two stages (N = 4096: one) in pass C -- with their own thread-twiddle table
// (capi.hip: upload_thread_twiddles_big_r4); loads, exchanges and the store are the same.
// CONV (forward plans): the fused fast convolution data <- IFFT(FFT(data) .* h) of SURVEY 8(f)-1.  The forward
// transform leaves register i holding X[t + T * bit_reverse5(i)]; multiplied by h there (rows of h through a buffer resource,
// default cache policy: every workgroup reads the same N points) and RENAMED z[bit_reverse5(i)] = x[i], that is the input
// layout of pass A, so the reverse transform (conjugated table values, +i rotations, 1/N at the store) runs on the same
// registers and the same LDS plane: one HBM read and one write per element instead of three of each.
// REAL: real-input packing, SURVEY 8(f)-3 -- the buffer holds 2N reals per transform, read as N complex.
// 1 (forward plans): split after the transform, X[k] from Z[k] and Z[N-k]; 2 (reverse plans): merge before it; h = W_2N^j,
// direction-folded (the formulas are fft_reg.hip's MODE 1 / 2).  Element k = t + T b and its partner N - k = (T - t) + T (31 - b)
// sit in different threads, so the pairs meet in LDS: the elements with b < 16 are parked as float2 in slot k, the owner of
// N - k (b >= 16) reads its partner there, computes BOTH results of the pair, keeps its own and puts the other back into the
// same slot, and the parked side reads its results back -- 64 ds_*_b64 per thread and no second copy of the data in registers.
// k = 0 and k = N/2 (thread 0) pair with nobody.
template <int L, bool REV, bool NT, bool R4 = false, bool CONV = false, int REAL = 0, bool PF = true>
__global__ __launch_bounds__((1 << L) / 32, 4) void sdsp_fft_big_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                                       float scale, uint64_t batch, const float2 *__restrict__ h)
{
    static_assert(!CONV || !REV, "the fused convolution belongs to forward plans");
    static_assert(REAL == 0 || (!CONV && (REAL == 1) == !REV), "real-input packing: split forward, merge reverse");
    static_assert(!R4 || L == 14 || L == 12, "radix-4 stages: N = 16384 (and N = 4096 for the real-input form)");
    constexpr int R = L - 10;
    constexpr uint32_t N = 1u << L, T = N / 32, M = N / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft_big_smem[]; // N floats: one plane of the transform
    auto lds_f32 = [&](uint32_t byte) -> float & { return *reinterpret_cast<float *>(sdsp_fft_big_smem + byte); };

    const uint32_t t = threadIdx.x;
    const uint32_t toff = t * 8u;
    const uint64_t xform = blockIdx.x;
    if (xform >= batch)
        return;
    // the transform's rows (T elements = 2 / 4 / 8 KiB apart) through a buffer resource: fft32.h, make_rows
    const __amdgpu_buffer_rsrc_t rows = make_rows(data + xform * N, N * sizeof(float2));

    float2 wa[5]; // pass A's thread twiddles (radix-2 form), in flight with the data
    if constexpr (!R4 && PF) {
#pragma unroll
        for (int st = 0; st < 5; st++)
            wa[st] = tw[st * T + t];
    }
    float2 x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = row_load<NT>(rows, toff, T * k * sizeof(float2));

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
        base_b[jl] = 4u * (blk * M + (v ^ (xb & ((1u << R) - 1))) + (((uint32_t)jl ^ (xb >> R)) << R));
    const uint32_t w = __brev(t) >> (32 - (L - 5));
    const uint32_t base_c = (128u * w) | (4u * ((((w >> R) & ((1u << (5 - R)) - 1)) << R) | ((w >> 5) & ((1u << R) - 1))));
    const uint32_t base_a = 4u * t;

    // the transform on the registers: y[k] = element t + T k  ->  y[i] = result[t + T * bit_reverse5(i)].  RV: its direction;
    // CJ: the table holds the other direction's thread twiddles; AGAIN: the LDS plane may still be read by the transform before
    auto transform = [&](float2 (&y)[32], const float2 (&wa)[5], auto rev_tag, auto conj_tag, auto again_tag) {
        constexpr bool RV = decltype(rev_tag)::value, CJ = decltype(conj_tag)::value, AGAIN = decltype(again_tag)::value;
        [[maybe_unused]] auto tab = [&](int slot) { // radix-4 form: the thread's value of a table slot
            float2 wv = tw[slot * T + t];
            if constexpr (CJ)
                wv.y = -wv.y;
            return wv;
        };
        if constexpr (R4) {
            float2 thr[3];
#pragma unroll
            for (int q = 0; q < 3; q++)
                thr[q] = tab(q); // W_N^((q + 1) t)
            r4_stage<RV, 4, 7, 2, true>(y, thr); // stage 0: quarter = register bits 4, 3; constant W_32^(q (k & 7))
#pragma unroll
            for (int q = 0; q < 3; q++)
                thr[q] = tab(3 + q); // W_4096^((q + 1) t)
            r4_stage<RV, 2, 1, 8, true>(y, thr); // stage 1: register bits 2, 1; constant W_8^(q (k & 1))
            // stage 2, first layer: register bit 0 is index bit L - 5, index bit L - 6 is the thread's top bit: the quarter (1, 1) =
            // odd registers of the upper half of the threads
            layer<1>(y);
            const bool upper = t >= T / 2;
#pragma unroll
            for (int k = 1; k < 32; k += 2) {
                const float2 r = rot_i<RV>(y[k]);
                y[k] = float2{ upper ? r.x : y[k].x, upper ? r.y : y[k].y };
            }
        } else {
            if constexpr (PF)
                fft32_dif_w<RV>(y, wa); // wa: pass A's thread twiddles, fetched by the caller under the data loads
            else
                fft32_dif<RV, true, 0, true, CJ>(y, tw + t, T);
        }

        // pass B's thread twiddles are fetched here, under the exchange: behind its barriers the first stage of the pass would
        // wait for an L2 round trip
        float2 wb[5];
        if constexpr (!R4 && PF) {
#pragma unroll
            for (int st = 0; st < 5; st++) {
                wb[st] = tw[(5 + st) * T + t]; // tw: thread-twiddle table [pass][stage][thread], see capi.hip
                if constexpr (CJ)
                    wb[st].y = -wb[st].y;
            }
        }

        // ---- exchange A -> B, one plane at a time
        if constexpr (AGAIN)
            __syncthreads(); // every wave has read the previous transform's last plane
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t ta = base_a;
            asm volatile("" : "+v"(ta)); // keep the 32 addresses of a plane out of long-lived registers
#pragma unroll
            for (int k = 0; k < 32; k++)
                lds_f32(4u * k * M + (ta ^ (4u * rot5<R>(k)))) = half ? y[k].y : y[k].x;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const float f = lds_f32(base_b[j % JL] + 128u * (j / JL));
                if (half)
                    y[j].y = f;
                else
                    y[j].x = f;
            }
            __syncthreads();
        }

        if constexpr (R4) {
            layer<16>(y); // stage 2, second layer: register bit 4 is index bit 8
            r4_split_twiddles<RV>(y, (blk & 1u) != 0, tab(6), tab(7), std::make_integer_sequence<int, 32>{});
            float2 thr[3];
#pragma unroll
            for (int q = 0; q < 3; q++)
                thr[q] = tab(8 + q); // W_256^((q + 1) v)
            r4_stage<RV, 3, 3, 4, true>(y, thr); // stage 3: register bits 3, 2; constant W_16^(q (j & 3))
#pragma unroll
            for (int q = 0; q < 3; q++)
                thr[q] = tab(11 + q); // W_64^((q + 1) v)
            r4_stage<RV, 1, 0, 0, true>(y, thr); // stage 4: register bits 1, 0; thread twiddles only
        } else {
            if constexpr (PF)
                fft32_dif_w<RV>(y, wb);
            else
                fft32_dif<RV, true, 0, true, CJ>(y, tw + 5 * T + t, T);
        }

        // ---- exchange B -> C
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t tc = base_c;
            asm volatile("" : "+v"(tc));
#pragma unroll
            for (int j = 0; j < 32; j++)
                lds_f32(base_b[j % JL] + 128u * (j / JL)) = half ? y[j].y : y[j].x;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const float f = lds_f32(tc ^ (4u * i));
                if (half)
                    y[i].y = f;
                else
                    y[i].x = f;
            }
            if (half == 0)
                __syncthreads();
        }

        if constexpr (R4) {
            if constexpr (R == 4) {
                const float2 none[3] = {};
                r4_stage<RV, 3, 3, 4, false>(y, none); // stage 5: register bits 3, 2; constants W_16^(q (i & 3)) only
            }
            r4_layers<RV, 1>(y); // the last stage (register bits 1, 0): no twiddles
        } else {
            fft32_dif<RV, false, 5 - R>(y, tw, 0);
        }
    };
    using yes = std::true_type;
    using no = std::false_type;

    // real-input packing: register r holds element k = t + T b(r), b(r) = r before the transform, bit_reverse5(r) after it
    [[maybe_unused]] auto real_pairs = [&](float2 (&y)[32], auto merge_tag, auto after_tag) {
        constexpr bool MERGE = decltype(merge_tag)::value, AFTER = decltype(after_tag)::value;
        auto lds_f2 = [&](uint32_t byte) -> float2 & { return *reinterpret_cast<float2 *>(sdsp_fft_big_smem + byte); };
        const __amdgpu_buffer_rsrc_t wrows = make_rows(h, 2 * N * sizeof(float2)); // W_2N^j, j < 2N
        if constexpr (AFTER)
            __syncthreads(); // every wave has read the transform's last plane
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int b = AFTER ? (int)(__brev((uint32_t)r) >> 27) : r;
            if (b < 16)
                lds_f2(8u * t + 8u * T * b) = y[r];
        }
        __syncthreads();
        const bool t0 = t == 0;
        const uint32_t pbase = 8u * (T - t); // partner of (t, b): slot (T - t) + T (31 - b) = N - k
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int b = AFTER ? (int)(__brev((uint32_t)r) >> 27) : r;
            if (b < 16)
                continue;
            const float2 pa = lds_f2(pbase + 8u * T * (31 - b));                 // element a = N - k
            const float2 w = row_load<false>(wrows, pbase, 8u * T * (31 - b));   // W_2N^a
            const float2 own = y[r];
            const float2 e = float2{ 0.5f * (pa.x + own.x), 0.5f * (pa.y - own.y) }; // (A + conj B) / 2
            const float2 d = float2{ 0.5f * (pa.x - own.x), 0.5f * (pa.y + own.y) }; // (A - conj B) / 2
            const float2 wd = cmul(d, w);
            float2 ra, rb;
            if constexpr (MERGE) { // Z[a] = E + i O, Z[N-a] = conj(E - i O)
                ra = float2{ e.x - wd.y, e.y + wd.x };
                rb = float2{ e.x + wd.y, wd.x - e.y };
            } else { // X[a] = E + T, X[N-a] = conj(E - T), T = -i W D
                ra = float2{ e.x + wd.y, e.y - wd.x };
                rb = float2{ e.x - wd.y, -wd.x - e.y };
            }
            if (b == 16) { // k = N/2 in thread 0: conj, no partner (the slot it touched is the padding behind the plane)
                rb.x = t0 ? own.x : rb.x;
                rb.y = t0 ? -own.y : rb.y;
            }
            y[r] = rb;
            lds_f2(pbase + 8u * T * (31 - b)) = ra;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int b = AFTER ? (int)(__brev((uint32_t)r) >> 27) : r;
            if (b >= 16)
                continue;
            float2 v = lds_f2(8u * t + 8u * T * b);
            if (b == 0) { // k = 0 in thread 0: (X[0], X[N]) packed, both real
                const float2 z = v;
                const float2 p0 = MERGE ? float2{ 0.5f * (z.x + z.y), 0.5f * (z.x - z.y) } : float2{ z.x + z.y, z.x - z.y };
                v.x = t0 ? p0.x : v.x;
                v.y = t0 ? p0.y : v.y;
            }
            y[r] = v;
        }
    };

    if constexpr (REAL == 2)
        real_pairs(x, yes{}, no{});
    transform(x, wa, std::integral_constant<bool, REV>{}, no{}, std::integral_constant<bool, REAL == 2>{});
    if constexpr (REAL == 1)
        real_pairs(x, no{}, yes{});

    if constexpr (CONV) {
        // x[i] = X[t + T * bit_reverse5(i)]: times h there, renamed into pass A's input order (no data moves: a renaming)
        const __amdgpu_buffer_rsrc_t hrows = make_rows(h, N * sizeof(float2));
        float2 z[32];
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int k = (int)(__brev((uint32_t)i) >> 27);
            z[k] = cmul(x[i], row_load<false>(hrows, toff, T * k * sizeof(float2)));
        }
        float2 wc[5]; // the reverse transform's pass A: the conjugates
#pragma unroll
        for (int st = 0; st < 5; st++)
            wc[st] = float2{ wa[st].x, -wa[st].y };
        transform(z, wc, yes{}, yes{}, yes{});
#pragma unroll
        for (int i = 0; i < 32; i++)
            x[i] = z[i];
    }

    // ---- store: position 32w + i holds X[bit_reverse_L(32w + i)] = X[t + T * bit_reverse5(i)]
#pragma unroll
    for (int i = 0; i < 32; i++) {
        float2 o = x[i];
        if constexpr (REV || CONV) { // reverse_fft::ScaleValues, fft.h:128-132
            o.x *= scale;
            o.y *= scale;
        }
        row_store<NT>(rows, toff, T * (__brev((uint32_t)i) >> 27) * sizeof(float2), o);
    }
}

template <int L, bool REV, bool NT, bool R4 = false, bool CONV = false, int REAL = 0, bool PF = true> int launch_l2(const fft_reg_args &a, hipStream_t s)
{
    constexpr size_t lds = (sizeof(float) << L) + (REAL ? 8 : 0); // real-input pairs: one float2 of padding behind the plane
    auto kern = sdsp_fft_big_kernel<L, REV, NT, R4, CONV, REAL, PF>;
    if constexpr (lds > 64 * 1024) {
        static std::atomic<uint64_t> attr_done{ 0 };
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done))
            return rc;
    }
    if (a.batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((uint32_t)a.batch), dim3((1u << L) / 32), lds, s, reinterpret_cast<float2 *>(a.data),
                       reinterpret_cast<const float2 *>(a.tw), a.scale, a.batch, reinterpret_cast<const float2 *>(a.tw2));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_big launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

// PF (the kernel's last template argument): fetch a pass's thread twiddles ahead of it -- pass A's under the data loads, pass
// B's under the first exchange -- instead of at each stage's start behind the exchange's barriers.  One-process A/B
// (profiles/r02_fft_big_lab.md, section 7): complex N = 8192 +1.2 points, N = 16384 +2.3; N = 32768 -1.2 and the
// real-input forms -0.4 .. -1.6 (their registers are full: the ten extra live values cost more than the round trip) -- so it is
// on for the plain radix-2 transforms up to N = 16384 only.
template <int L, bool REV, bool NT, bool R4 = false, bool CONV = false, int REAL = 0> int launch_l(const fft_reg_args &a, hipStream_t s)
{
    constexpr bool PF = !R4 && !CONV && REAL == 0 && L <= 14;
    return launch_l2<L, REV, NT, R4, CONV, REAL, PF>(a, s);
}

template <int L> int launch_dir(const fft_reg_args &a, hipStream_t s)
{
    if (a.real_mode == 1 || a.real_mode == 2) { // real-input packing: a.tw2 = W_2N^j
        if (!a.tw2 || (a.real_mode == 1) != !a.reverse)
            return fail(SDSP_HIP_ERR_INVALID_ARG, "fft_big real-input packing: split forward / merge reverse, W_2N needed");
        if constexpr (L == 14 || L == 12) {
            if (a.radix == 4) // a.tw: the radix-4 table
                return a.reverse ? launch_l<L, true, true, true, false, 2>(a, s) : launch_l<L, false, true, true, false, 1>(a, s);
        }
        if (a.radix != 2)
            return fail(SDSP_HIP_ERR_INVALID_ARG, "fft_big real-input packing: radix-2 stages (radix-4 stages at N = 16384)");
        return a.reverse ? launch_l<L, true, true, false, false, 2>(a, s) : launch_l<L, false, true, false, false, 1>(a, s);
    }
    if (a.real_mode == 3) { // fused convolution (forward radix-2 plans): a.tw2 = h
        if (a.reverse || !a.tw2)
            return fail(SDSP_HIP_ERR_INVALID_ARG, "fft_big convolution: forward plan and a filter spectrum needed");
        if constexpr (L == 14) {
            if (a.radix == 4) // a.tw: the radix-4 table
                return launch_l<L, false, true, true, true>(a, s);
        }
        if (a.radix != 2)
            return fail(SDSP_HIP_ERR_INVALID_ARG, "fft_big convolution: radix-2 stages (radix-4 stages at N = 16384)");
        return launch_l<L, false, true, false, true>(a, s);
    }
    if constexpr (L == 14) {
        if (a.radix == 4) // a.tw: the radix-4 table
            return a.reverse ? launch_l<L, true, true, true>(a, s) : launch_l<L, false, true, true>(a, s);
    }
    if (a.nontemporal)
        return a.reverse ? launch_l<L, true, true>(a, s) : launch_l<L, false, true>(a, s);
    return a.reverse ? launch_l<L, true, false>(a, s) : launch_l<L, false, false>(a, s);
}
} // namespace

// Radix-4 plans of N = 16384 run here too, as seven genuine radix-4 DIF stages (the R4 form of the kernel, fft32_r4.h)
// on their own thread-twiddle table.
bool fft_big_supports(uint32_t n, int radix)
{
    if (radix == 4)
        return n == 16384;
    return radix == 2 && (n == 8192 || n == 16384 || n == 32768);
}

// the fused convolution and real-input plans (n = n_real / 2): radix-2 stages n = 2048 .. 32768, radix-4 stages n = 16384
bool fft_big_conv_supports(uint32_t n, int radix)
{
    return (radix == 2 && (n == 2048 || n == 4096 || n == 8192 || n == 16384 || n == 32768)) || (radix == 4 && n == 16384);
}
// (N = 4096 radix 4: the radix-4 form exists for the real-input plans only -- the complex transform and its convolution have
// the tuned kernels of fft4096.hip)
bool fft_big_real_supports(uint32_t n, int radix) { return fft_big_conv_supports(n, radix) || (radix == 4 && n == 4096); }

int launch_fft_big_f32(const fft_reg_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (a.n) {
    case 2048: // real-input plans (n_real = 4096) and the fused convolution only: the complex transform has its own kernels
        if (a.real_mode == 0)
            break;
        return launch_dir<11>(a, s);
    case 4096: // likewise (n_real = 8192)
        if (a.real_mode == 0)
            break;
        return launch_dir<12>(a, s);
    case 8192: return launch_dir<13>(a, s);
    case 16384: return launch_dir<14>(a, s);
    case 32768: return launch_dir<15>(a, s);
    default: break;
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the large single-pass kernels");
}
} // namespace sdsp_hip

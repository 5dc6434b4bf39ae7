// fft_wave.hip -- batched N = 1024 complex FFT (f32 and f64), radix-2 (fft.h:258-299) or radix-4 (fft.h:301-360) stages,
// one transform per WAVE, for gfx950.  (N = 1024 in double is the reference's own test point, testFFT.cpp:239 / BASELINE
// config 1.)
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
// LDS slots (units of one complex element) are p ^ X(p >> SH) with X linear over GF(2), chosen by search
// (tools/model_fft_wave.py) so that every access pattern of both radices is conflict-free under the banking of
// MI355X_MICROARCH.md (LDS): f32: ds_read_b64 = 2 x 32 lanes over 64 banks, ds_write_b64 = 4 x 16 contiguous lanes over 32
// banks; f64: ds_read_b128 = 4 x 16 lanes (the guide's lane groups) over 64 banks, ds_write_b128 = 8 x 8 lanes over 32 banks.
// (A first map, built for "32 lanes over 64 banks" for the writes too, measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 25 %.)
#include <hip/hip_runtime.h>

#include "fft32.h"
#include "fft_passes.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
// Lanes of ONE wave exchange data through LDS with no workgroup barrier: the hardware serves a wave's LDS instructions in
// order, so a ds_read issued after a ds_write sees it.  What still has to be said is the compiler's side of it: the
// writes of other lanes must stay in front of this lane's reads (and the reads in front of the next exchange's writes).
// Wavefront-scope fences + the wave barrier state that dependency; they emit no instruction.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

namespace
{
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 gload(const float2 *p)
{
    const v2f_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(p));
    return float2{ v.x, v.y };
}
__device__ __forceinline__ void gstore(float2 *p, float2 a)
{
    __builtin_nontemporal_store(v2f_t{ a.x, a.y }, reinterpret_cast<v2f_t *>(p));
}
__device__ __forceinline__ double2 gload(const double2 *p)
{
    const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(p));
    return double2{ v.x, v.y };
}
__device__ __forceinline__ void gstore(double2 *p, double2 a)
{
    __builtin_nontemporal_store(v2d_t{ a.x, a.y }, reinterpret_cast<v2d_t *>(p));
}

// Real-input packing (fft_reg.hip MODE 1 / 2) inside a wave.  The buffer holds 2N real samples per
// transform, read as N = 64 P complex z[m] = x[2m] + i x[2m+1]; a lane owns the elements k = t + 64 r, r < P, of a
// transform.  The partner element (N - k) mod N of the split / merge lives in lane (64 - t) mod 64, row P - 1 - r -- for
// lane 0 in the lane itself, row (P - r) mod P -- so it arrives by ONE ds_bpermute per word instead of an LDS pass with barriers.
//   SPLIT (forward, after the transform):  X[k] = E + T, E = (Z[k] + conj Z[N-k]) / 2, T = -i W_2N^k (Z[k] - conj Z[N-k]) / 2;
//                                          element 0 is stored as (X[0], X[N]), both real
//   MERGE (inverse, before the transform): Z[k] = E + i conj(W_2N^k) (X[k] - conj X[N-k]) / 2 (tw2 is reverse-folded);
//                                          element 0 = ((X[0] + X[N]) / 2, (X[0] - X[N]) / 2)
// reg_of(r): the register that holds row r (the transform's output order for SPLIT, natural order for MERGE).
template <int P, bool MERGE, typename RegOf>
__device__ __forceinline__ void real_pack_stage(float2 *x, uint32_t t, const float2 *__restrict__ tw2, RegOf reg_of)
{
    const int src = (int)((64u - t) & 63u);
    float2 y[P];
#pragma unroll
    for (int r = 0; r < P; r++) {
        const int j = reg_of(r), jp = reg_of(P - 1 - r), j0 = reg_of((P - r) % P);
        const float2 za = x[j];
        const float2 other = float2{ __shfl(x[jp].x, src), __shfl(x[jp].y, src) };
        const float2 zb = t == 0 ? x[j0] : other;
        const float2 w = tw2[t + 64 * r];
        const float2 e = float2{ 0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y) }; // (za + conj zb) / 2
        const float2 d = float2{ 0.5f * (za.x - zb.x), 0.5f * (za.y + zb.y) }; // (za - conj zb) / 2
        const float2 wd = passes::cmul(d, w);
        float2 v = MERGE ? float2{ e.x - wd.y, e.y + wd.x }  // E + i (conj(W) D)
                         : float2{ e.x + wd.y, e.y - wd.x }; // E - i (W D)
        if (r == 0 && t == 0)
            v = MERGE ? float2{ 0.5f * (za.x + za.y), 0.5f * (za.x - za.y) } : float2{ za.x + za.y, za.x - za.y };
        y[j] = v;
    }
#pragma unroll
    for (int j = 0; j < P; j++)
        x[j] = y[j];
}

// X(h) for h = p >> SH (10 - SH bits): XOR of the rows of the bits set in h.  SH = 5: float2, SH = 4: double2
template <int SH> struct rows;
template <> struct rows<5> {
    static constexpr uint32_t r[6] = { 16, 29, 6, 23, 18, 0 };
};
template <> struct rows<4> {
    static constexpr uint32_t r[6] = { 5, 3, 15, 1, 6, 9 };
};
template <int SH> constexpr uint32_t xterm(uint32_t h)
{
    uint32_t x = 0;
    for (int b = 0; b < 10 - SH; b++)
        if ((h >> b) & 1)
            x ^= rows<SH>::r[b];
    return x;
}
template <int SH> __device__ __forceinline__ uint32_t xterm_dev(uint32_t h)
{
    uint32_t x = 0;
#pragma unroll
    for (int b = 0; b < 10 - SH; b++)
        x ^= ((h >> b) & 1) ? rows<SH>::r[b] : 0u;
    return x;
}

// WAVES: transforms (= waves) per workgroup: 4 in f32 (32 KiB of LDS), 2 in f64 (32 KiB)
// CONV (forward plans): data <- IFFT(FFT(data) .* h) in one kernel (what a reference user writes as
// fft_radix4(x); x[k] *= H[k]; fft_radix4<reverse_fft>(x)): the forward result X[t + 64 reversed(i)] in register i is the
// input layout of the transform's own first pass, so after the per-bin multiply (h: natural order, 8 KiB, from L2) the
// reverse transform runs on the same registers and LDS region with the conjugated thread twiddles; 1/N at the end.
// REAL (f32): 1 = real-input forward (split after the transform), 2 = real-input inverse (merge before it); h = W_2048^k
template <typename C, typename S, int RADIX, bool REV, int WAVES, bool CONV = false, int REAL = 0>
__global__ __launch_bounds__(64 * WAVES) void sdsp_fft1024_wave(C *__restrict__ data, const C *__restrict__ tw, uint64_t batch, S scale,
                                                                const C *__restrict__ h = nullptr)
{
    static_assert(!(CONV && REV), "the fused convolution belongs to forward plans");
    static_assert(REAL == 0 || (sizeof(C) == 8 && !CONV && (REAL == 1) == !REV), "real-input packing: f32, split forward / merge inverse");
    constexpr int SH = sizeof(C) == 8 ? 5 : 4;
    __shared__ __attribute__((aligned(16))) C lds_all[WAVES][1024];
    const uint32_t t = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    C *lds = lds_all[wave];
    const uint64_t f = static_cast<uint64_t>(blockIdx.x) * WAVES + wave;
    if (f >= batch)
        return; // wave-uniform; the kernel has no barrier
    // thread twiddles: [pass][value][thread], 64 threads per transform (fft_reg.hip: twl)
    auto twl = [&](int pass, int v) { return tw[(6 * pass + v) * 64 + t]; };

    C x[16];
    const C *src = data + f * 1024 + t;
#pragma unroll
    for (int k = 0; k < 16; k++)
        x[k] = gload(src + 64 * k);

    // RV: direction of this pass (differs from REV only in the second half of the fused convolution, which runs the reverse
    // transform with the conjugates of the forward thread twiddles)
    auto run_pass = [&](auto pass_tag, auto rev_tag) {
        constexpr int I = decltype(pass_tag)::value;
        constexpr bool RV = decltype(rev_tag)::value;
        auto twc = [&](int v) {
            C w = twl(I, v);
            if constexpr (RV != REV)
                w.y = -w.y;
            return w;
        };
        if constexpr (RADIX == 2) {
            C w[4];
            if constexpr (I < 2) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    w[j] = twc(j);
            }
            passes::r2_pass<RV, (I < 2), (I == 2 ? 2 : 0)>::run(x, w);
        } else {
            C w1[3], w2[3];
            if constexpr (I < 2) {
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    w1[q] = twc(q);
                    w2[q] = twc(q + 3);
                }
            }
            passes::r4_pass<RV, (I < 2), (I < 2)>(x, w1, w2);
        }
    };
    const uint32_t w = RADIX == 2 ? (__brev(t) >> 26) : (((t & 3u) << 4) | (t & 12u) | (t >> 4));
    auto transform = [&](auto rev_tag) {
    run_pass(std::integral_constant<int, 0>{}, rev_tag);

    // ---- exchange A -> B.  A: p = t + 64 k, p >> SH = (t >> SH) | (k << (6 - SH))
    wave_lds_sync(); // (a second transform of the fused convolution: the first one's last reads come first)
    {
        uint32_t ta = t ^ xterm_dev<SH>(t >> SH);
        asm volatile("" : "+v"(ta)); // one v_xor per access instead of sixteen live addresses
#pragma unroll
        for (int k = 0; k < 16; k++)
            lds[64 * k + (ta ^ xterm<SH>((uint32_t)k << (6 - SH)))] = x[k];
    }
    wave_lds_sync();
    // B: p = 64 b + v + 4 j, p >> SH = (b << (6 - SH)) | (j >> (SH - 2))
    uint32_t tb = 64u * (t >> 2) + ((t & 3u) ^ xterm_dev<SH>((t >> 2) << (6 - SH)));
    {
        uint32_t a = tb;
        asm volatile("" : "+v"(a));
#pragma unroll
        for (int j = 0; j < 16; j++)
            x[j] = lds[a ^ ((4u * j) ^ xterm<SH>((uint32_t)j >> (SH - 2)))];
    }

    run_pass(std::integral_constant<int, 1>{}, rev_tag);

    // ---- exchange B -> C (pass B's slots are written by the thread that read them)
    {
        uint32_t a = tb;
        asm volatile("" : "+v"(a));
#pragma unroll
        for (int j = 0; j < 16; j++)
            lds[a ^ ((4u * j) ^ xterm<SH>((uint32_t)j >> (SH - 2)))] = x[j];
    }
    wave_lds_sync();
    // C: p = 16 w + i, p >> SH = w >> (SH - 4); w = reversed(t) so that the outputs land at t + 64 * reversed(i)
    {
        uint32_t a = (16u * w) ^ xterm_dev<SH>(w >> (SH - 4));
        asm volatile("" : "+v"(a));
#pragma unroll
        for (int i = 0; i < 16; i++)
            x[i] = lds[a ^ (uint32_t)i];
    }

    run_pass(std::integral_constant<int, 2>{}, rev_tag);
    };
    if constexpr (REAL == 2)
        real_pack_stage<16, true>(x, t, h, [](int r) { return r; });
    transform(std::integral_constant<bool, REV>{});
    if constexpr (REAL == 1) // register i holds row row(i); both row maps are involutions
        real_pack_stage<16, false>(x, t, h, [](int r) { return RADIX == 2 ? (int)(__brev((uint32_t)r) >> 28) : 4 * (r & 3) + (r >> 2); });

    if constexpr (CONV) {
        // x[i] = X[t + 64 row(i)]: multiply by h there and renumber so that register row(i) is element t + 64 row(i) -- the
        // first pass's input layout
        C z[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int row = RADIX == 2 ? (int)(__brev((uint32_t)i) >> 28) : 4 * (i & 3) + (i >> 2);
            z[row] = passes::cmul(x[i], h[t + 64 * row]);
        }
#pragma unroll
        for (int i = 0; i < 16; i++)
            x[i] = z[i];
        transform(std::integral_constant<bool, true>{});
    }

    C *dst = data + f * 1024 + t;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        C v = x[i];
        if constexpr (REV || CONV) { // reverse_fft::ScaleValues, fft.h:128-132
            v.x *= scale;
            v.y *= scale;
        }
        const int row = RADIX == 2 ? (int)(__brev((uint32_t)i) >> 28) : 4 * (i & 3) + (i >> 2);
        gstore(dst + 64 * row, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same idea for N = 256, 512 and 2048 (radix-2 stages, f32): a wave owns 1024 consecutive points (N <= 512: 4 or 2 whole
// transforms) or one transform of 2048, P = N / 64 points of each transform per lane.  Loads and stores are the N = 1024
// kernel's (register j = element t + 64 j of the wave's block: 512 contiguous bytes per instruction); a transform's log2 N
// stages run as passes of log2 P stages on the lane's P points:
//   pass i < last   positions b (s_i P) + v + s_i k,  s_i = N >> (log2 P (i + 1)),  v = t mod s_i,  b = t / s_i; thread
//                   twiddle of global stage g: W_N^(v << g), from the plan's [stage][lane] table (capi.hip:
//                   upload_thread_twiddles_wave), times compile-time W_P constants
//   last pass       the remaining stages on the P contiguous positions P w + k, w = bit_reverse6(t): outputs X[t + 64 rev(k)]
// with one exchange through the wave's own LDS between passes (in order, no barrier).  Slot map: p ^ X(p >> SH), X linear
// over GF(2), rows found by search per size (tools/model_fft_wave.py): reads and writes conflict-free under the banking above.
// SH = 5 except at N = 256, where no map of p >> 5 exists and bit 4 joins the inputs (its row touches bits 0..3 only, so the
// map stays a bijection).
template <int L> struct rows2;
template <> struct rows2<8> {
    static constexpr int SH = 4;
    static constexpr uint32_t r[6] = { 4, 9, 17, 2, 0, 0 }; // conflict-free for the radix-2 and the radix-4 last-pass blocks
};
template <> struct rows2<9> {
    static constexpr int SH = 5;
    static constexpr uint32_t r[6] = { 10, 28, 15, 18, 0, 0 };
};
template <> struct rows2<11> {
    static constexpr int SH = 5;
    static constexpr uint32_t r[6] = { 25, 23, 13, 5, 7, 30 };
};
template <int L> constexpr uint32_t xterm2(uint32_t p) // X(p >> SH)
{
    uint32_t x = 0;
    for (int b = 0; b < L - rows2<L>::SH; b++)
        if ((p >> (rows2<L>::SH + b)) & 1)
            x ^= rows2<L>::r[b];
    return x;
}
template <int L> __device__ __forceinline__ uint32_t xterm2_dev(uint32_t p)
{
    uint32_t x = 0;
#pragma unroll
    for (int b = 0; b < L - rows2<L>::SH; b++)
        x ^= ((p >> (rows2<L>::SH + b)) & 1) ? rows2<L>::r[b] : 0u;
    return x;
}

// stages S0 .. log2(P) - 1 of the P-point radix-2 DIF network on x[0 .. P) (fft.h:286-291 in decimation-in-frequency
// form); the factor the lower output owes = thread twiddle w[s] (when TW) x the constant W_P^((k mod h) << s)
template <bool REV, bool TW, int P, int S0> __device__ __forceinline__ void dif_p(float2 *x, const float2 *w)
{
    constexpr int LP = P == 4 ? 2 : P == 8 ? 3 : P == 16 ? 4 : 5;
#pragma unroll
    for (int s = S0; s < LP; s++) {
        const int h = P >> (s + 1);
#pragma unroll
        for (int k = 0; k < P; k++) {
            if ((k & h) != 0)
                continue;
            const float2 a = x[k], b = x[k + h];
            x[k] = float2{ a.x + b.x, a.y + b.y };
            float2 d = float2{ a.x - b.x, a.y - b.y };
            const int e = ((k & (h - 1)) << s) * (32 / P); // W_32 exponent, 0 .. 15
            if (e == 8) {
                d = REV ? float2{ -d.y, d.x } : float2{ d.y, -d.x };
            } else if (e != 0) {
                const float cr = fft32::kC32[e], ci = REV ? fft32::kS32[e] : -fft32::kS32[e];
                d = float2{ d.x * cr - d.y * ci, d.x * ci + d.y * cr };
            }
            if constexpr (TW)
                d = passes::cmul(d, w[s]);
            x[k + h] = d;
        }
    }
}

// RADIX 4 (N = 256 = 4^4 only, P = 4): every pass is ONE radix-4 DIF stage (fft.h:311-349) on the lane's four points --
// output q owes W_N^(q v 4^i), three thread twiddles per pass from a [pass][q][lane] table -- and the last pass takes the
// block w = digit_reverse4(t), so that register k holds X[t + 64 k] (fft.h:351-355 folded into the assignment).
template <int L, bool REV, bool CONV, int REAL = 0, int RADIX = 2>
__global__ __launch_bounds__(256) void sdsp_fft_wave_f32(float2 *__restrict__ data, const float2 *__restrict__ tw, uint64_t batch, float scale,
                                                         const float2 *__restrict__ h)
{
    static_assert(!(CONV && REV), "the fused convolution belongs to forward plans");
    static_assert(REAL == 0 || (!CONV && (REAL == 1) == !REV), "real-input packing: split forward / merge inverse");
    static_assert(RADIX == 2 || (RADIX == 4 && L == 8), "radix-4 stages: N = 256");
    constexpr int LP = L - 6, P = 1 << LP, N = 1 << L, NP = (L + LP - 1) / LP, REM = L - LP * (NP - 1);
    constexpr int TPW = P >= 16 ? 1 : 16 / P, R = P * TPW; // transforms per wave, registers per lane
    __shared__ __attribute__((aligned(16))) float2 lds_all[4][64 * R];
    const uint32_t t = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float2 *lds = lds_all[wave];
    const uint64_t f0 = (static_cast<uint64_t>(blockIdx.x) * 4 + wave) * TPW; // the wave's first transform
    if (f0 >= batch)
        return; // wave-uniform; the kernel has no barrier
    const uint32_t live = batch - f0 >= (uint64_t)TPW ? (uint32_t)TPW : (uint32_t)(batch - f0);

    float2 x[R];
    float2 *const blk = data + f0 * N + t;
#pragma unroll
    for (int j = 0; j < R; j++) {
        x[j] = float2{ 0.0f, 0.0f };
        if (TPW == 1 || (uint32_t)(j / P) < live)
            x[j] = gload(blk + 64 * j);
    }

    // the last pass's block, and the row (register k -> X[t + 64 row(k)]) that follows from it
    const uint32_t w = RADIX == 2 ? (__brev(t) >> 26) : (((t & 3u) << 4) | (t & 12u) | (t >> 4));
    auto row_of = [](int k) { return RADIX == 2 ? (int)(__brev((uint32_t)k) >> (32 - LP)) : k; };
    auto transform = [&](auto rev_tag) {
        constexpr bool RV = decltype(rev_tag)::value;
        auto run_pass = [&](auto pass_tag) {
            constexpr int I = decltype(pass_tag)::value;
            constexpr bool last = I == NP - 1;
            constexpr int NW = RADIX == 4 ? 3 : LP; // thread twiddles per pass
            float2 wt[NW];
            if constexpr (!last) {
#pragma unroll
                for (int s = 0; s < NW; s++) {
                    wt[s] = tw[(I * NW + s) * 64 + t];
                    if constexpr (RV != REV)
                        wt[s].y = -wt[s].y;
                }
            }
#pragma unroll
            for (int g = 0; g < TPW; g++) {
                if constexpr (RADIX == 4) {
                    float2 *y = x + g * P;
                    passes::bfly4<RV>(y[0], y[1], y[2], y[3]);
                    if constexpr (!last) {
                        y[1] = passes::cmul(y[1], wt[0]);
                        y[2] = passes::cmul(y[2], wt[1]);
                        y[3] = passes::cmul(y[3], wt[2]);
                    }
                } else {
                    dif_p<RV, !last, P, (last ? LP - REM : 0)>(x + g * P, wt);
                }
            }
        };
        // slot address of register (g, k) in pass I's layout = g N + (A_I(t) ^ B_I(k))
        auto a_of = [&](auto pass_tag) -> uint32_t {
            constexpr int I = decltype(pass_tag)::value;
            if constexpr (I == NP - 1) {
                return (P * w) ^ xterm2_dev<L>(P * w);
            } else {
                constexpr uint32_t sg = (uint32_t)N >> (LP * (I + 1));
                const uint32_t base = (t / sg) * (sg * P) + (t % sg);
                return base ^ xterm2_dev<L>(base);
            }
        };
        auto exchange = [&](auto from_tag) { // pass I's layout -> pass I + 1's
            constexpr int I = decltype(from_tag)::value;
            constexpr uint32_t s_from = I == NP - 1 ? 1u : (uint32_t)N >> (LP * (I + 1));
            constexpr uint32_t s_to = I + 1 == NP - 1 ? 1u : (uint32_t)N >> (LP * (I + 2));
            wave_lds_sync(); // the previous exchange's reads (other lanes') come first
            {
                uint32_t a = a_of(from_tag);
                asm volatile("" : "+v"(a)); // one v_xor per access instead of R live addresses
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const uint32_t k = (uint32_t)(j % P), g = (uint32_t)(j / P);
                    lds[g * N + (a ^ ((s_from * k) ^ xterm2<L>(s_from * k)))] = x[j];
                }
            }
            wave_lds_sync();
            {
                uint32_t a = a_of(std::integral_constant<int, I + 1>{});
                asm volatile("" : "+v"(a));
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const uint32_t k = (uint32_t)(j % P), g = (uint32_t)(j / P);
                    x[j] = lds[g * N + (a ^ ((s_to * k) ^ xterm2<L>(s_to * k)))];
                }
            }
        };
        run_pass(std::integral_constant<int, 0>{});
        exchange(std::integral_constant<int, 0>{});
        run_pass(std::integral_constant<int, 1>{});
        exchange(std::integral_constant<int, 1>{});
        run_pass(std::integral_constant<int, 2>{});
        if constexpr (NP > 3) {
            exchange(std::integral_constant<int, 2>{});
            run_pass(std::integral_constant<int, 3>{});
        }
    };
    if constexpr (REAL == 2) {
#pragma unroll
        for (int g = 0; g < TPW; g++)
            real_pack_stage<P, true>(x + g * P, t, h, [](int r) { return r; });
    }
    transform(std::integral_constant<bool, REV>{});
    if constexpr (REAL == 1) {
#pragma unroll
        for (int g = 0; g < TPW; g++)
            real_pack_stage<P, false>(x + g * P, t, h, row_of); // row_of is an involution: the register that holds row r
    }

    if constexpr (CONV) {
        // register (g, k) holds X_g[t + 64 rev(k)]: multiply by h there, renumber to the first pass's input layout
        float2 z[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int k = j % P, g = j / P, row = row_of(k);
            z[g * P + row] = passes::cmul(x[j], h[t + 64 * row]);
        }
#pragma unroll
        for (int j = 0; j < R; j++)
            x[j] = z[j];
        transform(std::integral_constant<bool, true>{});
    }

#pragma unroll
    for (int j = 0; j < R; j++) {
        const int k = j % P, g = j / P, row = row_of(k);
        float2 v = x[j];
        if constexpr (REV || CONV) { // reverse_fft::ScaleValues, fft.h:128-132
            v.x *= scale;
            v.y *= scale;
        }
        if (TPW == 1 || (uint32_t)g < live)
            gstore(blk + 64 * (g * P + row), v);
    }
}

template <int L, bool REV, bool CONV, int REAL = 0, int RADIX = 2> int launch_w2(const fft_reg_args &a, hipStream_t s)
{
    constexpr int P = 1 << (L - 6), TPW = P >= 16 ? 1 : 16 / P;
    const uint64_t waves = (a.batch + TPW - 1) / TPW, blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL((sdsp_fft_wave_f32<L, REV, CONV, REAL, RADIX>), dim3((uint32_t)blocks), dim3(256), 0, s, reinterpret_cast<float2 *>(a.data),
                       reinterpret_cast<const float2 *>(a.tw), a.batch, a.scale, reinterpret_cast<const float2 *>(a.tw2));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_wave launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
template <int L, int RADIX = 2> int launch_w2_mode(const fft_reg_args &a, hipStream_t s)
{
    if (a.real_mode == 1 || a.real_mode == 2) { // real-input packing: tw2 = W_2N^k
        if (!a.tw2 || (a.real_mode == 2) != (a.reverse != 0))
            return fail(SDSP_HIP_ERR_INVALID_ARG, "real-input packing: forward plans split, reverse plans merge");
        return a.real_mode == 1 ? launch_w2<L, false, false, 1, RADIX>(a, s) : launch_w2<L, true, false, 2, RADIX>(a, s);
    }
    if (a.real_mode == 3) {
        if (a.reverse || !a.tw2)
            return fail(SDSP_HIP_ERR_INVALID_ARG, "fused convolution needs a forward plan and h");
        return launch_w2<L, false, true, 0, RADIX>(a, s);
    }
    return a.reverse ? launch_w2<L, true, false, 0, RADIX>(a, s) : launch_w2<L, false, false, 0, RADIX>(a, s);
}

template <typename C, typename S, int RADIX, bool REV, bool CONV = false> int launch_t(const fft_reg_args &a, S scale, hipStream_t s)
{
    constexpr int WAVES = sizeof(C) == 8 ? 4 : 2;
    const uint64_t blocks = (a.batch + WAVES - 1) / WAVES;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL((sdsp_fft1024_wave<C, S, RADIX, REV, WAVES, CONV>), dim3((uint32_t)blocks), dim3(64 * WAVES), 0, s,
                       reinterpret_cast<C *>(a.data), reinterpret_cast<const C *>(a.tw), a.batch, scale,
                       reinterpret_cast<const C *>(a.tw2));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_wave launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

template <typename C, typename S, int RADIX, int REAL> int launch_real_t(const fft_reg_args &a, S scale, hipStream_t s)
{
    constexpr int WAVES = 4;
    const uint64_t blocks = (a.batch + WAVES - 1) / WAVES;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL((sdsp_fft1024_wave<C, S, RADIX, REAL == 2, WAVES, false, REAL>), dim3((uint32_t)blocks), dim3(64 * WAVES), 0, s,
                       reinterpret_cast<C *>(a.data), reinterpret_cast<const C *>(a.tw), a.batch, scale,
                       reinterpret_cast<const C *>(a.tw2));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_wave launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

template <typename C, typename S> int launch_c(const fft_reg_args &a, S scale, hipStream_t s)
{
    if constexpr (sizeof(C) == 8) {
        if (a.real_mode == 1 || a.real_mode == 2) { // real-input packing: tw2 = W_2N^k
            if (!a.tw2 || (a.real_mode == 2) != (a.reverse != 0))
                return fail(SDSP_HIP_ERR_INVALID_ARG, "real-input packing: forward plans split, reverse plans merge");
            if (a.real_mode == 1)
                return a.radix == 2 ? launch_real_t<C, S, 2, 1>(a, scale, s) : launch_real_t<C, S, 4, 1>(a, scale, s);
            return a.radix == 2 ? launch_real_t<C, S, 2, 2>(a, scale, s) : launch_real_t<C, S, 4, 2>(a, scale, s);
        }
    }
    if (a.real_mode == 3) { // fused convolution: forward plan, tw2 = h
        if (a.reverse || !a.tw2)
            return fail(SDSP_HIP_ERR_INVALID_ARG, "fused convolution needs a forward plan and h");
        return a.radix == 2 ? launch_t<C, S, 2, false, true>(a, scale, s) : launch_t<C, S, 4, false, true>(a, scale, s);
    }
    if (a.radix == 2)
        return a.reverse ? launch_t<C, S, 2, true>(a, scale, s) : launch_t<C, S, 2, false>(a, scale, s);
    return a.reverse ? launch_t<C, S, 4, true>(a, scale, s) : launch_t<C, S, 4, false>(a, scale, s);
}
} // namespace

bool fft_wave_supports(uint32_t n, int radix)
{
    return n == 1024 && (radix == 2 || radix == 4);
}

// N = 256 / 512 / 2048 radix-2 stages, N = 256 radix-4 stages, f32: a.tw = the plan's thread-twiddle table (capi.hip:
// upload_thread_twiddles_wave)
bool fft_wave2_supports(uint32_t n, int radix)
{
    return (radix == 2 && (n == 256 || n == 512 || n == 2048)) || (radix == 4 && n == 256);
}

int launch_fft_wave2_f32(const fft_reg_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a.radix == 4 && a.n == 256)
        return launch_w2_mode<8, 4>(a, s);
    if (a.radix == 2) {
        switch (a.n) {
        case 256: return launch_w2_mode<8>(a, s);
        case 512: return launch_w2_mode<9>(a, s);
        case 2048: return launch_w2_mode<11>(a, s);
        default: break;
        }
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the one-wave kernels");
}

// a.tw: the plan's register-pass thread-twiddle table (twt_reg), in the plan's precision
int launch_fft_wave_f32(const fft_reg_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    if (!fft_wave_supports(a.n, a.radix))
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the one-wave kernels");
    return launch_c<float2, float>(a, a.scale, reinterpret_cast<hipStream_t>(stream));
}

int launch_fft_wave_f64(const fft_reg_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    if (!fft_wave_supports(a.n, a.radix))
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the one-wave kernels");
    return launch_c<double2, double>(a, a.scale_d, reinterpret_cast<hipStream_t>(stream));
}
} // namespace sdsp_hip

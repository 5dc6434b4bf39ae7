// fft4096_kernels.h -- device code of the batched N = 4096 complex f32 FFT for gfx950 (BASELINE configs 2 and 5).
// Included by fft4096.hip (the product) and by tools/lab_fft4096.hip (the measurement harness).
//
// GPU form of sdsp::fft_radix4<T,4096> (fft.h:301-360): the same six radix-4 DIF stages
// (fft.h:311-349), the +-i rotations done by swap/negate (fft.h:339-345), the base-4 digit
// reversal (fft.h:351-355) and the reverse-direction 1/N scale (fft.h:128-132) -- organised for
// the machine instead of for a scalar core:
//
//   * one 256-thread workgroup per transform, 16 points per thread in registers; the six stages
//     run as three register passes of two stages each (strides 1024/256, 64/16, 4/1) with two
//     exchanges through a 32 KiB LDS tile.  The tile is XOR-swizzled (addr = p ^ f(p >> 8)) so that
//     every ds_write_b64 / ds_read_b64 / ds_read_b128 of all three access patterns is bank-conflict
//     free without padding; the butterflies are in place, so a thread rewrites only slots it read.
//   * the twiddle W_N^(r*pos) that stage s owes stage s+1 factors into a per-thread part that is the
//     same for every transform (W^(r*t): 12 complex values per thread, read with coalesced loads from the
//     plan's thread-twiddle table -- the values of the row W_4096^j, precomputed in double, laid out
//     [value][thread]) and a compile-time W_16 constant.
//   * the digit reversal costs nothing: the last pass is assigned so that thread t holds the block
//     whose outputs land at t + 256*j, i.e. stores are as coalesced as the loads (512 contiguous
//     bytes per wave instruction both ways) and HBM sees every element exactly once each way.
//   * one workgroup per transform, streaming (non-temporal) loads and stores.
//
// HBM-bound by design: 64 KiB of traffic against ~250 kflop per transform.  No MFMA.
// The scheduling knobs of the radix-4 kernel (all bit-identical arithmetic) are template parameters; the product
// instantiates the measured best, tools/lab_fft4096.hip the whole grid (results: DESIGN.md section 5.1).
#pragma once

#include <hip/hip_runtime.h>

#include "fft32.h"

#include "fft_passes.h"

namespace sdsp_hip
{
namespace fft4096
{
constexpr float kC1 = 0.92387953251128673848f; // cos(pi/8)
constexpr float kS1 = 0.38268343236508978178f; // sin(pi/8)
constexpr float kH = 0.70710678118654752440f;  // sqrt(1/2)

__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return float2{ a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return float2{ a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return float2{ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x };
}
// a * (cr -/+ i*ci): compile-time constant, conjugated for the reverse transform
template <bool REV> __device__ __forceinline__ float2 cmulk(float2 a, float cr, float ci_fwd)
{
    const float ci = REV ? -ci_fwd : ci_fwd;
    return float2{ a.x * cr - a.y * ci, a.x * ci + a.y * cr };
}
// multiply by W_4 = -i (forward) / +i (reverse)
template <bool REV> __device__ __forceinline__ float2 rot90(float2 a)
{
    return REV ? float2{ -a.y, a.x } : float2{ a.y, -a.x };
}
// multiply by W_16^e, e compile-time
template <bool REV, int E> __device__ __forceinline__ float2 mul_w16(float2 a)
{
    if constexpr (E == 0)
        return a;
    else if constexpr (E == 1)
        return cmulk<REV>(a, kC1, -kS1);
    else if constexpr (E == 2) // h*(1 - i)
        return REV ? float2{ kH * (a.x - a.y), kH * (a.x + a.y) } : float2{ kH * (a.x + a.y), kH * (a.y - a.x) };
    else if constexpr (E == 3)
        return cmulk<REV>(a, kS1, -kC1);
    else if constexpr (E == 4)
        return rot90<REV>(a);
    else if constexpr (E == 6) // h*(-1 - i)
        return REV ? float2{ -kH * (a.x + a.y), kH * (a.x - a.y) } : float2{ kH * (a.y - a.x), -kH * (a.x + a.y) };
    else { // E == 9: -W_16^1
        static_assert(E == 9, "unexpected W_16 exponent");
        return cmulk<REV>(a, -kC1, kS1);
    }
}

// in-place radix-4 DIF butterfly on elements at offsets 0, g, 2g, 3g: fft.h:342-345
template <bool REV> __device__ __forceinline__ void bfly4(float2 &a, float2 &b, float2 &c, float2 &d)
{
    const float2 t0 = a + c, t1 = a - c, t2 = b + d, t3 = rot90<REV>(b - d);
    a = t0 + t2;
    b = t1 + t3;
    c = t0 - t2;
    d = t1 - t3;
}

// Two consecutive radix-4 DIF stages on 16 registers; x[k] is the element at base + k*stride.
// Stage X pairs k = j + 4r over r; its output twiddle W^(r*pos), pos = (thread part) + j*4*stride...
// factors into w1[r-1] (thread part, general) times W_16^(r*j) (constant).  Stage Y pairs
// k = 4r + r' over r' with output twiddle w2[r'-1] (general, absent in the last pass).
template <bool REV, bool TW1, bool TW2>
__device__ __forceinline__ void two_stages(float2 (&x)[16], const float2 (&w1)[3], const float2 (&w2)[3])
{
#pragma unroll
    for (int j = 0; j < 4; j++)
        bfly4<REV>(x[j], x[j + 4], x[j + 8], x[j + 12]);
    // constants W_16^(r*j)
    x[5] = mul_w16<REV, 1>(x[5]);
    x[6] = mul_w16<REV, 2>(x[6]);
    x[7] = mul_w16<REV, 3>(x[7]);
    x[9] = mul_w16<REV, 2>(x[9]);
    x[10] = mul_w16<REV, 4>(x[10]);
    x[11] = mul_w16<REV, 6>(x[11]);
    x[13] = mul_w16<REV, 3>(x[13]);
    x[14] = mul_w16<REV, 6>(x[14]);
    x[15] = mul_w16<REV, 9>(x[15]);
    if constexpr (TW1) {
#pragma unroll
        for (int r = 1; r < 4; r++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                x[j + 4 * r] = cmul(x[j + 4 * r], w1[r - 1]);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        bfly4<REV>(x[4 * r], x[4 * r + 1], x[4 * r + 2], x[4 * r + 3]);
        if constexpr (TW2) {
            x[4 * r + 1] = cmul(x[4 * r + 1], w2[0]);
            x[4 * r + 2] = cmul(x[4 * r + 2], w2[1]);
            x[4 * r + 3] = cmul(x[4 * r + 3], w2[2]);
        }
    }
}

// Streaming (non-temporal) global accesses: every element is touched exactly once each way, so
// keeping it out of the L2 / Infinity-Cache replacement state measured +11 % on this access shape
// (tools/membench.hip: 5.36 -> 5.96 TB/s read+write in place).
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 gload(const float2 *p)
{
    const v2f_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(p));
    return float2{ v.x, v.y };
}
__device__ __forceinline__ void gstore(float2 *p, float2 a)
{
    const v2f_t v = { a.x, a.y };
    __builtin_nontemporal_store(v, reinterpret_cast<v2f_t *>(p));
}

__device__ __forceinline__ uint32_t rev4bits(uint32_t v) // reverse the low 4 bits
{
    return __brev(v) >> 28;
}
constexpr int crev4(int j) { return ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3); }

// LDS addressing of one thread (float2 units), see the kernel below
struct lds_map {
    float2 *b_even, *b_odd; // pass B bases
    uint32_t c_base, c_x, t;
};
// BITREV: pass C takes block bit_reverse8(t) (radix 2) instead of digit_reverse4(t) (radix 4)
template <bool BITREV> __device__ __forceinline__ lds_map make_lds_map(float2 *lds, uint32_t t)
{
    lds_map mp;
    const uint32_t rr = t & 15, b = t >> 4;
    // pass A writes p = t + 256 k: slot 256 k + (t ^ (rev4bits(k) << 1))
    // pass B (in place) p = 256 b + rr + 16 k: X = rev4bits(b) << 1 flips rr's bits 3..1 and k's bit 0,
    //   i.e. slot b_base + 16 (k ^ b_flip) = (b_base +- 16 b_flip) + 16 k for even / odd k
    const uint32_t xb = rev4bits(b) << 1;
    const uint32_t b_base = 256 * b + (rr ^ (xb & 15));
    const uint32_t b_flip = (xb >> 4) & 1;
    mp.b_even = lds + b_base + 16 * b_flip;
    mp.b_odd = lds + b_base - 16 * b_flip;
    // pass C reads p = 16 m + k, m chosen so that the outputs land at t + 256 j
    const uint32_t m = BITREV ? (__brev(t) >> 24) : (((t & 3) << 6) | (((t >> 2) & 3) << 4) | (((t >> 4) & 3) << 2) | (t >> 6));
    const uint32_t xc = rev4bits(m >> 4) << 1;
    mp.c_base = 256 * (m >> 4) + 16 * ((m & 15) ^ (xc >> 4));
    mp.c_x = (xc >> 1) & 7; // pair index i -> i ^ c_x
    mp.t = t;
    return mp;
}
// Loop-invariant address VECTORS are deliberately not kept in registers (they cost ~40 VGPRs): pass B uses two bases +
// immediate offsets, passes A and C rebuild theirs with one v_xor per access from a value the optimiser cannot hoist.
__device__ __forceinline__ void lds_write_a(float2 *lds, const lds_map &mp, const float2 (&x)[16])
{
    uint32_t ta = mp.t;
    asm volatile("" : "+v"(ta));
#pragma unroll
    for (int k = 0; k < 16; k++)
        lds[256 * k + (ta ^ ((__brev((uint32_t)k) >> 28) << 1))] = x[k];
}
// ORDER 0: registers in ascending order; 1: in the order the first butterflies consume them (0, 4, 8, 12, 1, 5, ...)
template <int ORDER> __device__ __forceinline__ void lds_read_b(const lds_map &mp, float2 (&x)[16])
{
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int k = ORDER == 1 ? 4 * (j & 3) + (j >> 2) : j;
        x[k] = (k & 1) ? mp.b_odd[16 * k] : mp.b_even[16 * k];
    }
}
__device__ __forceinline__ void lds_write_b(const lds_map &mp, const float2 (&x)[16])
{
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k & 1)
            mp.b_odd[16 * k] = x[k];
        else
            mp.b_even[16 * k] = x[k];
    }
}
__device__ __forceinline__ void lds_read_c(const float2 *lds, const lds_map &mp, float2 (&x)[16])
{
    uint32_t cx = mp.c_x;
    asm volatile("" : "+v"(cx));
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float4 v = *reinterpret_cast<const float4 *>(&lds[mp.c_base + 2 * (i ^ cx)]);
        x[2 * i] = float2{ v.x, v.y };
        x[2 * i + 1] = float2{ v.z, v.w };
    }
}

// ---- sdsp::fft_radix4<T, 4096>, one workgroup per transform ----------------------------------------------------
// Scheduling knobs (identical arithmetic, identical bits):
//   BAR   0: no barrier after the last LDS read (a workgroup's only transform needs none)
//         1: barrier after the last LDS read, before the last two stages (keeps the four waves' stores together)
//         2: barrier right before the stores
//   SORD  order in which a thread issues its 16 row stores (rows are 2 KiB apart): 0: 0,4,8,12,1,5,... (register
//         order), 1: ascending, 2: bit-reversed (0,8,4,12,...)
//   LORD  order of the 16 row loads: 0 ascending, 1 bit-reversed, 2 the order the first butterflies consume them
//   LDSB  order of pass B's LDS reads: 0 ascending, 1 consumption order
//   WAVES launch bound (waves per SIMD the register allocator must leave room for)
//   MAP   0: workgroup b transforms row b; 1 (lab only, batch % 8 == 0): the workgroups of XCD b % 8 walk their own
//         contiguous eighth of the batch
template <bool REV, int BAR, int SORD, int LORD, int LDSB, int WAVES, int MAP = 0>
__global__ __launch_bounds__(256, WAVES) void sdsp_fft4096_r4_f32(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                                  uint64_t batch, float scale)
{
    // LDS slot of logical position p (8-byte units): p ^ (rev4bits(p >> 8) << 1)
    __shared__ __attribute__((aligned(16))) float2 lds[4096];
    const uint32_t t = threadIdx.x;

    // ---- per-thread twiddles, identical for every transform: fetched once (fft.h:309 uses the
    // same single row exp(-+2 pi i j / N) of the table).  `tw` is the plan's THREAD-TWIDDLE table (capi.hip:
    // upload_thread_twiddles_4096), the same values as the row W_4096^j laid out [value][thread] so that these are
    // coalesced loads: gathering them from the row (strides of 8..96 bytes per lane) cost about as many cache-line
    // requests as the transform's data
    float2 wA1[3], wA2[3], wB1[3], wB2[3];
    const uint32_t rr = t & 15;
#pragma unroll
    for (int r = 1; r < 4; r++) {
        wA1[r - 1] = tw[(r - 1) * 256 + t];        // W_4096^(r t)
        wA2[r - 1] = tw[(r + 2) * 256 + t];        // W_1024^(r t)  = W_4096^(4 r t)
        wB1[r - 1] = tw[1536 + (r - 1) * 16 + rr]; // W_256^(r rr)  = W_4096^(16 r rr)
        wB2[r - 1] = tw[1536 + (r + 2) * 16 + rr]; // W_64^(r rr)   = W_4096^(64 r rr)
    }
    const lds_map mp = make_lds_map<false>(lds, t);

    const uint64_t f = MAP == 1 ? (blockIdx.x & 7u) * (batch >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    if (f >= batch)
        return;
    float2 x[16];
    // plain pointers: the compiler builds 64-bit VGPR addresses for most rows (2 KiB apart), but at 84 VGPRs that costs this
    // kernel nothing, and the buffer-resource form that pays off in fft_big.hip and in the convolution below measured
    // 0.5 points SLOWER here (A/B in one run, 2 GiB batches: 75.1-75.2 % against 75.7-76.1 %)
    const float2 *src = data + f * 4096 + t;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int k = LORD == 1 ? crev4(j) : LORD == 2 ? 4 * (j & 3) + (j >> 2) : j;
        x[k] = gload(src + 256 * k);
    }
    // ---- pass A: stages 0,1 (groups 1024, 256), fft.h:311-349 with i = 0,1
    two_stages<REV, true, true>(x, wA1, wA2);
    lds_write_a(lds, mp, x);
    __syncthreads();
    // ---- pass B: stages 2,3 (groups 64, 16)
    lds_read_b<LDSB>(mp, x);
    two_stages<REV, true, true>(x, wB1, wB2);
    lds_write_b(mp, x);
    __syncthreads();
    // ---- pass C: stages 4,5 (groups 4, 1); only W_16 constants
    lds_read_c(lds, mp, x);
    if constexpr (BAR == 1)
        __syncthreads();
    two_stages<REV, false, false>(x, wA1, wA2);
    if constexpr (BAR == 2)
        __syncthreads();

    // ---- store; register k = 4 d1 + d0 holds X[t + 256 * (4 d0 + d1)]: fft.h:351-355 folded
    float2 *dst = data + f * 4096 + t;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int row = SORD == 1 ? j : SORD == 2 ? crev4(j) : 4 * (j & 3) + (j >> 2);
        const int k = 4 * (row & 3) + (row >> 2); // the register that holds that row
        float2 v = x[k];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            v.x *= scale;
            v.y *= scale;
        }
        gstore(dst + 256 * row, v);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused fast convolution: y = IFFT( FFT(x) .* H ) per transform in ONE kernel.
// The forward transform above leaves thread t holding X[t + 256 j], j < 16 -- which is exactly the
// layout its own first pass consumes -- so after the per-bin multiply the reverse transform (fft.h
// reverse_fft policy: conjugate twiddles, +i rotations, 1/N scale) runs on the same registers and the
// same LDS tile.  HBM sees one read and one write per element instead of three of each.

// all six stages on one transform held as x[k] = element t + 256 k; returns with
// x[k] = result[t + 256 * (4 (k & 3) + (k >> 2))].  w*: the twiddles; conjugated here when CONJ (the fused convolution
// runs its reverse half from the FORWARD plan's table; fft_mix.hip passes tables already folded for the direction).
template <bool REV, bool CONJ = REV>
__device__ __forceinline__ void fft4096_in_regs(float2 (&x)[16], float2 *lds, const lds_map &mp, const float2 (&wA1)[3],
                                                const float2 (&wA2)[3], const float2 (&wB1)[3], const float2 (&wB2)[3])
{
    float2 a1[3], a2[3], b1[3], b2[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        a1[r] = float2{ wA1[r].x, CONJ ? -wA1[r].y : wA1[r].y };
        a2[r] = float2{ wA2[r].x, CONJ ? -wA2[r].y : wA2[r].y };
        b1[r] = float2{ wB1[r].x, CONJ ? -wB1[r].y : wB1[r].y };
        b2[r] = float2{ wB2[r].x, CONJ ? -wB2[r].y : wB2[r].y };
    }
    two_stages<REV, true, true>(x, a1, a2);
    lds_write_a(lds, mp, x);
    __syncthreads();
    lds_read_b<0>(mp, x);
    two_stages<REV, true, true>(x, b1, b2);
    lds_write_b(mp, x);
    __syncthreads();
    lds_read_c(lds, mp, x);
    __syncthreads(); // every read of the tile is done: the next transform may overwrite it
    two_stages<REV, false, false>(x, a1, a2);
}

template <bool REVERSE_HALF = true> // (a template so that only the translation unit that launches it instantiates it)
__global__ __launch_bounds__(256, 2) void sdsp_fft4096_conv_f32(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                               const float2 *__restrict__ h, uint64_t batch)
{
    __shared__ __attribute__((aligned(16))) float2 lds[4096];
    const uint32_t t = threadIdx.x;
    float2 wA1[3], wA2[3], wB1[3], wB2[3];
    const uint32_t rr = t & 15;
#pragma unroll
    for (int r = 1; r < 4; r++) { // thread-twiddle table, see sdsp_fft4096_r4_f32
        wA1[r - 1] = tw[(r - 1) * 256 + t];
        wA2[r - 1] = tw[(r + 2) * 256 + t];
        wB1[r - 1] = tw[1536 + (r - 1) * 16 + rr];
        wB2[r - 1] = tw[1536 + (r + 2) * 16 + rr];
    }
    const lds_map mp = make_lds_map<false>(lds, t);
    const __amdgpu_buffer_rsrc_t hrows = fft32::make_rows(h, 4096 * sizeof(float2));

    for (uint64_t f = blockIdx.x; f < batch; f += gridDim.x) {
        float2 x[16], z[16];
        // data and H rows through buffer resources (fft32.h: make_rows): with plain pointers the 32 + 16 row addresses were
        // 64-bit VGPR pairs and the kernel needed 126 VGPRs (four workgroups per CU); this form needs 92 (five) -- 57 -> 66 %
        const __amdgpu_buffer_rsrc_t rows = fft32::make_rows(data + f * 4096, 4096 * sizeof(float2));
#pragma unroll
        for (int k = 0; k < 16; k++)
            x[k] = fft32::row_load<true>(rows, t * 8u, 256u * 8u * k);
        fft4096_in_regs<false>(x, lds, mp, wA1, wA2, wB1, wB2);
        // x[k] = X[t + 256 j], j = 4 (k & 3) + (k >> 2): multiply by H[t + 256 j] and renumber so that
        // z[j] is element t + 256 j of the spectrum -- the input layout of the transform's first pass
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int j = 4 * (k & 3) + (k >> 2);
            z[j] = cmul(x[k], fft32::row_load<false>(hrows, t * 8u, 256u * 8u * j));
        }
        fft4096_in_regs<REVERSE_HALF>(z, lds, mp, wA1, wA2, wB1, wB2);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            float2 v = z[k];
            v.x *= 1.0f / 4096.0f; // reverse_fft::ScaleValues, fft.h:128-132
            v.y *= 1.0f / 4096.0f;
            fft32::row_store<true>(rows, t * 8u, 256u * 8u * (4 * (k & 3) + (k >> 2)), v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sdsp::fft_radix2<T,4096> (fft.h:258-299) with the same machinery: twelve radix-2 DIF stages as three
// register passes of four stages (pair distances 2048..256, 128..16, 8..1), the same in-place LDS
// tile and XOR swizzle (the bank analysis carries over to the bit-reversed block assignment), four
// thread twiddles per pass instead of six.  The bit reversal (fft.h:269-273) is folded into the last
// pass's assignment: thread t takes block bit_reverse8(t), whose outputs land at t + 256*bit_reverse4(k).
template <bool REV>
__global__ __launch_bounds__(256, 3) void sdsp_fft4096_r2_f32(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                              uint64_t batch, float scale)
{
    __shared__ __attribute__((aligned(16))) float2 lds[4096];
    const uint32_t t = threadIdx.x;
    const uint32_t rr = t & 15;
    float2 wA[4], wB[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { // thread-twiddle table (coalesced), see sdsp_fft4096_r4_f32
        wA[j] = tw[j * 256 + t];        // stage j of pass A: W_4096^(t 2^j)
        wB[j] = tw[1024 + j * 16 + rr]; // pass B: W_4096^(16 rr 2^j)
    }
    const lds_map mp = make_lds_map<true>(lds, t);
    const uint64_t f = blockIdx.x;
    if (f >= batch)
        return;
    float2 x[16];
    const float2 *src = data + f * 4096 + t;
#pragma unroll
    for (int k = 0; k < 16; k++)
        x[k] = gload(src + 256 * k);
    passes::r2_pass<REV, true, 0>::run(x, wA);
    lds_write_a(lds, mp, x);
    __syncthreads();
    lds_read_b<0>(mp, x);
    passes::r2_pass<REV, true, 0>::run(x, wB);
    lds_write_b(mp, x);
    __syncthreads();
    lds_read_c(lds, mp, x);
    __syncthreads(); // keeps the four waves' stores together: without it 77.3 % -> 73.7 % (round 1)
    passes::r2_pass<REV, false, 0>::run(x, wA);
    float2 *dst = data + f * 4096 + t;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        float2 v = x[k];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            v.x *= scale;
            v.y *= scale;
        }
        gstore(dst + 256 * (int)(__brev((uint32_t)k) >> 28), v);
    }
}
} // namespace fft4096
} // namespace sdsp_hip

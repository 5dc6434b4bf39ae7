// fft4096.hip -- batched N = 4096 radix-4 complex f32 FFT for gfx950 (BASELINE configs 2 and 5).
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
//   * one workgroup per transform, streaming (non-temporal) loads and stores: 74.9 % of the 8 TB/s HBM
//     peak, which is where in-place read+write traffic plateaus on this part whatever its shape
//     (tools/delaybench.hip).  The variant table at the bottom keeps what was tried instead -- persistent
//     workgroups with register prefetch, two or more consecutive transforms per workgroup, occupancy caps,
//     cache policies, row orders -- with the measured result of each.
//
// HBM-bound by design: 64 KiB of traffic against ~250 kflop per transform.  No MFMA.
#include <hip/hip_runtime.h>

#include "fft_passes.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
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
template <bool NT> __device__ __forceinline__ float2 gload(const float2 *p)
{
    if constexpr (NT) {
        const v2f_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(p));
        return float2{ v.x, v.y };
    } else {
        return *p;
    }
}
template <bool NT> __device__ __forceinline__ void gstore(float2 *p, float2 a)
{
    if constexpr (NT) {
        const v2f_t v = { a.x, a.y };
        __builtin_nontemporal_store(v, reinterpret_cast<v2f_t *>(p));
    } else {
        *p = a;
    }
}

__device__ __forceinline__ uint32_t rev4bits(uint32_t v) // reverse the low 4 bits
{
    return __brev(v) >> 28;
}

// PREFETCH: fetch transform i+1 into registers during passes B/C of transform i (costs 32 VGPRs).
// WAVES: occupancy the register allocator must leave room for (waves per SIMD = workgroups per CU).
// ORD: the order in which a thread issues its 16 row loads / row stores (rows are 2 KiB apart):
//   0 loads ascending, stores 0,4,8,12,1,5,...   1 both bit-reversed   2 both (5j+3) mod 16
//   3 loads ascending, stores (5j+3) mod 16      4 loads (5j+3) mod 16, stores as 0
constexpr int row_order(int ord, int j) { return ord == 1 ? ((j & 1) << 3 | (j & 2) << 1 | (j & 4) >> 1 | (j & 8) >> 3) : (5 * j + 3) % 16; }
template <bool REV, int PREFETCH, int WAVES, int NTP, int CHUNK = 2, int ORD = 0>
__global__ __launch_bounds__(256, WAVES) void sdsp_fft4096_r4_f32(float2 *__restrict__ data,
                                                                  const float2 *__restrict__ tw,
                                                                  uint64_t batch, float scale)
{
    // NTP: cache policy -- 0 default, 1 non-temporal loads and stores, 2 loads only, 3 stores only
    constexpr bool NTL = NTP == 1 || NTP == 2, NT = NTP == 1 || NTP == 3;
    // LDS slot of logical position p (8-byte units): p ^ (rev4bits(p >> 8) << 1)
    __shared__ __attribute__((aligned(16))) float2 lds[4096];

    const uint32_t t = threadIdx.x;

    // ---- per-thread twiddles, identical for every transform: fetched once (fft.h:309 uses the
    // same single row exp(-+2 pi i j / N) of the table)
    float2 wA1[3], wA2[3], wB1[3], wB2[3];
    const uint32_t rr = t & 15, b = t >> 4;
    // `tw` is the plan's THREAD-TWIDDLE table (capi.hip: make_thread_twiddles), the same values as the
    // row W_4096^j laid out [value][thread] so that these are coalesced loads: gathering them from the row
    // (strides of 8..96 bytes per lane) cost about as many cache-line requests as the transform's data
#pragma unroll
    for (int r = 1; r < 4; r++) {
        wA1[r - 1] = tw[(r - 1) * 256 + t];        // W_4096^(r t)
        wA2[r - 1] = tw[(r + 2) * 256 + t];        // W_1024^(r t)  = W_4096^(4 r t)
        wB1[r - 1] = tw[1536 + (r - 1) * 16 + rr]; // W_256^(r rr)  = W_4096^(16 r rr)
        wB2[r - 1] = tw[1536 + (r + 2) * 16 + rr]; // W_64^(r rr)   = W_4096^(64 r rr)
    }

    // ---- LDS addressing (float2 units).  Loop-invariant address VECTORS are deliberately not kept
    // in registers (they cost ~40 VGPRs): pass B uses two bases + immediate offsets, passes A and C
    // rebuild theirs with one v_xor per access from a value the optimiser cannot hoist.
    // pass A writes p = t + 256 k: slot 256 k + (t ^ (rev4bits(k) << 1))
    // pass B (in place) p = 256 b + rr + 16 k: X = rev4bits(b) << 1 flips rr's bits 3..1 and k's bit 0,
    //   i.e. slot b_base + 16 (k ^ b_flip) = (b_base +- 16 b_flip) + 16 k for even / odd k
    const uint32_t xb = rev4bits(b) << 1;
    const uint32_t b_base = 256 * b + (rr ^ (xb & 15));
    const uint32_t b_flip = (xb >> 4) & 1;
    float2 *const lds_b_even = lds + b_base + 16 * b_flip;
    float2 *const lds_b_odd = lds + b_base - 16 * b_flip;
    // pass C reads p = 16 m + k, m = digit_reverse4(t) so that outputs land at t + 256 j
    const uint32_t m = ((t & 3) << 6) | (((t >> 2) & 3) << 4) | (((t >> 4) & 3) << 2) | (t >> 6);
    const uint32_t xc = rev4bits(m >> 4) << 1;
    const uint32_t c_base = 256 * (m >> 4) + 16 * ((m & 15) ^ (xc >> 4));
    const uint32_t c_x = (xc >> 1) & 7; // pair index i -> i ^ c_x

    float2 x[16], nx[PREFETCH ? 16 : 1];
    // PREFETCH == 2 ("pair"): the workgroup owns transforms 2b and 2b+1 (64 contiguous KiB), loads both up
    // front and runs them one after the other.
    constexpr uint64_t kStep = PREFETCH == 2 ? 1 : 0;
    uint64_t f = PREFETCH == 2 ? CHUNK * (uint64_t)blockIdx.x : blockIdx.x;
    const uint64_t f_end = PREFETCH == 2 ? (f + CHUNK < batch ? f + CHUNK : batch) : batch;
    if (PREFETCH && f < batch) {
        const float2 *src = data + f * 4096 + t;
#pragma unroll
        for (int k = 0; k < 16; k++)
            x[k] = gload<NTL>(src + 256 * k);
        if constexpr (PREFETCH == 2) {
            if (f + 1 < batch) {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    nx[k] = gload<NTL>(src + 4096 + 256 * k);
            }
        }
    }
    for (; f < f_end; f += (kStep ? kStep : gridDim.x)) {
        if constexpr (!PREFETCH) {
            const float2 *src = data + f * 4096 + t;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int k = (ORD == 1 || ORD == 2 || ORD == 4) ? row_order(ORD == 4 ? 2 : ORD, j) : j;
                x[k] = gload<NTL>(src + 256 * k);
            }
        }
        // ---- pass A: stages 0,1 (groups 1024, 256), fft.h:311-349 with i = 0,1
        two_stages<REV, true, true>(x, wA1, wA2);
        {
            uint32_t ta = t;
            asm volatile("" : "+v"(ta)); // keep the 16 xor'ed addresses out of loop-invariant registers
#pragma unroll
            for (int k = 0; k < 16; k++)
                lds[256 * k + (ta ^ ((__brev((uint32_t)k) >> 28) << 1))] = x[k];
        }
        __syncthreads();

        // prefetch the next transform; in flight during passes B and C
        const uint64_t fn = f + (kStep ? kStep : gridDim.x);
        if constexpr (PREFETCH == 1) {
            if (fn < batch) {
                const float2 *src = data + fn * 4096 + t;
#pragma unroll
                for (int k = 0; k < 16; k++)
                    nx[k] = gload<NTL>(src + 256 * k);
            }
        }

        // ---- pass B: stages 2,3 (groups 64, 16)
#pragma unroll
        for (int k = 0; k < 16; k++)
            x[k] = (k & 1) ? lds_b_odd[16 * k] : lds_b_even[16 * k];
        two_stages<REV, true, true>(x, wB1, wB2);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k & 1)
                lds_b_odd[16 * k] = x[k];
            else
                lds_b_even[16 * k] = x[k];
        }
        __syncthreads();

        // ---- pass C: stages 4,5 (groups 4, 1); only W_16 constants
        {
            uint32_t cx = c_x;
            asm volatile("" : "+v"(cx));
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float4 v = *reinterpret_cast<const float4 *>(&lds[c_base + 2 * (i ^ cx)]);
                x[2 * i] = float2{ v.x, v.y };
                x[2 * i + 1] = float2{ v.z, v.w };
            }
        }
        // every read of this transform is done: the tile may be overwritten -- by the NEXT transform of this
        // workgroup, so a workgroup on its last (usually only) transform skips the barrier (wave-uniform test)
        if (f + (kStep ? kStep : gridDim.x) < f_end || CHUNK == 1)
            __syncthreads();
        two_stages<REV, false, false>(x, wA1, wA2);
        if constexpr (CHUNK == 3)
            __syncthreads();

        // ---- store; register k = 4 d1 + d0 holds X[t + 256 * (4 d0 + d1)]: fft.h:351-355 folded
        float2 *dst = data + f * 4096 + t;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            // CHUNK == 1 (variant 12): issue the stores in ascending address order
            // ORD 1..3: the row sequence is row_order(j); register k = 4 (row & 3) + (row >> 2) holds that row
            constexpr int kOrdS = ORD == 3 ? 2 : ORD;
            const int rj = row_order(kOrdS, j);
            const int k = (ORD >= 1 && ORD <= 3) ? 4 * (rj & 3) + (rj >> 2) : (PREFETCH == 0 && CHUNK == 1) ? 4 * (j & 3) + (j >> 2) : j;
            float2 v = x[k];
            if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
                v.x *= scale;
                v.y *= scale;
            }
            gstore<NT>(dst + 256 * (4 * (k & 3) + (k >> 2)), v);
        }
        if constexpr (PREFETCH) {
            if (fn < batch) {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    x[k] = nx[k];
            }
        }
        if constexpr (PREFETCH == 2 && CHUNK > 2) {
            if (fn + 1 < f_end) { // keep one transform of loads in flight behind the one being computed
                const float2 *src = data + (fn + 1) * 4096 + t;
#pragma unroll
                for (int k = 0; k < 16; k++)
                    nx[k] = gload<NTL>(src + 256 * k);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused fast convolution: y = IFFT( FFT(x) .* H ) per transform in ONE kernel.
// The forward transform above leaves thread t holding X[t + 256 j], j < 16 -- which is exactly the
// layout its own first pass consumes -- so after the per-bin multiply the reverse transform (fft.h
// reverse_fft policy: conjugate twiddles, +i rotations, 1/N scale) runs on the same registers and the
// same LDS tile.  HBM sees one read and one write per element instead of three of each.

struct lds_map {
    float2 *b_even, *b_odd; // pass B bases (see sdsp_fft4096_r4_f32)
    uint32_t c_base, c_x, t;
};

// all six stages on one transform held as x[k] = element t + 256 k; returns with
// x[k] = result[t + 256 * (4 (k & 3) + (k >> 2))].  w*: FORWARD twiddles, conjugated here when REV
// (CONJ = false: the table is already folded for the direction).
template <bool REV, bool CONJ = REV>
__device__ __forceinline__ void fft4096_in_regs(float2 (&x)[16], float2 *lds, const lds_map &mp,
                                                const float2 (&wA1)[3], const float2 (&wA2)[3],
                                                const float2 (&wB1)[3], const float2 (&wB2)[3])
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
    {
        uint32_t ta = mp.t;
        asm volatile("" : "+v"(ta));
#pragma unroll
        for (int k = 0; k < 16; k++)
            lds[256 * k + (ta ^ ((__brev((uint32_t)k) >> 28) << 1))] = x[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++)
        x[k] = (k & 1) ? mp.b_odd[16 * k] : mp.b_even[16 * k];
    two_stages<REV, true, true>(x, b1, b2);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k & 1)
            mp.b_odd[16 * k] = x[k];
        else
            mp.b_even[16 * k] = x[k];
    }
    __syncthreads();
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
    __syncthreads(); // every read of the tile is done: the next transform may overwrite it
    two_stages<REV, false, false>(x, a1, a2);
}

template <bool NT>
__global__ __launch_bounds__(256, 2) void sdsp_fft4096_conv_f32(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                               const float2 *__restrict__ h, uint64_t batch)
{
    __shared__ __attribute__((aligned(16))) float2 lds[4096];
    const uint32_t t = threadIdx.x;
    float2 wA1[3], wA2[3], wB1[3], wB2[3];
    const uint32_t rr = t & 15, b = t >> 4;
#pragma unroll
    for (int r = 1; r < 4; r++) { // thread-twiddle table, see sdsp_fft4096_r4_f32
        wA1[r - 1] = tw[(r - 1) * 256 + t];
        wA2[r - 1] = tw[(r + 2) * 256 + t];
        wB1[r - 1] = tw[1536 + (r - 1) * 16 + rr];
        wB2[r - 1] = tw[1536 + (r + 2) * 16 + rr];
    }
    lds_map mp;
    const uint32_t xb = rev4bits(b) << 1;
    const uint32_t b_base = 256 * b + (rr ^ (xb & 15));
    const uint32_t b_flip = (xb >> 4) & 1;
    mp.b_even = lds + b_base + 16 * b_flip;
    mp.b_odd = lds + b_base - 16 * b_flip;
    const uint32_t m = ((t & 3) << 6) | (((t >> 2) & 3) << 4) | (((t >> 4) & 3) << 2) | (t >> 6);
    const uint32_t xc = rev4bits(m >> 4) << 1;
    mp.c_base = 256 * (m >> 4) + 16 * ((m & 15) ^ (xc >> 4));
    mp.c_x = (xc >> 1) & 7;
    mp.t = t;

    for (uint64_t f = blockIdx.x; f < batch; f += gridDim.x) {
        float2 x[16], z[16];
        const float2 *src = data + f * 4096 + t;
#pragma unroll
        for (int k = 0; k < 16; k++)
            x[k] = gload<NT>(src + 256 * k);
        fft4096_in_regs<false>(x, lds, mp, wA1, wA2, wB1, wB2);
        // x[k] = X[t + 256 j], j = 4 (k & 3) + (k >> 2): multiply by H[t + 256 j] and renumber so that
        // z[j] is element t + 256 j of the spectrum -- the input layout of the transform's first pass
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int j = 4 * (k & 3) + (k >> 2);
            z[j] = cmul(x[k], h[t + 256 * j]);
        }
        fft4096_in_regs<true>(z, lds, mp, wA1, wA2, wB1, wB2);
        float2 *dst = data + f * 4096 + t;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            float2 v = z[k];
            v.x *= 1.0f / 4096.0f; // reverse_fft::ScaleValues, fft.h:128-132
            v.y *= 1.0f / 4096.0f;
            gstore<NT>(dst + 256 * (4 * (k & 3) + (k >> 2)), v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sdsp::fft_radix2<T,4096> (fft.h:258-299) with the same machinery: twelve radix-2 DIF stages as three
// register passes of four stages (pair distances 2048..256, 128..16, 8..1), the same in-place LDS
// tile and XOR swizzle (the bank analysis carries over to the bit-reversed block assignment), four
// thread twiddles per pass instead of six.  The bit reversal (fft.h:269-273) is folded into the last
// pass's assignment: thread t takes block bit_reverse8(t), whose outputs land at t + 256*bit_reverse4(k).
// PAIR: the workgroup owns transforms 2b and 2b+1 (64 contiguous KiB), loads both up front and runs them
// one after the other (see sdsp_fft4096_r4_f32).
template <bool REV, bool NT, bool PAIR>
__global__ __launch_bounds__(256, 3) void sdsp_fft4096_r2_f32(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                              uint64_t batch, float scale)
{
    __shared__ __attribute__((aligned(16))) float2 lds[4096];
    const uint32_t t = threadIdx.x;
    const uint32_t rr = t & 15, b = t >> 4;
    float2 wA[4], wB[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { // thread-twiddle table (coalesced), see sdsp_fft4096_r4_f32
        wA[j] = tw[j * 256 + t];         // stage j of pass A: W_4096^(t 2^j)
        wB[j] = tw[1024 + j * 16 + rr];  // pass B: W_4096^(16 rr 2^j)
    }
    const uint32_t xb = rev4bits(b) << 1;
    const uint32_t b_base = 256 * b + (rr ^ (xb & 15));
    const uint32_t b_flip = (xb >> 4) & 1;
    float2 *const lds_b_even = lds + b_base + 16 * b_flip;
    float2 *const lds_b_odd = lds + b_base - 16 * b_flip;
    const uint32_t m = __brev(t) >> 24; // bit_reverse8(t)
    const uint32_t xc = rev4bits(m >> 4) << 1;
    const uint32_t c_base = 256 * (m >> 4) + 16 * ((m & 15) ^ (xc >> 4));
    const uint32_t c_x = (xc >> 1) & 7;

    float2 x[16], nx[PAIR ? 16 : 1];
    uint64_t f = PAIR ? 2 * (uint64_t)blockIdx.x : blockIdx.x;
    const uint64_t f_end = PAIR ? (f + 2 < batch ? f + 2 : batch) : batch;
    if constexpr (PAIR) {
        const float2 *src = data + f * 4096 + t;
        if (f < batch) {
#pragma unroll
            for (int k = 0; k < 16; k++)
                x[k] = gload<NT>(src + 256 * k);
        }
        if (f + 1 < batch) {
#pragma unroll
            for (int k = 0; k < 16; k++)
                nx[k] = gload<NT>(src + 4096 + 256 * k);
        }
    }
    for (; f < f_end; f += (PAIR ? 1 : gridDim.x)) {
        if constexpr (!PAIR) {
            const float2 *src = data + f * 4096 + t;
#pragma unroll
            for (int k = 0; k < 16; k++)
                x[k] = gload<NT>(src + 256 * k);
        }
        passes::r2_pass<REV, true, 0>::run(x, wA);
        {
            uint32_t ta = t;
            asm volatile("" : "+v"(ta));
#pragma unroll
            for (int k = 0; k < 16; k++)
                lds[256 * k + (ta ^ ((__brev((uint32_t)k) >> 28) << 1))] = x[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++)
            x[k] = (k & 1) ? lds_b_odd[16 * k] : lds_b_even[16 * k];
        passes::r2_pass<REV, true, 0>::run(x, wB);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k & 1)
                lds_b_odd[16 * k] = x[k];
            else
                lds_b_even[16 * k] = x[k];
        }
        __syncthreads();
        {
            uint32_t cx = c_x;
            asm volatile("" : "+v"(cx));
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float4 v = *reinterpret_cast<const float4 *>(&lds[c_base + 2 * (i ^ cx)]);
                x[2 * i] = float2{ v.x, v.y };
                x[2 * i + 1] = float2{ v.z, v.w };
            }
        }
        __syncthreads(); // also keeps the four waves' stores together: without it 77.3 % -> 73.7 %
        passes::r2_pass<REV, false, 0>::run(x, wA);
        float2 *dst = data + f * 4096 + t;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            float2 v = x[k];
            if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
                v.x *= scale;
                v.y *= scale;
            }
            gstore<NT>(dst + 256 * (int)(__brev((uint32_t)k) >> 28), v);
        }
        if constexpr (PAIR) {
#pragma unroll
            for (int k = 0; k < 16; k++)
                x[k] = nx[k];
        }
    }
}

int cu_count()
{
    static int cached = 0;
    if (cached)
        return cached;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return 256;
    cached = prop.multiProcessorCount;
    return cached;
}

template <int PREFETCH, int WAVES, int NT, int CHUNK = 2, int ORD = 0>
void launch_variant(const fft4096_args &a, uint64_t grid, hipStream_t s, uint32_t pad_lds = 0)
{
    float2 *d = reinterpret_cast<float2 *>(a.data);
    const float2 *w = reinterpret_cast<const float2 *>(a.tw);
    if (a.reverse)
        hipLaunchKernelGGL((sdsp_fft4096_r4_f32<true, PREFETCH, WAVES, NT, CHUNK, ORD>), dim3((uint32_t)grid), dim3(256), pad_lds, s, d,
                           w, a.batch, a.scale);
    else
        hipLaunchKernelGGL((sdsp_fft4096_r4_f32<false, PREFETCH, WAVES, NT, CHUNK, ORD>), dim3((uint32_t)grid), dim3(256), pad_lds, s, d,
                           w, a.batch, a.scale);
}

// All variants run the same arithmetic in the same order (bit-identical results); they differ in
// how HBM latency is hidden and in the cache policy of the streaming accesses.
struct variant_desc {
    bool prefetch; // register prefetch of the next transform (persistent grids only)
    int waves;     // launch bound: workgroups per CU the register allocator leaves room for
    int per_cu;    // persistent grid = per_cu x CUs workgroups; 0 = one workgroup per transform
    bool nt;       // non-temporal global accesses
};
constexpr variant_desc kVariants[] = {
    { false, 3, 0, true },  // 0 default: one workgroup per transform, nt; 100 VGPRs -> 4 workgroups per CU.
                            //   74.9 % of HBM peak (71.6 % before the thread-twiddle table made its twiddle loads coalesced,
                            //   74.2 % with a barrier after the last LDS read that only a looping workgroup needs)
    { false, 4, 0, true },  // 1 as 0 at 4 per CU: 20 B/lane of scratch cost 17 %
    { true, 3, 3, true },   // 2 persistent + register prefetch: 5.49 TB/s
    { true, 2, 2, true },   // 3
    { false, 3, 6, true },  // 4 persistent, no prefetch, 2x oversubscribed
    { false, 3, 0, false }, // 5 as 0 with the default cache policy
    { true, 3, 3, false },  // 6 as 2 with the default cache policy
    { true, 3, 0, true },   // 7 "pair": one workgroup per TWO consecutive transforms (64 KiB), both loaded up front, run
                            //   one after the other:
                            //   74.0 % while twiddles were gathered (it halved that cost), 71.4 % with the table
    { true, 2, 0, true },   // 8 as 7 at 2 per CU
    { true, 3, 0, true },   // 9 four consecutive transforms per workgroup, one transform of loads kept in flight: 69.2 %
    { true, 3, 0, true },   // 10 eight: 67.4 %
    { true, 4, 0, true },   // 11 as 7 at 4 per CU
    { false, 3, 0, true },  // 12 as 0 with the stores issued in ascending address order and the (redundant) barrier after
                            //    the last LDS read of a workgroup's only transform kept: 74.1 % where 0 gives 74.9 %
    { false, 3, 0, true },  // 13 as 0 with a barrier right before the stores (keeps the four waves' stores together): 74.8 %
                            //    (squeezing 0 to 96 VGPRs for a fifth workgroup per CU cost 24 B/lane of scratch: 67.7 %)
    { false, 3, 0, true },  // 14 as 0 with 8 KiB of unused dynamic LDS: caps the CU at 4 workgroups
    { false, 3, 0, true },  // 15 as 0 with 21 KiB: caps it at 3 (0 / 14 / 15: 74.9 / 75.1 / 75.1 % -- occupancy is not the limiter;
                            //    a pair kernel that also delays the first transform's stores to make them one 64-KiB burst: 69.5 %)
    { false, 3, 0, true },  // 16 capped at 2 workgroups per CU
    { false, 3, 0, true },  // 17 capped at 1 (2 / 1 per CU: 74.7 / 46.4 %: two workgroups per CU already reach the plateau)
    { false, 3, 0, true },  // 18 non-temporal loads, default-policy stores: 70.9 %
    { false, 3, 0, true },  // 19 default-policy loads, non-temporal stores: 69.1 % (5, both default: 69.0 %; 0, both nt: 74.8 %)
    { false, 3, 0, true },  // 20 row order: loads and stores bit-reversed: 75.3 % where 0 gives 74.6 % in the same run
    { false, 3, 0, true },  // 21 loads and stores (5j+3) mod 16: 74.4 %
    { false, 3, 0, true },  // 22 loads ascending, stores (5j+3) mod 16: 74.9 %
    { false, 3, 0, true },  // 23 loads (5j+3) mod 16, stores as 0: 74.2 % -- the order of a thread's rows is worth < 1 point
};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
} // namespace

int launch_fft4096_r2_f32(const fft4096_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    if (a.batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float2 *d = reinterpret_cast<float2 *>(a.data);
    const float2 *w = reinterpret_cast<const float2 *>(a.tw);
    const dim3 pair_grid((uint32_t)((a.batch + 1) / 2)), grid((uint32_t)a.batch);
    if (a.pair) {
        if (a.reverse)
            hipLaunchKernelGGL((sdsp_fft4096_r2_f32<true, true, true>), pair_grid, dim3(256), 0, s, d, w, a.batch, a.scale);
        else
            hipLaunchKernelGGL((sdsp_fft4096_r2_f32<false, true, true>), pair_grid, dim3(256), 0, s, d, w, a.batch, a.scale);
    } else if (a.reverse)
        hipLaunchKernelGGL((sdsp_fft4096_r2_f32<true, true, false>), grid, dim3(256), 0, s, d, w, a.batch, a.scale);
    else
        hipLaunchKernelGGL((sdsp_fft4096_r2_f32<false, true, false>), grid, dim3(256), 0, s, d, w, a.batch, a.scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft4096 r2 launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

int launch_fft4096_conv_f32(void *data, const void *tw, const void *h, uint64_t batch, void *stream)
{
    if (batch == 0)
        return SDSP_HIP_OK;
    if (batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL((sdsp_fft4096_conv_f32<true>), dim3((uint32_t)batch), dim3(256), 0, s,
                       reinterpret_cast<float2 *>(data), reinterpret_cast<const float2 *>(tw),
                       reinterpret_cast<const float2 *>(h), batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft4096 conv launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

const char *fft4096_kernel_name(int variant)
{
    (void)variant;
    return "sdsp_fft4096_r4_f32";
}

int fft4096_num_variants() { return kNumVariants; }

int launch_fft4096_r4_f32(const fft4096_args &a, int variant, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    if (variant < 0 || variant >= kNumVariants)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "unknown fft4096 variant");
    const variant_desc v = kVariants[variant];
    uint64_t grid = v.per_cu ? (uint64_t)cu_count() * v.per_cu : a.batch;
    if (grid > a.batch)
        grid = a.batch;
    if (grid > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (variant) {
    case 0: launch_variant<0, 3, true>(a, grid, s); break;
    case 1: launch_variant<0, 4, true>(a, grid, s); break;
    case 2: launch_variant<1, 3, true>(a, grid, s); break;
    case 3: launch_variant<1, 2, true>(a, grid, s); break;
    case 4: launch_variant<0, 3, true>(a, grid, s); break;
    case 5: launch_variant<0, 3, false>(a, grid, s); break;
    case 6: launch_variant<1, 3, false>(a, grid, s); break;
    case 7: launch_variant<2, 3, true>(a, (a.batch + 1) / 2, s); break;
    case 8: launch_variant<2, 2, true>(a, (a.batch + 1) / 2, s); break;
    case 9: launch_variant<2, 3, true, 4>(a, (a.batch + 3) / 4, s); break;
    case 10: launch_variant<2, 3, true, 8>(a, (a.batch + 7) / 8, s); break;
    case 11: launch_variant<2, 4, true, 2>(a, (a.batch + 1) / 2, s); break;
    case 12: launch_variant<0, 3, true, 1>(a, grid, s); break;
    case 13: launch_variant<0, 3, true, 3>(a, grid, s); break;
    case 14: launch_variant<0, 3, true>(a, grid, s, 8 * 1024); break;  // 40 KiB of LDS: at most 4 workgroups per CU
    case 15: launch_variant<0, 3, true>(a, grid, s, 21 * 1024); break; // 53 KiB: at most 3
    case 16: launch_variant<0, 3, true>(a, grid, s, 48 * 1024); break; // 80 KiB: at most 2
    case 17: launch_variant<0, 3, true>(a, grid, s, 64 * 1024 - 256); break; // 96 KiB: 1
    case 18: launch_variant<0, 3, 2>(a, grid, s); break; // nt loads, default-policy stores
    case 19: launch_variant<0, 3, 3>(a, grid, s); break; // default-policy loads, nt stores
    case 20: launch_variant<0, 3, 1, 2, 1>(a, grid, s); break;
    case 21: launch_variant<0, 3, 1, 2, 2>(a, grid, s); break;
    case 22: launch_variant<0, 3, 1, 2, 3>(a, grid, s); break;
    default: launch_variant<0, 3, 1, 2, 4>(a, grid, s); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft4096 launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace sdsp_hip

// fft32_r4.h -- radix-4 DIF stages (sdsp::fft_radix4, fft.h:311-349) on the 32 registers of fft_big.hip's threads.
//
// The kernel's dataflow is the in-place binary one: "layer" l pairs the registers that differ in one register bit.  A radix-4
// stage is two layers with the reference's twiddle placement: after the first layer the quarter (b1, b0) = (1, 1) is rotated
// by -+i (temp2_timesi / temp4_timesi, fft.h:337-338); after the second the quarters (0,1), (1,0), (1,1) are multiplied by
// W_G^(2n), W_G^(n), W_G^(3n) -- the coefficients the reference applies when the next stage loads its inputs (fft.h:322-335),
// G = N / 4^s, n = position mod G/4.  Each factor = a per-thread table value x a compile-time constant W_64^e.  Against the
// reference's digit order the quarters (0,1) and (1,0) sit swapped, which only changes the final permutation: the result
// comes out BIT-reversed like the radix-2 stages' and takes the same store.  tools/model_fft_big_r4.py replays all of it
// (layers, rotations, every multiplier as table value x constant) against numpy.fft.
#pragma once

#include <utility>

#include "fft32.h"

namespace sdsp_hip
{
namespace fft32
{
// cos / sin of 2 pi e / 64
__device__ constexpr float kC64[64] = { 1.0f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f, 0.0f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f, -1.0f, -0.995184727f, -0.98078528f, -0.956940336f, -0.923879533f, -0.881921264f, -0.831469612f, -0.773010453f, -0.707106781f, -0.634393284f, -0.555570233f, -0.471396737f, -0.382683432f, -0.290284677f, -0.195090322f, -0.0980171403f, 0.0f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f, 0.707106781f, 0.773010453f, 0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f };
__device__ constexpr float kS64[64] = { 0.0f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f, 0.707106781f, 0.773010453f, 0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f, 1.0f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f, 0.0f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f, -1.0f, -0.995184727f, -0.98078528f, -0.956940336f, -0.923879533f, -0.881921264f, -0.831469612f, -0.773010453f, -0.707106781f, -0.634393284f, -0.555570233f, -0.471396737f, -0.382683432f, -0.290284677f, -0.195090322f, -0.0980171403f };

template <bool REV> __device__ __forceinline__ float2 rot_i(float2 a) // times -i (forward) / +i (reverse)
{
    return REV ? float2{ -a.y, a.x } : float2{ a.y, -a.x };
}
// a * W_64^E (forward) / its conjugate (reverse)
template <bool REV, int E> __device__ __forceinline__ float2 mul_w64(float2 a)
{
    constexpr int e = E & 63;
    if constexpr (e == 0) {
        return a;
    } else if constexpr (e == 16) {
        return rot_i<REV>(a);
    } else if constexpr (e == 32) {
        return float2{ -a.x, -a.y };
    } else if constexpr (e == 48) {
        return rot_i<!REV>(a);
    } else {
        constexpr float c = kC64[e], s = REV ? kS64[e] : -kS64[e];
        return float2{ a.x * c - a.y * s, a.x * s + a.y * c };
    }
}

// one layer: butterflies on the register pairs (k, k + H)
template <int H> __device__ __forceinline__ void layer(float2 (&x)[32])
{
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if ((k & H) != 0)
            continue;
        const float2 a = x[k], b = x[k + H];
        x[k] = a + b;
        x[k + H] = a - b;
    }
}

// both layers of a stage whose quarter bits are the register bits B1, B1 - 1
template <bool REV, int B1> __device__ __forceinline__ void r4_layers(float2 (&x)[32])
{
    constexpr int H1 = 1 << B1, H0 = H1 >> 1;
    layer<H1>(x);
#pragma unroll
    for (int k = 0; k < 32; k++)
        if ((k & H1) != 0 && (k & H0) != 0)
            x[k] = rot_i<REV>(x[k]);
    layer<H0>(x);
}

// the stage's output twiddles: register K in quarter q' owes W_G^(q n); its constant part is W_64^(q * (K & LOWMASK) * EUNIT)
// and its thread part thr[q - 1] (when THR)
template <bool REV, int B1, int LOWMASK, int EUNIT, bool THR, int K>
__device__ __forceinline__ void r4_twiddle_one(float2 (&x)[32], const float2 (&thr)[3])
{
    constexpr int b1 = (K >> B1) & 1, b0 = (K >> (B1 - 1)) & 1;
    constexpr int q = b1 ? (b0 ? 3 : 1) : (b0 ? 2 : 0);
    if constexpr (q != 0) {
        float2 v = mul_w64<REV, q * (K & LOWMASK) * EUNIT>(x[K]);
        if constexpr (THR)
            v = cmul(v, thr[q - 1]);
        x[K] = v;
    }
}
template <bool REV, int B1, int LOWMASK, int EUNIT, bool THR, int... Ks>
__device__ __forceinline__ void r4_twiddles(float2 (&x)[32], const float2 (&thr)[3], std::integer_sequence<int, Ks...>)
{
    (r4_twiddle_one<REV, B1, LOWMASK, EUNIT, THR, Ks>(x, thr), ...);
}
template <bool REV, int B1, int LOWMASK, int EUNIT, bool THR>
__device__ __forceinline__ void r4_stage(float2 (&x)[32], const float2 (&thr)[3])
{
    r4_layers<REV, B1>(x);
    r4_twiddles<REV, B1, LOWMASK, EUNIT, THR>(x, thr, std::make_integer_sequence<int, 32>{});
}

// Stage 2 of N = 16384 straddles the first exchange.  Second half (pass B): the layer on register bit 4, then register J owes
// W_1024^(q n), n = v + 16 (J & 15), with q = (thread's block odd ? 1 : 0) for J < 16 and (odd ? 3 : 2) for J >= 16: the thread part
// comes from the table (two values per thread), the constant W_64^(q (J & 15)) is one of two literals picked by the parity.
template <bool REV, int J> __device__ __forceinline__ void r4_split_twiddle_one(float2 (&x)[32], bool odd, float2 thr_lo, float2 thr_hi)
{
    constexpr int jj = J & 15;
    constexpr int qe = J < 16 ? 0 : 2, qo = J < 16 ? 1 : 3;
    float2 v = x[J];
    if constexpr (jj != 0) {
        constexpr int ee = (qe * jj) & 63, eo = (qo * jj) & 63;
        const float c = odd ? kC64[eo] : kC64[ee];
        const float sf = odd ? kS64[eo] : kS64[ee];
        const float s = REV ? sf : -sf;
        v = float2{ v.x * c - v.y * s, v.x * s + v.y * c };
    }
    x[J] = cmul(v, J < 16 ? thr_lo : thr_hi);
}
template <bool REV, int... Js>
__device__ __forceinline__ void r4_split_twiddles(float2 (&x)[32], bool odd, float2 thr_lo, float2 thr_hi, std::integer_sequence<int, Js...>)
{
    (r4_split_twiddle_one<REV, Js>(x, odd, thr_lo, thr_hi), ...);
}

// ---- the same layers on double2 (fft_big64.hip's radix-4 form, round 3).  Kept as overloads, not as templates over the complex
// type: templating the float2 functions changed the f32 kernels' register allocation (N = 16384 radix-4 fused convolution:
// 68 -> 180 bytes of scratch), and the f32 code is the tuned one.
__device__ constexpr double kC64d[64] = { 1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867, 0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476, 0.6343932841636455, 0.5555702330196023, 0.4713967368259978, 0.38268343236508984, 0.29028467725446233, 0.19509032201612833, 0.09801714032956077, 0.0, -0.09801714032956065, -0.1950903220161282, -0.29028467725446216, -0.3826834323650897, -0.4713967368259977, -0.555570233019602, -0.6343932841636454, -0.7071067811865475, -0.773010453362737, -0.8314696123025453, -0.8819212643483549, -0.9238795325112867, -0.9569403357322088, -0.9807852804032304, -0.9951847266721968, -1.0, -0.9951847266721969, -0.9807852804032304, -0.9569403357322089, -0.9238795325112868, -0.881921264348355, -0.8314696123025455, -0.7730104533627371, -0.7071067811865477, -0.6343932841636459, -0.5555702330196022, -0.47139673682599786, -0.38268343236509034, -0.29028467725446244, -0.19509032201612866, -0.09801714032956045, 0.0, 0.09801714032956009, 0.1950903220161283, 0.29028467725446205, 0.38268343236509, 0.4713967368259976, 0.5555702330196018, 0.6343932841636456, 0.7071067811865474, 0.7730104533627367, 0.8314696123025452, 0.8819212643483548, 0.9238795325112865, 0.9569403357322088, 0.9807852804032303, 0.9951847266721969 };
__device__ constexpr double kS64d[64] = { 0.0, 0.0980171403295606, 0.19509032201612825, 0.29028467725446233, 0.3826834323650898, 0.47139673682599764, 0.5555702330196022, 0.6343932841636455, 0.7071067811865475, 0.773010453362737, 0.8314696123025452, 0.8819212643483549, 0.9238795325112867, 0.9569403357322089, 0.9807852804032304, 0.9951847266721968, 1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322089, 0.9238795325112867, 0.881921264348355, 0.8314696123025455, 0.7730104533627371, 0.7071067811865476, 0.6343932841636455, 0.5555702330196022, 0.47139673682599786, 0.3826834323650899, 0.2902846772544624, 0.1950903220161286, 0.09801714032956083, 0.0, -0.09801714032956059, -0.19509032201612836, -0.2902846772544621, -0.38268343236508967, -0.47139673682599764, -0.555570233019602, -0.6343932841636453, -0.7071067811865475, -0.7730104533627367, -0.8314696123025452, -0.8819212643483549, -0.9238795325112865, -0.9569403357322088, -0.9807852804032303, -0.9951847266721969, -1.0, -0.9951847266721969, -0.9807852804032304, -0.9569403357322089, -0.9238795325112866, -0.881921264348355, -0.8314696123025455, -0.7730104533627369, -0.7071067811865477, -0.6343932841636459, -0.5555702330196022, -0.4713967368259979, -0.3826834323650904, -0.2902846772544625, -0.19509032201612872, -0.0980171403295605 };

template <bool REV> __device__ __forceinline__ double2 rot_i(double2 a) // times -i (forward) / +i (reverse)
{
    return REV ? double2{ -a.y, a.x } : double2{ a.y, -a.x };
}
// a * W_64^E (forward) / its conjugate (reverse)
template <bool REV, int E> __device__ __forceinline__ double2 mul_w64(double2 a)
{
    constexpr int e = E & 63;
    if constexpr (e == 0) {
        return a;
    } else if constexpr (e == 16) {
        return rot_i<REV>(a);
    } else if constexpr (e == 32) {
        return double2{ -a.x, -a.y };
    } else if constexpr (e == 48) {
        return rot_i<!REV>(a);
    } else {
        constexpr double c = kC64d[e], s = REV ? kS64d[e] : -kS64d[e];
        return double2{ a.x * c - a.y * s, a.x * s + a.y * c };
    }
}

// one layer: butterflies on the register pairs (k, k + H)
template <int H> __device__ __forceinline__ void layer(double2 (&x)[32])
{
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if ((k & H) != 0)
            continue;
        const double2 a = x[k], b = x[k + H];
        x[k] = a + b;
        x[k + H] = a - b;
    }
}

// both layers of a stage whose quarter bits are the register bits B1, B1 - 1
template <bool REV, int B1> __device__ __forceinline__ void r4_layers(double2 (&x)[32])
{
    constexpr int H1 = 1 << B1, H0 = H1 >> 1;
    layer<H1>(x);
#pragma unroll
    for (int k = 0; k < 32; k++)
        if ((k & H1) != 0 && (k & H0) != 0)
            x[k] = rot_i<REV>(x[k]);
    layer<H0>(x);
}

// the stage's output twiddles: register K in quarter q' owes W_G^(q n); its constant part is W_64^(q * (K & LOWMASK) * EUNIT)
// and its thread part thr[q - 1] (when THR)
template <bool REV, int B1, int LOWMASK, int EUNIT, bool THR, int K>
__device__ __forceinline__ void r4_twiddle_one(double2 (&x)[32], const double2 (&thr)[3])
{
    constexpr int b1 = (K >> B1) & 1, b0 = (K >> (B1 - 1)) & 1;
    constexpr int q = b1 ? (b0 ? 3 : 1) : (b0 ? 2 : 0);
    if constexpr (q != 0) {
        double2 v = mul_w64<REV, q * (K & LOWMASK) * EUNIT>(x[K]);
        if constexpr (THR)
            v = cmul(v, thr[q - 1]);
        x[K] = v;
    }
}
template <bool REV, int B1, int LOWMASK, int EUNIT, bool THR, int... Ks>
__device__ __forceinline__ void r4_twiddles(double2 (&x)[32], const double2 (&thr)[3], std::integer_sequence<int, Ks...>)
{
    (r4_twiddle_one<REV, B1, LOWMASK, EUNIT, THR, Ks>(x, thr), ...);
}
template <bool REV, int B1, int LOWMASK, int EUNIT, bool THR>
__device__ __forceinline__ void r4_stage(double2 (&x)[32], const double2 (&thr)[3])
{
    r4_layers<REV, B1>(x);
    r4_twiddles<REV, B1, LOWMASK, EUNIT, THR>(x, thr, std::make_integer_sequence<int, 32>{});
}

// Stage 2 of N = 16384 straddles the first exchange.  Second half (pass B): the layer on register bit 4, then register J owes
// W_1024^(q n), n = v + 16 (J & 15), with q = (thread's block odd ? 1 : 0) for J < 16 and (odd ? 3 : 2) for J >= 16: the thread part
// comes from the table (two values per thread), the constant W_64^(q (J & 15)) is one of two literals picked by the parity.
template <bool REV, int J> __device__ __forceinline__ void r4_split_twiddle_one(double2 (&x)[32], bool odd, double2 thr_lo, double2 thr_hi)
{
    constexpr int jj = J & 15;
    constexpr int qe = J < 16 ? 0 : 2, qo = J < 16 ? 1 : 3;
    double2 v = x[J];
    if constexpr (jj != 0) {
        constexpr int ee = (qe * jj) & 63, eo = (qo * jj) & 63;
        const double c = odd ? kC64d[eo] : kC64d[ee];
        const double sf = odd ? kS64d[eo] : kS64d[ee];
        const double s = REV ? sf : -sf;
        v = double2{ v.x * c - v.y * s, v.x * s + v.y * c };
    }
    x[J] = cmul(v, J < 16 ? thr_lo : thr_hi);
}
template <bool REV, int... Js>
__device__ __forceinline__ void r4_split_twiddles(double2 (&x)[32], bool odd, double2 thr_lo, double2 thr_hi, std::integer_sequence<int, Js...>)
{
    (r4_split_twiddle_one<REV, Js>(x, odd, thr_lo, thr_hi), ...);
}

} // namespace fft32
} // namespace sdsp_hip

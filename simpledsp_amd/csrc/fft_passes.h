// fft_passes.h -- register-pass building blocks shared by the FFT kernels (device code, f32).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace sdsp_hip
{
namespace passes
{
constexpr float kC1 = 0.92387953251128673848f; // cos(pi/8)
constexpr float kS1 = 0.38268343236508978178f; // sin(pi/8)
constexpr float kH = 0.70710678118654752440f;  // sqrt(1/2)

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return float2{ a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return float2{ a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return float2{ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x };
}
template <bool REV> __device__ __forceinline__ float2 rot90(float2 a) // * -i (forward) / +i (reverse)
{
    return REV ? float2{ -a.y, a.x } : float2{ a.y, -a.x };
}
// a * W_16^E, E in [0, 16), compile-time
template <bool REV, int E> __device__ __forceinline__ float2 mul_w16(float2 a)
{
    static_assert(E >= 0 && E < 16, "W_16 exponent");
    if constexpr (E == 0) {
        return a;
    } else if constexpr (E == 4) {
        return rot90<REV>(a);
    } else if constexpr (E == 8) {
        return float2{ -a.x, -a.y };
    } else if constexpr (E == 12) {
        return rot90<!REV>(a);
    } else {
        constexpr float c[16] = { 1.f, kC1, kH, kS1, 0.f, -kS1, -kH, -kC1, -1.f, -kC1, -kH, -kS1, 0.f, kS1, kH, kC1 };
        constexpr float s[16] = { 0.f, kS1, kH, kC1, 1.f, kC1, kH, kS1, 0.f, -kS1, -kH, -kC1, -1.f, -kC1, -kH, -kS1 };
        const float cr = c[E], ci = REV ? s[E] : -s[E]; // exp(-+ 2 pi i E / 16)
        return float2{ a.x * cr - a.y * ci, a.x * ci + a.y * cr };
    }
}

// Radix-2: stages J0..3 of the 4-stage DIF network on x[16]; stage j pairs (k, k + (8 >> j)) -- the
// butterfly of fft.h:286-291 in decimation-in-frequency form.  The factor the lower output owes
// splits into the thread twiddle w[j] (ignored when !TW) and the constant W_16^((k mod h) << j).
template <bool REV, bool TW, int J0> struct r2_pass {
    template <int J, int K> static __device__ __forceinline__ void bfly(float2 (&x)[16], const float2 (&w)[4])
    {
        constexpr int h = 8 >> J;
        if constexpr ((K & h) == 0) {
            const float2 a = x[K], b = x[K + h];
            x[K] = cadd(a, b);
            float2 d = mul_w16<REV, ((K & (h - 1)) << J) & 15>(csub(a, b));
            if constexpr (TW)
                d = cmul(d, w[J]);
            x[K + h] = d;
        }
    }
    template <int J, int... Ks>
    static __device__ __forceinline__ void stage(float2 (&x)[16], const float2 (&w)[4], std::integer_sequence<int, Ks...>)
    {
        (bfly<J, Ks>(x, w), ...);
    }
    static __device__ __forceinline__ void run(float2 (&x)[16], const float2 (&w)[4])
    {
        using seq = std::make_integer_sequence<int, 16>;
        if constexpr (J0 <= 0)
            stage<0>(x, w, seq{});
        if constexpr (J0 <= 1)
            stage<1>(x, w, seq{});
        if constexpr (J0 <= 2)
            stage<2>(x, w, seq{});
        stage<3>(x, w, seq{});
    }
};
} // namespace passes
} // namespace sdsp_hip

// fft_passes.h -- register-pass building blocks shared by the FFT kernels (device code).
// Generic over the complex vector type C (float2 / double2).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace sdsp_hip
{
namespace passes
{
constexpr double kC1 = 0.92387953251128673848; // cos(pi/8)
constexpr double kS1 = 0.38268343236508978178; // sin(pi/8)
constexpr double kH = 0.70710678118654752440;  // sqrt(1/2)

template <typename C> __device__ __forceinline__ C cadd(C a, C b) { return C{ a.x + b.x, a.y + b.y }; }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { return C{ a.x - b.x, a.y - b.y }; }
template <typename C> __device__ __forceinline__ C cmul(C a, C b)
{
    return C{ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x };
}
template <bool REV, typename C> __device__ __forceinline__ C rot90(C a) // * -i (forward) / +i (reverse)
{
    return REV ? C{ -a.y, a.x } : C{ a.y, -a.x };
}
// a * W_16^E, E in [0, 16), compile-time
template <bool REV, int E, typename C> __device__ __forceinline__ C mul_w16(C a)
{
    using R = decltype(a.x);
    static_assert(E >= 0 && E < 16, "W_16 exponent");
    if constexpr (E == 0) {
        return a;
    } else if constexpr (E == 4) {
        return rot90<REV>(a);
    } else if constexpr (E == 8) {
        return C{ -a.x, -a.y };
    } else if constexpr (E == 12) {
        return rot90<!REV>(a);
    } else {
        constexpr double c[16] = { 1., kC1, kH, kS1, 0., -kS1, -kH, -kC1, -1., -kC1, -kH, -kS1, 0., kS1, kH, kC1 };
        constexpr double s[16] = { 0., kS1, kH, kC1, 1., kC1, kH, kS1, 0., -kS1, -kH, -kC1, -1., -kC1, -kH, -kS1 };
        const R cr = (R)c[E], ci = (R)(REV ? s[E] : -s[E]); // exp(-+ 2 pi i E / 16)
        return C{ a.x * cr - a.y * ci, a.x * ci + a.y * cr };
    }
}

// Radix-2: stages J0..3 of the 4-stage DIF network on x[16]; stage j pairs (k, k + (8 >> j)) -- the
// butterfly of fft.h:286-291 in decimation-in-frequency form.  The factor the lower output owes
// splits into the thread twiddle w[j] (ignored when !TW) and the constant W_16^((k mod h) << j).
template <bool REV, bool TW, int J0> struct r2_pass {
    template <int J, int K, typename C> static __device__ __forceinline__ void bfly(C (&x)[16], const C (&w)[4])
    {
        constexpr int h = 8 >> J;
        if constexpr ((K & h) == 0) {
            const C a = x[K], b = x[K + h];
            x[K] = cadd(a, b);
            C d = mul_w16<REV, ((K & (h - 1)) << J) & 15>(csub(a, b));
            if constexpr (TW)
                d = cmul(d, w[J]);
            x[K + h] = d;
        }
    }
    template <int J, typename C, int... Ks>
    static __device__ __forceinline__ void stage(C (&x)[16], const C (&w)[4], std::integer_sequence<int, Ks...>)
    {
        (bfly<J, Ks>(x, w), ...);
    }
    template <typename C> static __device__ __forceinline__ void run(C (&x)[16], const C (&w)[4])
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

// Radix-4 (fft.h:342-345): stage X pairs k = j + 4q over q (offset 4), stage Y pairs 4q + q' over q'
// (offset 1).  w1[q-1]: thread twiddle of stage X's output q; w2[q'-1]: of stage Y's output q'.
template <bool REV, typename C> __device__ __forceinline__ void bfly4(C &a, C &b, C &c, C &d)
{
    const C t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = rot90<REV>(csub(b, d));
    a = cadd(t0, t2);
    b = cadd(t1, t3);
    c = csub(t0, t2);
    d = csub(t1, t3);
}
template <bool REV, bool TW, bool BOTH, typename C>
__device__ __forceinline__ void r4_pass(C (&x)[16], const C (&w1)[3], const C (&w2)[3])
{
    if constexpr (BOTH) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            bfly4<REV>(x[j], x[j + 4], x[j + 8], x[j + 12]);
        x[5] = mul_w16<REV, 1>(x[5]);
        x[6] = mul_w16<REV, 2>(x[6]);
        x[7] = mul_w16<REV, 3>(x[7]);
        x[9] = mul_w16<REV, 2>(x[9]);
        x[10] = mul_w16<REV, 4>(x[10]);
        x[11] = mul_w16<REV, 6>(x[11]);
        x[13] = mul_w16<REV, 3>(x[13]);
        x[14] = mul_w16<REV, 6>(x[14]);
        x[15] = mul_w16<REV, 9>(x[15]);
        if constexpr (TW) {
#pragma unroll
            for (int q = 1; q < 4; q++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    x[j + 4 * q] = cmul(x[j + 4 * q], w1[q - 1]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        bfly4<REV>(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
        if constexpr (TW) {
            x[4 * q + 1] = cmul(x[4 * q + 1], w2[0]);
            x[4 * q + 2] = cmul(x[4 * q + 2], w2[1]);
            x[4 * q + 3] = cmul(x[4 * q + 3], w2[2]);
        }
    }
}
} // namespace passes
} // namespace sdsp_hip

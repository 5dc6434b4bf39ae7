// fft32.h -- 32-points-per-thread building blocks shared by the large-transform kernels
// (fft1m.hip: N = 2^20 four-step; fft_big.hip: N = 8192 .. 32768 in one HBM pass): streaming
// loads/stores, saddr-form addressing, and five radix-2 DIF stages (fft.h:276-294's butterflies in
// decimation-in-frequency order) on 32 registers.
#pragma once

#include <hip/hip_runtime.h>

namespace sdsp_hip
{
namespace fft32
{
typedef float v2f_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 nt_load(const float2 *p)
{
    const v2f_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(p));
    return float2{ v.x, v.y };
}
__device__ __forceinline__ void nt_store(float2 *p, float2 a)
{
    const v2f_t v = { a.x, a.y };
    __builtin_nontemporal_store(v, reinterpret_cast<v2f_t *>(p));
}
// uniform base (SGPR pair) + 32-bit per-thread byte offset: the global_load/store "saddr" form, one
// VGPR of address instead of a 64-bit pair per access
__device__ __forceinline__ const float2 *at(const float2 *base, uint32_t byte_off)
{
    return reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ float2 *at(float2 *base, uint32_t byte_off)
{
    return reinterpret_cast<float2 *>(reinterpret_cast<char *>(base) + byte_off);
}
// the same helpers for complex doubles (the two-pass kernels in double, fft_2pass.hip)
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load(const double2 *p)
{
    const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(p));
    return double2{ v.x, v.y };
}
__device__ __forceinline__ void nt_store(double2 *p, double2 a)
{
    const v2d_t v = { a.x, a.y };
    __builtin_nontemporal_store(v, reinterpret_cast<v2d_t *>(p));
}
__device__ __forceinline__ const double2 *at(const double2 *base, uint32_t byte_off)
{
    return reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ double2 *at(double2 *base, uint32_t byte_off)
{
    return reinterpret_cast<double2 *>(reinterpret_cast<char *>(base) + byte_off);
}
// Rows of one transform through a buffer resource: `buffer_load/store_dwordx2 v, voffset, s[rsrc], soffset offen` takes the
// row's byte offset in an SGPR (any size) next to ONE per-thread VGPR offset.  With rows 4-8 KiB apart -- beyond the 13-bit
// immediate of global_load -- the compiler builds a 64-bit VGPR address per row from plain pointers however they are written
// (it reassociates to (base + thread) + row); at N = 32768 it then spilled 18 of those pairs and reloaded each in front of its
// store behind an s_waitcnt vmcnt(0), which serialised the store phase.
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rows(const float2 *base, uint32_t bytes)
{
    // word 3 = 0x00020000: DATA_FORMAT 32 (raw dword access), no swizzle, no index stride -- the gfx9 / CDNA raw-buffer setting
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(base), 0, (int)bytes, 0x00020000);
}
template <bool NT> __device__ __forceinline__ float2 row_load(__amdgpu_buffer_rsrc_t rows, uint32_t thread_off, uint32_t row_off)
{
    const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(rows, thread_off, row_off, NT ? 2 : 0); // aux bit 1 = nt
    const unsigned int lo = v.x, hi = v.y; // (__builtin_bit_cast applied to v.y directly reads element 0: seen with ROCm 7.2's clang)
    return float2{ __uint_as_float(lo), __uint_as_float(hi) };
}
template <bool NT> __device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t rows, uint32_t thread_off, uint32_t row_off, float2 a)
{
    const v2u_t v = { __float_as_uint(a.x), __float_as_uint(a.y) };
    __builtin_amdgcn_raw_buffer_store_b64(v, rows, thread_off, row_off, NT ? 2 : 0);
}
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
typedef float v4f32_t __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void row_store16(__amdgpu_buffer_rsrc_t rows, uint32_t thread_off, uint32_t row_off, v4f32_t a)
{
    const v4u_t v = { __float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w) };
    __builtin_amdgcn_raw_buffer_store_b128(v, rows, thread_off, row_off, NT ? 2 : 0);
}
__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return float2{ a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return float2{ a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return float2{ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x };
}
__device__ __forceinline__ double2 operator+(double2 a, double2 b) { return double2{ a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ double2 operator-(double2 a, double2 b) { return double2{ a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return double2{ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x };
}

// cos / sin of 2*pi*j/32, j < 16
__device__ constexpr float kC32[16] = { 1.0f,
                                        0.98078528040323044913f,
                                        0.92387953251128675613f,
                                        0.83146961230254523708f,
                                        0.70710678118654752440f,
                                        0.55557023301960222474f,
                                        0.38268343236508977173f,
                                        0.19509032201612826785f,
                                        0.0f,
                                        -0.19509032201612826785f,
                                        -0.38268343236508977173f,
                                        -0.55557023301960222474f,
                                        -0.70710678118654752440f,
                                        -0.83146961230254523708f,
                                        -0.92387953251128675613f,
                                        -0.98078528040323044913f };
__device__ constexpr float kS32[16] = { 0.0f,
                                        0.19509032201612826785f,
                                        0.38268343236508977173f,
                                        0.55557023301960222474f,
                                        0.70710678118654752440f,
                                        0.83146961230254523708f,
                                        0.92387953251128675613f,
                                        0.98078528040323044913f,
                                        1.0f,
                                        0.98078528040323044913f,
                                        0.92387953251128675613f,
                                        0.83146961230254523708f,
                                        0.70710678118654752440f,
                                        0.55557023301960222474f,
                                        0.38268343236508977173f,
                                        0.19509032201612826785f };

// the same table in double (correctly rounded), and the accessor that picks the table of a complex type's scalar
__device__ constexpr double kC32d[16] = { 1.0,
                                          0.98078528040323044913,
                                          0.92387953251128675613,
                                          0.83146961230254523708,
                                          0.70710678118654752440,
                                          0.55557023301960222474,
                                          0.38268343236508977173,
                                          0.19509032201612826785,
                                          0.0,
                                          -0.19509032201612826785,
                                          -0.38268343236508977173,
                                          -0.55557023301960222474,
                                          -0.70710678118654752440,
                                          -0.83146961230254523708,
                                          -0.92387953251128675613,
                                          -0.98078528040323044913 };
__device__ constexpr double kS32d[16] = { 0.0,
                                          0.19509032201612826785,
                                          0.38268343236508977173,
                                          0.55557023301960222474,
                                          0.70710678118654752440,
                                          0.83146961230254523708,
                                          0.92387953251128675613,
                                          0.98078528040323044913,
                                          1.0,
                                          0.98078528040323044913,
                                          0.92387953251128675613,
                                          0.83146961230254523708,
                                          0.70710678118654752440,
                                          0.55557023301960222474,
                                          0.38268343236508977173,
                                          0.19509032201612826785 };
template <typename C> struct w32;
template <> struct w32<float2> {
    using real = float;
    static __device__ constexpr float c(int e) { return kC32[e]; }
    static __device__ constexpr float s(int e) { return kS32[e]; }
};
template <> struct w32<double2> {
    using real = double;
    static __device__ constexpr double c(int e) { return kC32d[e]; }
    static __device__ constexpr double s(int e) { return kS32d[e]; }
};

// Five radix-2 DIF stages on 32 registers: x[k] is the element at base + k*stride.  Stage s pairs
// (k, k + h), h = 16 >> s.  The lower output owes the twiddle W_{2H}^(pos mod H) (fft.h:286 applies
// the same factor on the DIT side); it factors into the thread's w[s] (absent when TW is false) and
// the compile-time constant W_32^((k mod h) << s).  After full unrolling e is a literal.
// `wsrc` is the table W_1024^j (LDS or global) and `u` the thread's index: stage s fetches its
// thread twiddle W_1024^(u << s) when the stage starts, so it does not occupy registers earlier.
// S0 > 0 skips the first S0 stages: the 32 registers then hold 2^S0 independent groups of 32 >> S0 points.
// TABLE: wsrc points at the thread's column of a [stage][thread] thread-twiddle table with row pitch u, so
// stage s reads wsrc[s * u] (coalesced across lanes) instead of gathering wsrc[u << s] from the row W^j.
// CONJ: the table holds the other direction's values (the fused convolution runs its reverse transform on the forward
// plan's table): conjugate what is fetched.
template <bool REV, bool TW, int S0 = 0, bool TABLE = false, bool CONJ = false, typename C = float2>
__device__ __forceinline__ void fft32_dif(C (&x)[32], const C *wsrc, uint32_t u)
{
    using R = typename w32<C>::real;
#pragma unroll
    for (int s = S0; s < 5; s++) {
        const int h = 16 >> s;
        C ws = C{ R(1), R(0) };
        if constexpr (TW) {
            ws = TABLE ? wsrc[s * u] : wsrc[u << s];
            if constexpr (CONJ)
                ws.y = -ws.y;
        }
#pragma unroll
        for (int k = 0; k < 32; k++) {
            if ((k & h) != 0)
                continue;
            const C a = x[k], b = x[k + h];
            x[k] = a + b;
            C d = a - b;
            const int e = (k & (h - 1)) << s; // W_32 exponent, 0..15
            if (e == 8) {
                d = REV ? C{ -d.y, d.x } : C{ d.y, -d.x }; // -i / +i by swap and negate
            } else if (e != 0) {
                const R cr = w32<C>::c(e), ci = REV ? w32<C>::s(e) : -w32<C>::s(e);
                d = C{ d.x * cr - d.y * ci, d.x * ci + d.y * cr };
            }
            if constexpr (TW)
                d = cmul(d, ws);
            x[k + h] = d;
        }
    }
}

// the same five stages with the thread twiddles already in registers: w[s] = the stage's W^(u << s).  The caller fetches them
// ahead of the pass (under the data loads, under an LDS exchange), so the first stage does not wait for an L2 round trip
template <bool REV> __device__ __forceinline__ void fft32_dif_w(float2 (&x)[32], const float2 (&w)[5])
{
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const int h = 16 >> s;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            if ((k & h) != 0)
                continue;
            const float2 a = x[k], b = x[k + h];
            x[k] = a + b;
            float2 d = a - b;
            const int e = (k & (h - 1)) << s; // W_32 exponent, 0..15
            if (e == 8) {
                d = REV ? float2{ -d.y, d.x } : float2{ d.y, -d.x };
            } else if (e != 0) {
                const float cr = kC32[e], ci = REV ? kS32[e] : -kS32[e];
                d = float2{ d.x * cr - d.y * ci, d.x * ci + d.y * cr };
            }
            x[k + h] = cmul(d, w[s]);
        }
    }
}

__device__ __forceinline__ uint32_t brev5(uint32_t v) { return __brev(v) >> 27; }

} // namespace fft32
} // namespace sdsp_hip

// fft_big64.hip -- the registers-resident single-pass FFT kernel of fft_big.hip in DOUBLE precision (round 3):
// batched N = 4096 / 8192 / 16384 complex128 transforms with radix-2 butterfly stages (sdsp::fft_radix2, fft.h:258-299; the
// reference computes in double, fft.h:51-52, and the drop-in sdsp::fft_radix2/4<T,N> route to the f64 kernels).
//
// Before this kernel N = 8192 in double ran with its whole 128 KiB tile in LDS (fft_reg64.hip: one workgroup per CU, 52 % of
// HBM peak) and N = 16384 as three streaming passes (fft_mid.hip: 24.5 %, traffic 3x the algorithmic bytes).  Here the
// transform lives in REGISTERS -- 32 complex doubles per thread = 128 VGPRs, N/32 threads per transform, one transform per
// workgroup -- and LDS is only the exchange medium between the three register passes, crossed one plane (real, then
// imaginary) at a time: 8 N bytes of LDS per transform, so four workgroups of N = 4096, two of N = 8192 or one of N = 16384
// share a CU (the shapes of f32 N = 8192 / 16384 / 32768 in fft_big.hip).  ONE pass over HBM at every size.
//
//   load     x[k] = data[t + T*k]            T = N/32; a wave reads 1 KiB contiguous per instruction (16 bytes per lane)
//   pass A   five DIF stages, strides N/2 .. N/32      thread twiddles W_N^(t << s)
//   exchange position k*M + t  ->  blk*M + v + (j << R)            M = N/32, R = log2(N) - 10
//   pass B   five DIF stages inside the 32 blocks of M points     thread twiddles W_N^(32 v << s)
//   exchange position blk*M + v + (j << R)  ->  32*w + i,  w = bit_reverse(t)
//   pass C   the last R stages (2, 3 or 4) on 32 contiguous positions, constants only
//   store    X[t + T*bit_reverse5(i)] = x[i]           coalesced: the bit reversal (fft.h:269-273) is folded into w
//
// The position -> LDS slot map is fft_big.hip's XOR swizzle (sw<L>): a ds_read_b64 is serviced in two groups of 32 lanes with
// bank pair = slot mod 32, the same condition the b32 accesses of the f32 kernel meet, so all three read patterns are
// conflict-free (tools/model_fft_big_lds.py checks the forms below against sw<L> for every thread and register).
// Thread twiddles: the plan's [pass][stage][thread] table in double (capi.hip: upload_thread_twiddles_big, precision f64),
// combined with compile-time W_32 constants.  Parity: the reference's own bound 4 N eps against the oracle.
#include <hip/hip_runtime.h>

#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));

// rows of one transform through a buffer resource (fft32.h: make_rows): the row's byte offset in an SGPR next to ONE VGPR
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rows(const double2 *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(base), 0, (int)bytes, 0x00020000);
}
template <bool NT> __device__ __forceinline__ double2 row_load(__amdgpu_buffer_rsrc_t rows, uint32_t thread_off, uint32_t row_off)
{
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rows, thread_off, row_off, NT ? 2 : 0);
    const unsigned int a = v.x, b = v.y, c = v.z, d = v.w;
    return double2{ __hiloint2double((int)b, (int)a), __hiloint2double((int)d, (int)c) };
}
template <bool NT> __device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t rows, uint32_t thread_off, uint32_t row_off, double2 a)
{
    const v4u_t v = { (unsigned)__double2loint(a.x), (unsigned)__double2hiint(a.x), (unsigned)__double2loint(a.y),
                      (unsigned)__double2hiint(a.y) };
    __builtin_amdgcn_raw_buffer_store_b128(v, rows, thread_off, row_off, NT ? 2 : 0);
}
__device__ __forceinline__ double2 add2(double2 a, double2 b) { return double2{ a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ double2 sub2(double2 a, double2 b) { return double2{ a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return double2{ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x }; }

// cos / sin of 2*pi*j/32, j < 16 (correctly rounded doubles)
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

// Five radix-2 DIF stages on 32 registers (fft32.h: fft32_dif, in double): stage s pairs (k, k + h), h = 16 >> s; the lower
// output owes W^(pos mod H) = the thread's value of the stage (TW: wsrc[s * pitch], coalesced) x the literal W_32^((k mod h) << s).
// S0 > 0 skips the first S0 stages (pass C: the last R stages on 2^S0 groups of 32 >> S0 points).
template <bool REV, bool TW, int S0 = 0> __device__ __forceinline__ void fft32_dif_d(double2 (&x)[32], const double2 *wsrc, uint32_t pitch)
{
#pragma unroll
    for (int s = S0; s < 5; s++) {
        const int h = 16 >> s;
        double2 ws = double2{ 1.0, 0.0 };
        if constexpr (TW)
            ws = wsrc[s * pitch];
#pragma unroll
        for (int k = 0; k < 32; k++) {
            if ((k & h) != 0)
                continue;
            const double2 a = x[k], b = x[k + h];
            x[k] = add2(a, b);
            double2 d = sub2(a, b);
            const int e = (k & (h - 1)) << s; // W_32 exponent, 0..15
            if (e == 8) {
                d = REV ? double2{ -d.y, d.x } : double2{ d.y, -d.x }; // -i / +i by swap and negate
            } else if (e != 0) {
                const double cr = kC32d[e], ci = REV ? kS32d[e] : -kS32d[e];
                d = double2{ d.x * cr - d.y * ci, d.x * ci + d.y * cr };
            }
            if constexpr (TW)
                d = cmul(d, ws);
            x[k + h] = d;
        }
    }
}

// fft_big.hip's swizzle term for a position whose 32-block index (position >> (5 + R)) is k: k's five bits rotated by R
template <int R> __device__ __forceinline__ constexpr uint32_t rot5(uint32_t k)
{
    return ((k & ((1u << (5 - R)) - 1)) << R) | ((k >> (5 - R)) & ((1u << R) - 1));
}

template <int L, bool REV, bool NT>
__global__ __launch_bounds__((1 << L) / 32, 2) void sdsp_fft_big_f64_kernel(double2 *__restrict__ data, const double2 *__restrict__ tw,
                                                                           double scale, uint64_t batch)
{
    constexpr int R = L - 10;
    static_assert(R >= 2 && R <= 4, "N = 4096 / 8192 / 16384");
    constexpr uint32_t N = 1u << L, T = N / 32, M = N / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft_big64_smem[]; // N doubles: one plane of the transform
    auto lds_f64 = [&](uint32_t byte) -> double & { return *reinterpret_cast<double *>(sdsp_fft_big64_smem + byte); };

    const uint32_t t = threadIdx.x;
    const uint32_t toff = t * 16u;
    const uint64_t xform = blockIdx.x;
    if (xform >= batch)
        return;
    const __amdgpu_buffer_rsrc_t rows = make_rows(data + xform * N, N * sizeof(double2));

    double2 x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = row_load<NT>(rows, toff, T * k * sizeof(double2));

    // LDS byte addresses of the three access patterns: fft_big.hip's, with 8-byte slots
    //   pattern A  position k*M + t          ->  8*k*M + (8t ^ 8*rot(k))                  rot(k) is a literal
    //   pattern B  position pb + (j << R)    ->  baseB[j mod 2^(5-R)] + 256*(j >> (5-R))   no arithmetic per access
    //   pattern C  position 32*w + i         ->  (256*w + 8*xc) ^ 8*i                      xc per thread
    const uint32_t blk = t >> R, v = t & ((1u << R) - 1);
    constexpr int JL = 1 << (5 - R);
    const uint32_t xb = rot5<R>(blk);
    uint32_t base_b[JL];
#pragma unroll
    for (int jl = 0; jl < JL; jl++)
        base_b[jl] = 8u * (blk * M + (v ^ (xb & ((1u << R) - 1))) + (((uint32_t)jl ^ (xb >> R)) << R));
    const uint32_t w = __brev(t) >> (32 - (L - 5));
    const uint32_t base_c = (256u * w) | (8u * ((((w >> R) & ((1u << (5 - R)) - 1)) << R) | ((w >> 5) & ((1u << R) - 1))));
    const uint32_t base_a = 8u * t;

    fft32_dif_d<REV, true>(x, tw + t, T); // pass A: tw = [pass][stage][thread]

    // ---- exchange A -> B, one plane at a time
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t ta = base_a;
        asm volatile("" : "+v"(ta)); // keep the 32 addresses of a plane out of long-lived registers
#pragma unroll
        for (int k = 0; k < 32; k++)
            lds_f64(8u * k * M + (ta ^ (8u * rot5<R>(k)))) = half ? x[k].y : x[k].x;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const double f = lds_f64(base_b[j % JL] + 256u * (j / JL));
            if (half)
                x[j].y = f;
            else
                x[j].x = f;
        }
        __syncthreads();
    }

    fft32_dif_d<REV, true>(x, tw + 5 * T + t, T); // pass B

    // ---- exchange B -> C
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t tc = base_c;
        asm volatile("" : "+v"(tc));
#pragma unroll
        for (int j = 0; j < 32; j++)
            lds_f64(base_b[j % JL] + 256u * (j / JL)) = half ? x[j].y : x[j].x;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const double f = lds_f64(tc ^ (8u * i));
            if (half)
                x[i].y = f;
            else
                x[i].x = f;
        }
        if (half == 0)
            __syncthreads();
    }

    fft32_dif_d<REV, false, 5 - R>(x, tw, 0); // pass C: constants only

    // ---- store: position 32w + i holds X[bit_reverse_L(32w + i)] = X[t + T * bit_reverse5(i)]
#pragma unroll
    for (int i = 0; i < 32; i++) {
        double2 o = x[i];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            o.x *= scale;
            o.y *= scale;
        }
        row_store<NT>(rows, toff, T * (__brev((uint32_t)i) >> 27) * sizeof(double2), o);
    }
}

template <int L, bool REV> int launch_l(const fft_reg_args &a, hipStream_t s)
{
    constexpr size_t lds = sizeof(double) << L;
    auto kern = sdsp_fft_big_f64_kernel<L, REV, true>;
    if constexpr (lds > 64 * 1024) {
        static std::atomic<uint64_t> attr_done{ 0 };
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done))
            return rc;
    }
    if (a.batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((uint32_t)a.batch), dim3((1u << L) / 32), lds, s, reinterpret_cast<double2 *>(a.data),
                       reinterpret_cast<const double2 *>(a.tw), a.scale_d, a.batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_big64 launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace

bool fft_big64_supports(uint32_t n, int radix) { return radix == 2 && (n == 4096 || n == 8192 || n == 16384); }

int launch_fft_big_f64(const fft_reg_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (a.n) {
    case 4096: return a.reverse ? launch_l<12, true>(a, s) : launch_l<12, false>(a, s);
    case 8192: return a.reverse ? launch_l<13, true>(a, s) : launch_l<13, false>(a, s);
    case 16384: return a.reverse ? launch_l<14, true>(a, s) : launch_l<14, false>(a, s);
    default: break;
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the double-precision registers-resident kernel");
}
} // namespace sdsp_hip

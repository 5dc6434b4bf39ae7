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
// R4: radix-4 plans (sdsp::fft_radix4, fft.h:301-360 -- the reference's own precision and stage type) of N = 4096 = 4^6 and
// N = 16384 = 4^7 run genuine radix-4 DIF stages in the same kernel: fft32_r4.h's layers on double2, arranged exactly as in
// fft_big.hip's R4 form (two stages and half of the third in pass A, ...), on the plan's radix-4 thread-twiddle table in double.
#include <hip/hip_runtime.h>

#include "fft32.h"
#include "fft32_r4.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
using namespace fft32; // operator+/-, cmul, fft32_dif and the radix-4 layers, generic over float2 / double2

// rows of one transform through a buffer resource (fft32.h: make_rows): the row's byte offset in an SGPR next to ONE VGPR
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rows64(const double2 *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(base), 0, (int)bytes, 0x00020000);
}
template <bool NT> __device__ __forceinline__ double2 row_load64(__amdgpu_buffer_rsrc_t rows, uint32_t thread_off, uint32_t row_off)
{
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rows, thread_off, row_off, NT ? 2 : 0);
    const unsigned int a = v.x, b = v.y, c = v.z, d = v.w;
    return double2{ __hiloint2double((int)b, (int)a), __hiloint2double((int)d, (int)c) };
}
// The row offset travels in the VECTOR offset here, not in soffset.  A buffer store of more than 64 bits whose data registers
// are overwritten by the next VALU instruction needs a wait state in between; hipcc (ROCm 7.2) inserts it only when soffset is
// NOT a register (the rule of older parts) -- with the row in an SGPR it emitted `buffer_store_dwordx4 v[32:35], v116, s[0:3], s54
// offen nt` directly followed by `v_xor_b32 v32, 0xa0, v117`, and on some boxes, in some launches, the first data dword of a few
// lanes left as that address temp: doubles with a correct high word and an LDS address as the low word, results off by ~2e-7
// (profiles/r03_store_hazard.md).  With soffset = 0 the compiler's hazard recogniser covers the store.
template <bool NT> __device__ __forceinline__ void row_store64(__amdgpu_buffer_rsrc_t rows, uint32_t thread_off, uint32_t row_off, double2 a)
{
    const v4u_t v = { (unsigned)__double2loint(a.x), (unsigned)__double2hiint(a.x), (unsigned)__double2loint(a.y),
                      (unsigned)__double2hiint(a.y) };
    __builtin_amdgcn_raw_buffer_store_b128(v, rows, thread_off + row_off, 0, NT ? 2 : 0);
}

// fft_big.hip's swizzle term for a position whose 32-block index (position >> (5 + R)) is k: k's five bits rotated by R
template <int R> __device__ __forceinline__ constexpr uint32_t rot5(uint32_t k)
{
    return ((k & ((1u << (5 - R)) - 1)) << R) | ((k >> (5 - R)) & ((1u << R) - 1));
}

// the transform of one workgroup on registers: in x[k] = element t + T k, out x[i] = X[t + T bit_reverse5(i)] (position 32 w + i of the
// bit-reversed order, w = bit_reverse(t)).  REV: the direction of the constants; CONJ: the table holds the other direction's thread
// twiddles (the fused convolution runs its reverse transform on the forward plan's table).  The caller owns the barrier in front of a
// SECOND transform of the same workgroup (the first one's last plane is still being read)
template <int L, bool REV, bool R4, bool CONJ>
__device__ __forceinline__ void big64_transform(double2 (&x)[32], const double2 *__restrict__ tw, uint32_t t)
{
    constexpr int R = L - 10;
    constexpr uint32_t N = 1u << L, T = N / 32, M = N / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft_big64_smem[]; // N doubles: one plane of the transform
    auto lds_f64 = [&](uint32_t byte) -> double & { return *reinterpret_cast<double *>(sdsp_fft_big64_smem + byte); };
    static_assert(!(R4 && CONJ), "the convolution form runs radix-2 stages");
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

    [[maybe_unused]] auto tab = [&](int slot) { return tw[slot * T + t]; }; // radix-4 form: the thread's value of a table slot
    if constexpr (R4) { // fft_big.hip's arrangement of the seven (six) stages: see the comments there
        double2 thr[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
            thr[q] = tab(q); // W_N^((q + 1) t)
        r4_stage<REV, 4, 7, 2, true>(x, thr); // stage 0: quarter = register bits 4, 3; constant W_32^(q (k & 7))
#pragma unroll
        for (int q = 0; q < 3; q++)
            thr[q] = tab(3 + q); // W_(N/4)^((q + 1) t)
        r4_stage<REV, 2, 1, 8, true>(x, thr); // stage 1: register bits 2, 1; constant W_8^(q (k & 1))
        layer<1>(x); // stage 2, first layer: the quarter (1, 1) = odd registers of the upper half of the threads
        const bool upper = t >= T / 2;
#pragma unroll
        for (int k = 1; k < 32; k += 2) {
            const double2 r = rot_i<REV>(x[k]);
            x[k] = double2{ upper ? r.x : x[k].x, upper ? r.y : x[k].y };
        }
    } else {
        fft32_dif<REV, true, 0, true, CONJ>(x, tw + t, T); // pass A: tw = [pass][stage][thread]
    }

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

    if constexpr (R4) {
        layer<16>(x); // stage 2, second layer: register bit 4
        r4_split_twiddles<REV>(x, (blk & 1u) != 0, tab(6), tab(7), std::make_integer_sequence<int, 32>{});
        double2 thr[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
            thr[q] = tab(8 + q);
        r4_stage<REV, 3, 3, 4, true>(x, thr); // stage 3: register bits 3, 2; constant W_16^(q (j & 3))
#pragma unroll
        for (int q = 0; q < 3; q++)
            thr[q] = tab(11 + q);
        r4_stage<REV, 1, 0, 0, true>(x, thr); // stage 4: register bits 1, 0; thread twiddles only
    } else {
        fft32_dif<REV, true, 0, true, CONJ>(x, tw + 5 * T + t, T); // pass B
    }

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

    if constexpr (R4) {
        if constexpr (R == 4) {
            const double2 none[3] = {};
            r4_stage<REV, 3, 3, 4, false>(x, none); // stage 5 (N = 16384): register bits 3, 2; constants W_16^(q (i & 3)) only
        }
        r4_layers<REV, 1>(x); // the last stage (register bits 1, 0): no twiddles
    } else {
        fft32_dif<REV, false, 5 - R>(x, tw, 0); // pass C: constants only
    }

}

template <int L, bool REV, bool NT, bool R4 = false>
__global__ __launch_bounds__((1 << L) / 32, 2) void sdsp_fft_big_f64_kernel(double2 *__restrict__ data, const double2 *__restrict__ tw,
                                                                           double scale, uint64_t batch)
{
    constexpr int R = L - 10;
    static_assert(R >= 2 && R <= 4, "N = 4096 / 8192 / 16384");
    constexpr uint32_t N = 1u << L, T = N / 32;

    const uint32_t t = threadIdx.x;
    const uint32_t toff = t * 16u;
    const uint64_t xform = blockIdx.x;
    if (xform >= batch)
        return;
    static_assert(!R4 || L == 14 || L == 12, "radix-4 stages: N = 16384 = 4^7 and N = 4096 = 4^6");
    const __amdgpu_buffer_rsrc_t rows = make_rows64(data + xform * N, N * sizeof(double2));

    double2 x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = row_load64<NT>(rows, toff, T * k * sizeof(double2));

    big64_transform<L, REV, R4, false>(x, tw, t);

    // ---- store: position 32w + i holds X[bit_reverse_L(32w + i)] = X[t + T * bit_reverse5(i)]
#pragma unroll
    for (int i = 0; i < 32; i++) {
        double2 o = x[i];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            o.x *= scale;
            o.y *= scale;
        }
        row_store64<NT>(rows, toff, T * (__brev((uint32_t)i) >> 27) * sizeof(double2), o);
    }
}


// Fused fast convolution data <- IFFT(FFT(data) .* h) in double, N = 4096 / 8192 / 16384, radix-2 stages: the forward
// transform leaves register i holding X[t + T bit_reverse5(i)]; multiplied by H there (H rows through a buffer resource, default
// cache policy: every workgroup reads the same N values) and RENAMED z[bit_reverse5(i)] = x[i] -- no data moves -- that is the
// reverse transform's input layout, which runs on the same registers and LDS plane with the forward plan's table conjugated.
template <int L, bool NT>
__global__ __launch_bounds__((1 << L) / 32, 2) void sdsp_fft_big_f64_conv_kernel(double2 *__restrict__ data, const double2 *__restrict__ tw,
                                                                                const double2 *__restrict__ h, double scale, uint64_t batch)
{
    constexpr uint32_t N = 1u << L, T = N / 32;
    const uint32_t t = threadIdx.x;
    const uint32_t toff = t * 16u;
    const uint64_t xform = blockIdx.x;
    if (xform >= batch)
        return;
    const __amdgpu_buffer_rsrc_t rows = make_rows64(data + xform * N, N * sizeof(double2));
    const __amdgpu_buffer_rsrc_t rows_h = make_rows64(h, N * sizeof(double2));
    double2 x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = row_load64<NT>(rows, toff, T * k * sizeof(double2));
    big64_transform<L, false, false, false>(x, tw, t);
    double2 z[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        constexpr uint32_t es = sizeof(double2);
        const uint32_t r = __brev((uint32_t)i) >> 27;
        z[r] = cmul(x[i], row_load64<false>(rows_h, toff, T * r * es));
    }
    __syncthreads(); // every wave has read the forward transform's last plane
    big64_transform<L, true, false, true>(z, tw, t);
#pragma unroll
    for (int i = 0; i < 32; i++) {
        double2 o = z[i];
        o.x *= scale; // reverse_fft::ScaleValues, fft.h:128-132
        o.y *= scale;
        row_store64<NT>(rows, toff, T * (__brev((uint32_t)i) >> 27) * sizeof(double2), o);
    }
}

// Real-input packing in double around the same transform: n_real real samples = N = n_real / 2 complex ones.
// REAL = 1 (forward): transform, then split  X[a] = E + T, X[N-a] = conj(E - T),  E = (Z[a] + conj Z[N-a]) / 2, T = -i W_2N^a (Z[a] - conj Z[N-a]) / 2;
// REAL = 2 (inverse): merge (the inverse of that), then the reverse transform.  Register r holds element k = t + T b(r), b(r) = r before the
// transform and bit_reverse5(r) after it; the partner of (t, b) is (T - t, 31 - b).  The lower half of the elements (b < 16) is parked in LDS
// as complex values -- N/2 x 16 B: the plane's size, + one slot of padding for k = N/2 -- the upper half computes both results of its pair,
// keeps one and writes the other back (fft_big.hip's scheme).  w2n: W_2N^j, j < 2N, of the plan's direction.
template <int L, bool MERGE, bool AFTER>
__device__ __forceinline__ void big64_real_pairs(double2 (&y)[32], const double2 *__restrict__ w2n, uint32_t t)
{
    constexpr uint32_t N = 1u << L, T = N / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft_big64_smem[];
    auto lds_c = [&](uint32_t byte) -> double2 & { return *reinterpret_cast<double2 *>(sdsp_fft_big64_smem + byte); };
    const __amdgpu_buffer_rsrc_t wrows = make_rows64(w2n, 2 * N * sizeof(double2));
    if constexpr (AFTER)
        __syncthreads(); // every wave has read the transform's last plane
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const int b = AFTER ? (int)(__brev((uint32_t)r) >> 27) : r;
        if (b < 16)
            lds_c(16u * t + 16u * T * b) = y[r];
    }
    __syncthreads();
    const bool t0 = t == 0;
    const uint32_t pbase = 16u * (T - t); // partner of (t, b): slot (T - t) + T (31 - b) = N - k
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const int b = AFTER ? (int)(__brev((uint32_t)r) >> 27) : r;
        if (b < 16)
            continue;
        const double2 pa = lds_c(pbase + 16u * T * (31 - b));                  // element a = N - k
        const double2 w = row_load64<false>(wrows, pbase, 16u * T * (31 - b));  // W_2N^a
        const double2 own = y[r];
        const double2 e = double2{ 0.5 * (pa.x + own.x), 0.5 * (pa.y - own.y) }; // (A + conj B) / 2
        const double2 d = double2{ 0.5 * (pa.x - own.x), 0.5 * (pa.y + own.y) }; // (A - conj B) / 2
        const double2 wd = cmul(d, w);
        double2 ra, rb;
        if constexpr (MERGE) { // Z[a] = E + i O, Z[N-a] = conj(E - i O)
            ra = double2{ e.x - wd.y, e.y + wd.x };
            rb = double2{ e.x + wd.y, wd.x - e.y };
        } else { // X[a] = E + T, X[N-a] = conj(E - T), T = -i W D
            ra = double2{ e.x + wd.y, e.y - wd.x };
            rb = double2{ e.x - wd.y, -wd.x - e.y };
        }
        if (b == 16) { // k = N/2 in thread 0: conj, no partner (the slot it touched is the padding behind the plane)
            rb.x = t0 ? own.x : rb.x;
            rb.y = t0 ? -own.y : rb.y;
        }
        y[r] = rb;
        lds_c(pbase + 16u * T * (31 - b)) = ra;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const int b = AFTER ? (int)(__brev((uint32_t)r) >> 27) : r;
        if (b >= 16)
            continue;
        double2 v = lds_c(16u * t + 16u * T * b);
        if (b == 0) { // k = 0 in thread 0: (X[0], X[N]) packed, both real
            const double2 z = v;
            const double2 p0 = MERGE ? double2{ 0.5 * (z.x + z.y), 0.5 * (z.x - z.y) } : double2{ z.x + z.y, z.x - z.y };
            v.x = t0 ? p0.x : v.x;
            v.y = t0 ? p0.y : v.y;
        }
        y[r] = v;
    }
}

template <int L, int REAL, bool NT>
__global__ __launch_bounds__((1 << L) / 32, 2) void sdsp_fft_big_f64_real_kernel(double2 *__restrict__ data, const double2 *__restrict__ tw,
                                                                                const double2 *__restrict__ w2n, double scale, uint64_t batch)
{
    static_assert(REAL == 1 || REAL == 2, "1 = forward (split after the transform), 2 = inverse (merge before it)");
    constexpr uint32_t N = 1u << L, T = N / 32;
    const uint32_t t = threadIdx.x;
    const uint32_t toff = t * 16u;
    const uint64_t xform = blockIdx.x;
    if (xform >= batch)
        return;
    const __amdgpu_buffer_rsrc_t rows = make_rows64(data + xform * N, N * sizeof(double2));
    double2 x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = row_load64<NT>(rows, toff, T * k * sizeof(double2));
    if constexpr (REAL == 2) {
        big64_real_pairs<L, true, false>(x, w2n, t);
        __syncthreads(); // every wave has read its parked values back: the plane is free for the transform
    }
    big64_transform<L, REAL == 2, false, false>(x, tw, t);
    if constexpr (REAL == 1)
        big64_real_pairs<L, false, true>(x, w2n, t);
#pragma unroll
    for (int i = 0; i < 32; i++) {
        double2 o = x[i];
        if constexpr (REAL == 2) { // reverse_fft::ScaleValues, fft.h:128-132
            o.x *= scale;
            o.y *= scale;
        }
        row_store64<NT>(rows, toff, T * (__brev((uint32_t)i) >> 27) * sizeof(double2), o);
    }
}

template <int L, bool REV, bool R4 = false> int launch_l(const fft_reg_args &a, hipStream_t s)
{
    constexpr size_t lds = sizeof(double) << L;
    auto kern = sdsp_fft_big_f64_kernel<L, REV, true, R4>;
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

template <int L> int launch_conv_l(const fft_reg_args &a, hipStream_t s)
{
    constexpr size_t lds = sizeof(double) << L;
    auto kern = sdsp_fft_big_f64_conv_kernel<L, true>;
    if constexpr (lds > 64 * 1024) {
        static std::atomic<uint64_t> attr_done{ 0 };
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done))
            return rc;
    }
    if (a.batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((uint32_t)a.batch), dim3((1u << L) / 32), lds, s, reinterpret_cast<double2 *>(a.data),
                       reinterpret_cast<const double2 *>(a.tw), reinterpret_cast<const double2 *>(a.tw2), a.scale_d, a.batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_big64 convolution launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

template <int L, int REAL> int launch_real_l(const fft_reg_args &a, hipStream_t s)
{
    constexpr size_t lds = (sizeof(double) << L) + 16; // one complex slot of padding behind the plane (k = N/2)
    auto kern = sdsp_fft_big_f64_real_kernel<L, REAL, true>;
    if constexpr (lds > 64 * 1024) {
        static std::atomic<uint64_t> attr_done{ 0 };
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done))
            return rc;
    }
    if (a.batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((uint32_t)a.batch), dim3((1u << L) / 32), lds, s, reinterpret_cast<double2 *>(a.data),
                       reinterpret_cast<const double2 *>(a.tw), reinterpret_cast<const double2 *>(a.tw2), a.scale_d, a.batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_big64 real-input launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace

// real-input plans in double, n = n_real / 2 = 4096 / 8192 / 16384, radix-2 stages
bool fft_big64_real_supports(uint32_t n, int radix) { return radix == 2 && (n == 4096 || n == 8192 || n == 16384); }
// the fused convolution in double runs radix-2 stages (a radix-2 plan's table): N = 4096 / 8192 / 16384
bool fft_big64_conv_supports(uint32_t n, int radix) { return radix == 2 && (n == 4096 || n == 8192 || n == 16384); }

bool fft_big64_supports(uint32_t n, int radix)
{
    if (radix == 4)
        return n == 4096 || n == 16384; // genuine radix-4 stages (the R4 form)
    return radix == 2 && (n == 4096 || n == 8192 || n == 16384);
}

int launch_fft_big_f64(const fft_reg_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a.real_mode == 1 || a.real_mode == 2) { // real-input plans: W_2N in tw2; the direction is the mode's
        const bool inv = a.real_mode == 2;
        switch (a.radix == 2 ? a.n : 0u) {
        case 4096: return inv ? launch_real_l<12, 2>(a, s) : launch_real_l<12, 1>(a, s);
        case 8192: return inv ? launch_real_l<13, 2>(a, s) : launch_real_l<13, 1>(a, s);
        case 16384: return inv ? launch_real_l<14, 2>(a, s) : launch_real_l<14, 1>(a, s);
        default: return fail(SDSP_HIP_ERR_UNSUPPORTED, "real-input plans in double on this kernel: radix 2, n_real = 8192 / 16384 / 32768");
        }
    }
    if (a.real_mode == 3) { // fused convolution: h travels in tw2
        switch (a.radix == 2 ? a.n : 0u) {
        case 4096: return launch_conv_l<12>(a, s);
        case 8192: return launch_conv_l<13>(a, s);
        case 16384: return launch_conv_l<14>(a, s);
        default: return fail(SDSP_HIP_ERR_UNSUPPORTED, "fused convolution in double: radix-2 plans of N = 4096 / 8192 / 16384");
        }
    }
    if (a.radix == 4) { // a.tw: the radix-4 thread-twiddle table in double
        if (a.n == 4096)
            return a.reverse ? launch_l<12, true, true>(a, s) : launch_l<12, false, true>(a, s);
        if (a.n == 16384)
            return a.reverse ? launch_l<14, true, true>(a, s) : launch_l<14, false, true>(a, s);
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "radix-4 stages in double: N = 4096 / 16384");
    }
    switch (a.n) {
    case 4096: return a.reverse ? launch_l<12, true>(a, s) : launch_l<12, false>(a, s);
    case 8192: return a.reverse ? launch_l<13, true>(a, s) : launch_l<13, false>(a, s);
    case 16384: return a.reverse ? launch_l<14, true>(a, s) : launch_l<14, false>(a, s);
    default: break;
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the double-precision registers-resident kernel");
}
} // namespace sdsp_hip

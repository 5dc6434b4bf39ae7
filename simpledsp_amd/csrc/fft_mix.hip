// fft_mix.hip -- batched N = 8192 and N = 16384 complex f32 FFT for gfx950: ONE leading radix-2 / radix-4 DIF stage in
// registers, then two / four N = 4096 transforms through the tuned radix-4 machinery of fft4096_kernels.h -- the
// "N = 2 * 4^k through the radix-4 kernel with one radix-2 stage" of SURVEY 8(f)-4, and genuine radix-4 butterflies
// (fft.h:311-349) for sdsp::fft_radix4<T,16384>.
//
//   y_q[n] = ( sum_r x[n + 4096 r] w_R^(r q) ) * W_N^(n q)        n < 4096, q < R        (the DIF stage, fft.h:322-345)
//   X[R k + q] = FFT_4096(y_q)[k]
//
// One 256-thread workgroup per transform.  Thread t loads x[t + 256 m], m < 16 R (512 contiguous bytes per wave
// instruction, ascending addresses): the stage's partners n + 4096 r are its own registers m = k + 16 r, and what it
// leaves, y_q[t + 256 k], k < 16, is exactly the register layout the N = 4096 kernel's first pass consumes.  The R
// sub-transforms then run one after the other on the same 32 KiB LDS tile (fft4096_in_regs); each leaves the thread
// holding FFT(y_q)[t + 256 j], so X[R (t + 256 j) + q], q < R, are R adjacent elements: one 16-byte (R = 2) or two
// 16-byte (R = 4) stores per lane, 1-2 KiB contiguous per wave instruction.  HBM sees every element once each way.
// The stage's twiddle W_N^(n q), n = t + 256 k, factors into a per-thread value W_N^(q t) (R - 1 coalesced loads from the
// plan's table) and the compile-time constant W_(16 R)^(q k).
#include <hip/hip_runtime.h>

#include "fft4096_kernels.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
using namespace fft4096;

// cos / sin of 2 pi e / 64
__device__ constexpr float kC64[64] = { 1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 6.123233996e-17f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f, -1.836970199e-16f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f };
__device__ constexpr float kS64[64] = { 0.000000000e+00f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f, 1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 1.224646799e-16f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f };

// a * W_64^E (forward) / its conjugate (reverse), E compile-time
template <bool REV, int E> __device__ __forceinline__ float2 mul_w64(float2 a)
{
    constexpr int e = E & 63;
    if constexpr (e == 0)
        return a;
    else if constexpr (e == 16)
        return rot90<REV>(a);
    else if constexpr (e == 32)
        return float2{ -a.x, -a.y };
    else if constexpr (e == 48)
        return rot90<!REV>(a);
    else
        return cmulk<REV>(a, kC64[e], -kS64[e]);
}

template <int R, bool REV, int K> __device__ __forceinline__ void lead_one(float2 (&x)[R][16], const float2 (&ws)[R - 1])
{
    if constexpr (R == 2) {
        const float2 a = x[0][K], b = x[1][K];
        x[0][K] = a + b;
        x[1][K] = cmul(mul_w64<REV, 2 * K>(a - b), ws[0]); // W_32^k = W_64^(2k)
    } else {
        bfly4<REV>(x[0][K], x[1][K], x[2][K], x[3][K]); // outputs q = 0 .. 3 in place (fft.h:342-345)
        x[1][K] = cmul(mul_w64<REV, K>(x[1][K]), ws[0]);
        x[2][K] = cmul(mul_w64<REV, 2 * K>(x[2][K]), ws[1]);
        x[3][K] = cmul(mul_w64<REV, 3 * K>(x[3][K]), ws[2]);
    }
}
template <int R, bool REV, int... Ks>
__device__ __forceinline__ void lead_stage(float2 (&x)[R][16], const float2 (&ws)[R - 1], std::integer_sequence<int, Ks...>)
{
    (lead_one<R, REV, Ks>(x, ws), ...);
}

// the leading radix-4 stage of the 512-thread form: y[r][j] = x[t + 512 (j + 8 r)] -> y_q[t + 512 j]; the twiddle W_N^(q n),
// n = t + 512 j, is ws[q - 1] = W_N^(q t) times the constant W_32^(q j) = W_64^(2 q j)
template <bool REV, int J> __device__ __forceinline__ void lead_one512(float2 (&y)[4][8], const float2 (&ws)[3])
{
    bfly4<REV>(y[0][J], y[1][J], y[2][J], y[3][J]);
    y[1][J] = cmul(mul_w64<REV, 2 * J>(y[1][J]), ws[0]);
    y[2][J] = cmul(mul_w64<REV, 4 * J>(y[2][J]), ws[1]);
    y[3][J] = cmul(mul_w64<REV, 6 * J>(y[3][J]), ws[2]);
}
template <bool REV, int... Js>
__device__ __forceinline__ void lead_stage512(float2 (&y)[4][8], const float2 (&ws)[3], std::integer_sequence<int, Js...>)
{
    (lead_one512<REV, Js>(y, ws), ...);
}

typedef float v4f_t __attribute__((ext_vector_type(4)));

// tw: the sub-transforms' thread-twiddle table (layout of upload_thread_twiddles_4096, radix 4, values W_4096^j = W_N^(R j));
// tws: [q - 1][t] = W_N^(q t).  Both direction-folded (conjugated for reverse plans).
template <int R, bool REV>
__global__ __launch_bounds__(256, R == 2 ? 4 : 2) void sdsp_fft_mix_f32(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                                        const float2 *__restrict__ tws, uint64_t batch, float scale)
{
    constexpr uint32_t N = R * 4096u;
    __shared__ __attribute__((aligned(16))) float2 lds[4096];
    const uint32_t t = threadIdx.x;
    float2 wA1[3], wA2[3], wB1[3], wB2[3], ws[R - 1];
    const uint32_t rr = t & 15;
#pragma unroll
    for (int r = 1; r < 4; r++) { // see sdsp_fft4096_r4_f32
        wA1[r - 1] = tw[(r - 1) * 256 + t];
        wA2[r - 1] = tw[(r + 2) * 256 + t];
        wB1[r - 1] = tw[1536 + (r - 1) * 16 + rr];
        wB2[r - 1] = tw[1536 + (r + 2) * 16 + rr];
    }
#pragma unroll
    for (int q = 1; q < R; q++)
        ws[q - 1] = tws[(q - 1) * 256 + t];
    const lds_map mp = make_lds_map<false>(lds, t);

    const uint64_t f = blockIdx.x;
    if (f >= batch)
        return;
    float2 x[R][16];
    const float2 *src = data + f * N + t;
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int k = 0; k < 16; k++)
            x[r][k] = gload(src + 256 * (k + 16 * r));

    lead_stage<R, REV>(x, ws, std::make_integer_sequence<int, 16>{});
#pragma unroll
    for (int q = 0; q < R; q++)
        fft4096_in_regs<REV, false>(x[q], lds, mp, wA1, wA2, wB1, wB2);

    // x[q][k] = FFT(y_q)[t + 256 j], j = 4 (k & 3) + (k >> 2)  ->  X[R (t + 256 j) + q]
    float2 *dst = data + f * N + (size_t)R * t;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int j = 4 * (k & 3) + (k >> 2);
#pragma unroll
        for (int q = 0; q < R; q += 2) {
            float2 a = x[q][k], b = x[q + 1][k];
            if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
                a.x *= scale, a.y *= scale, b.x *= scale, b.y *= scale;
            }
            const v4f_t v = { a.x, a.y, b.x, b.y };
            __builtin_nontemporal_store(v, reinterpret_cast<v4f_t *>(dst + R * 256 * j + q));
        }
    }
}

// ---- N = 16384 = 4 x 4096 on a 512-thread workgroup: two halves of 256 threads, two sub-transforms each ---------------
// The 256-thread form above needs 64 points per thread at R = 4 (192-198 VGPRs, two 4-wave workgroups per CU: 46 % of HBM
// peak).  Here a thread holds 32: thread t < 512 loads x[t + 512 m], m < 32 -- the leading radix-4 stage's partners
// n + 4096 r are still its own registers (m = j + 8 r) -- and the stage leaves y_q[t + 512 j], j < 8, for all four q.  Half
// H = t >> 8 of the workgroup then runs the sub-transforms q = 2H, 2H + 1 in the tuned layout (thread tt = t & 255 holds
// y_q[tt + 256 k], k < 16): k of one parity it already has, the other parity sits in thread t ^ 256, which wants exactly
// the sixteen values this thread has no use for -- ONE pairwise exchange of 128 bytes per thread through LDS (the 64 KiB of
// the two halves' FFT tiles, before they are needed).  Each half has its own 32 KiB tile; 64 data VGPRs; two 512-thread
// workgroups per CU -- the footprint of four N = 8192 workgroups.  A lane stores X[4 (tt + 256 j) + 2H], + 1: 16 bytes at a
// 32-byte stride; the other half of the workgroup writes the 16 bytes in between at the same time (the halves run in
// lockstep) -- which measured 28 % of HBM peak (partial sectors), so a second pairwise exchange regroups the results and
// every lane stores 32 contiguous bytes.
template <bool REV>
__global__ __launch_bounds__(512, 4) void sdsp_fft_mix4_f32(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                            const float2 *__restrict__ tws, uint64_t batch, float scale)
{
    constexpr uint32_t N = 16384;
    __shared__ __attribute__((aligned(16))) float2 lds[2 * 4096];
    const uint32_t t = threadIdx.x, H = t >> 8, tt = t & 255;
    float2 wA1[3], wA2[3], wB1[3], wB2[3], ws[3];
    const uint32_t rr = tt & 15;
#pragma unroll
    for (int r = 1; r < 4; r++) { // see sdsp_fft4096_r4_f32
        wA1[r - 1] = tw[(r - 1) * 256 + tt];
        wA2[r - 1] = tw[(r + 2) * 256 + tt];
        wB1[r - 1] = tw[1536 + (r - 1) * 16 + rr];
        wB2[r - 1] = tw[1536 + (r + 2) * 16 + rr];
        ws[r - 1] = tws[(r - 1) * 512 + t]; // W_N^(r t)
    }
    const lds_map mp = make_lds_map<false>(lds + 4096 * H, tt);

    const uint64_t f = blockIdx.x;
    if (f >= batch)
        return;
    float2 y[4][8]; // y[r][j] = x[t + 512 (j + 8 r)], then y_q[t + 512 j]
    const float2 *src = data + f * N + t;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < 8; j++)
            y[r][j] = gload(src + 512 * (j + 8 * r));
    lead_stage512<REV>(y, ws, std::make_integer_sequence<int, 8>{});

    // pairwise exchange with thread t ^ 256: it gets the two sub-transforms this half does not run
    float2 z[2][16];
    {
        float2 *const xbuf = lds; // [16 values][512 threads]
        if (H == 0) {
#pragma unroll
            for (int v = 0; v < 16; v++)
                xbuf[v * 512 + t] = y[2 + (v >> 3)][v & 7];
        } else {
#pragma unroll
            for (int v = 0; v < 16; v++)
                xbuf[v * 512 + t] = y[v >> 3][v & 7];
        }
        __syncthreads();
        // own values: k of parity H; received: the other parity (y_q[tt + 256 k] lives in thread tt + 256 (k & 1), j = k >> 1)
        if (H == 0) {
#pragma unroll
            for (int v = 0; v < 16; v++) {
                z[v >> 3][2 * (v & 7)] = y[v >> 3][v & 7];
                z[v >> 3][2 * (v & 7) + 1] = xbuf[v * 512 + (t ^ 256)];
            }
        } else {
#pragma unroll
            for (int v = 0; v < 16; v++) {
                z[v >> 3][2 * (v & 7) + 1] = y[2 + (v >> 3)][v & 7];
                z[v >> 3][2 * (v & 7)] = xbuf[v * 512 + (t ^ 256)];
            }
        }
        __syncthreads(); // the exchange buffer is the FFT tiles
    }
    fft4096_in_regs<REV, false>(z[0], lds + 4096 * H, mp, wA1, wA2, wB1, wB2);
    fft4096_in_regs<REV, false>(z[1], lds + 4096 * H, mp, wA1, wA2, wB1, wB2);

    // z[ql][k] = FFT(y_(2H+ql))[tt + 256 j], j = 4 (k & 3) + (k >> 2).  X[4 (tt + 256 j) + q], q < 4, are 32 contiguous bytes
    // of which this thread holds q = 2H, 2H + 1 and its partner t ^ 256 the other two: a second pairwise exchange gives
    // each thread all four for the j of one parity (H), so a lane stores 32 contiguous bytes (16-byte pieces of the two halves
    // interleaved at a 32-byte stride measured 28 % of HBM peak: partial sectors).  Registers with (k >> 2) & 1 == H stay.
    float2 other[2][8]; // the partner's sub-transforms, for this thread's eight j
    {
        float2 *const xbuf = lds; // [16 values][512 threads]; every LDS read of the sub-transforms is behind a barrier
        if (H == 0) {
#pragma unroll
            for (int v = 0; v < 16; v++) // send odd j: k = 4 + (v & 3) + 8 ((v >> 2) & 1)
                xbuf[v * 512 + t] = z[v >> 3][4 + (v & 3) + 8 * ((v >> 2) & 1)];
        } else {
#pragma unroll
            for (int v = 0; v < 16; v++) // send even j: k = (v & 3) + 8 ((v >> 2) & 1)
                xbuf[v * 512 + t] = z[v >> 3][(v & 3) + 8 * ((v >> 2) & 1)];
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 16; v++)
            other[v >> 3][v & 7] = xbuf[v * 512 + (t ^ 256)];
    }
    float2 *dst = data + f * N + 4 * tt;
#pragma unroll
    for (int i = 0; i < 8; i++) { // this thread's i-th j: k = (i & 3) + 8 (i >> 2) + 4 H
        const int kk = (i & 3) + 8 * (i >> 2);
        float2 mine0 = H ? z[0][kk + 4] : z[0][kk], mine1 = H ? z[1][kk + 4] : z[1][kk];
        float2 o0 = other[0][i], o1 = other[1][i];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            mine0.x *= scale, mine0.y *= scale, mine1.x *= scale, mine1.y *= scale;
            o0.x *= scale, o0.y *= scale, o1.x *= scale, o1.y *= scale;
        }
        // q = 0, 1 belong to half 0, q = 2, 3 to half 1
        const v4f_t lo = H ? v4f_t{ o0.x, o0.y, o1.x, o1.y } : v4f_t{ mine0.x, mine0.y, mine1.x, mine1.y };
        const v4f_t hi = H ? v4f_t{ mine0.x, mine0.y, mine1.x, mine1.y } : v4f_t{ o0.x, o0.y, o1.x, o1.y };
        const int k = kk; // j of register k (either half): 4 (k & 3) + (k >> 2), + 1 for the odd half
        float2 *q = dst + 4 * 256 * (4 * (k & 3) + (k >> 2)) + 4 * 256 * (int)H;
        __builtin_nontemporal_store(lo, reinterpret_cast<v4f_t *>(q));
        __builtin_nontemporal_store(hi, reinterpret_cast<v4f_t *>(q + 2));
    }
}

template <int R> int launch_r(const fft_mix_args &a, hipStream_t s)
{
    if (a.batch > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    float2 *d = reinterpret_cast<float2 *>(a.data);
    const float2 *tw = reinterpret_cast<const float2 *>(a.tw), *tws = reinterpret_cast<const float2 *>(a.tw_lead);
    const dim3 grid((uint32_t)a.batch);
    if (a.reverse)
        hipLaunchKernelGGL((sdsp_fft_mix_f32<R, true>), grid, dim3(256), 0, s, d, tw, tws, a.batch, a.scale);
    else
        hipLaunchKernelGGL((sdsp_fft_mix_f32<R, false>), grid, dim3(256), 0, s, d, tw, tws, a.batch, a.scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_mix launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace

// N = 8192 (either stage type asked for: radix 2 plans, or radix 0 = auto) and N = 16384 (radix 4 and radix 2 plans)
bool fft_mix_supports(uint32_t n) { return n == 8192 || n == 16384; }

int launch_fft_mix_f32(const fft_mix_args &a, void *stream)
{
    if (a.batch == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a.n == 8192)
        return launch_r<2>(a, s);
    if (a.n == 16384) {
        if (a.batch > 0x7fffffffull)
            return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
        float2 *d = reinterpret_cast<float2 *>(a.data);
        const float2 *tw = reinterpret_cast<const float2 *>(a.tw), *tws = reinterpret_cast<const float2 *>(a.tw_lead);
        if (a.reverse)
            hipLaunchKernelGGL(sdsp_fft_mix4_f32<true>, dim3((uint32_t)a.batch), dim3(512), 0, s, d, tw, tws, a.batch, a.scale);
        else
            hipLaunchKernelGGL(sdsp_fft_mix4_f32<false>, dim3((uint32_t)a.batch), dim3(512), 0, s, d, tw, tws, a.batch, a.scale);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return fail(SDSP_HIP_ERR_HIP, std::string("fft_mix launch: ") + hipGetErrorString(e));
        return SDSP_HIP_OK;
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the mixed-radix kernels");
}
} // namespace sdsp_hip

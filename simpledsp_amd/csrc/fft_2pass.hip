// fft_2pass.hip -- batched N = 2^16 .. 2^19 complex f32 FFT (and, round 3, N = 2^16 .. 2^20 in double: the same templates
// on double2) in TWO passes over HBM, for gfx950: the tile scheme of the
// N = 2^20 kernels (fft1m_kernels.h) generalised to N = N1 x N2 with N1, N2 in {256, 512, 1024}.
//
// Radix-2 butterfly stages of sdsp::fft_radix2 (fft.h:276-294) as a four-step decomposition, the transform viewed as a
// row-major [n1][n2] matrix with N2 columns:
//
//   pass 1 (sdsp_fft2p_cols)  for 16 adjacent columns n2: log2 N1 radix-2 DIF stages over n1 (stride N2), times the
//                             inter-pass twiddle W_N^(n2*k1), written to the intermediate [n2 / 16][k1][n2 % 16] (a tile's
//                             output is one contiguous block of 128 * N1 bytes)
//   pass 2 (sdsp_fft2p_rows)  for 16 adjacent rows k1: log2 N2 stages over n2, written transposed, X[k1 + N1*k2], back into
//                             the caller's buffer (128-byte segments)
//
// Both passes: a sequence of length Ns = 32 T is held by T threads with 32 points each; the first register pass runs five DIF
// stages across the thread's 32 points (row stride T), ONE exchange through LDS (plane by plane: real, then imaginary)
// regroups them so that a thread holds 32 consecutive points, and the second register pass runs the remaining log2 T stages
// on 32 / T independent groups of T points.  Stage twiddles W_Ns^(2^s u) come from an LDS copy of W_1024 (pass 1) or a
// [stage][lane] table built from it (pass 2) times compile-time W_32 constants (fft32.h); the inter-pass twiddle is
// W_1024^(m >> (L - 10)) from that table times a two-term series for the remaining angle (< 2 pi / 1024), as in fft1m.
// LDS planes are XOR-swizzled: all reads conflict-free; pass 2's writes 2-way (N2 = 512) / 4-way (N2 = 256) on ds_write_b32.
//
// Replaces, for these sizes, the three streaming passes of fft_mid.hip (16-point column step, 16 x batch row transforms,
// untwist): two passes over HBM instead of three.
#include <hip/hip_runtime.h>

#include "fft32.h"
#include "handoff.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
using namespace fft32;

constexpr int kTile = 16; // sequences per tile

template <int BITS> __device__ __forceinline__ uint32_t brev_bits(uint32_t v) { return BITS == 0 ? 0u : (__brev(v) >> (32 - BITS)); }

// W_N^m, N = 2^L, m < N: coarse factor from the LDS copy of W_1024, fine factor (angle < 2 pi / 1024) from a short series:
// two terms are exact to fp32; double takes the terms up to th^6 / 720 and th^7 / 5040 (< 1e-19 relative)
template <int L, bool REV> __device__ __forceinline__ float2 twiddle_n(const float2 *w1k, uint32_t m)
{
    constexpr int FB = L - 10; // fine bits
    const float th = (float)(m & ((1u << FB) - 1u)) * (6.283185307179586476925f / (float)(1u << L));
    const float th2 = th * th;
    const float sn = th - th * th2 * 0.16666667f;
    const float2 fine = float2{ 1.0f - 0.5f * th2, REV ? sn : -sn };
    return cmul(w1k[m >> FB], fine);
}
template <int L, bool REV> __device__ __forceinline__ double2 twiddle_n(const double2 *w1k, uint32_t m)
{
    constexpr int FB = L - 10;
    const double th = (double)(m & ((1u << FB) - 1u)) * (6.283185307179586476925 / (double)(1u << L));
    const double t2 = th * th;
    const double cs = 1.0 - t2 * (0.5 - t2 * (1.0 / 24.0 - t2 * (1.0 / 720.0)));
    const double sn = th * (1.0 - t2 * (1.0 / 6.0 - t2 * (1.0 / 120.0 - t2 * (1.0 / 5040.0))));
    const double2 fine = double2{ cs, REV ? sn : -sn };
    return cmul(w1k[m >> FB], fine);
}

// ---- pass 1: 16 columns of one transform; 16 * T1 threads ----------------------------------------------------
// in_x / ws_x: the transform's input matrix / its intermediate.  LDS: plane N1 x 16 floats, w1k = W_1024 (8 KiB, staged by
// the caller), qtab 32 x 16 float2 (4 KiB)
// t: the thread's index within the tile's 16 * T1 threads (the persistent kernel runs tiles side by side in one workgroup);
// WT: write-through stores of the intermediate (the persistent kernel's hand-off, handoff.h)
template <int L, int L1, bool REV, typename C, bool WT = false>
__device__ __forceinline__ void cols_tile2p(const C *in_x, C *ws_x, uint32_t tile, typename w32<C>::real *plane, const C *w1k, C *qtab,
                                            uint32_t t)
{
    using Real = typename w32<C>::real;
    constexpr int L2 = L - L1, N1 = 1 << L1, N2 = 1 << L2, T1 = N1 / 32;
    const uint32_t c = t & 15, u = t >> 4; // u < T1
    const uint32_t n2 = tile * kTile + c;

    // rows u + T1 k of column n2
    const C *src_tile = in_x + tile * kTile;
    const uint32_t toff = (u * N2 + c) * (uint32_t)sizeof(C);
    C x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = nt_load(at(src_tile + (size_t)T1 * N2 * k, toff));
    __syncthreads(); // w1k staged (the two-launch kernels stage it while these loads are in flight)

    // the column part of the inter-pass twiddle: W_N^(n2 * j * 2^(L1-5)), j < 32; 32 / T1 entries per thread
#pragma unroll
    for (int i = 0; i < 32 / T1; i++) {
        const uint32_t j = u + T1 * i;
        qtab[j * 16 + c] = twiddle_n<L, REV>(w1k, (n2 * j) << (L1 - 5));
    }
    fft32_dif<REV, true>(x, w1k, u * (1024 / N1)); // W_N1^(u 2^s) = W_1024^((u 1024/N1) << s)

    // exchange rows {u + T1 k} -> {32 u + k}; slot(row, col) = (row * 16 + col) ^ (((row >> 5) & 1) << 4)
    {
        Real *const w0 = plane + (u * 16 + c);
        Real *const w1 = plane + ((u * 16 + c) ^ 16);
        const int flip = (int)(u & 1) * 16;
        const Real *const r_even = plane + (512 * u + c) + flip;
        const Real *const r_odd = plane + (512 * u + c) - flip;
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int k = 0; k < 32; k++) // row u + T1 k: bit 5 of the row = bit of k (no carry from u < T1)
                ((((k * T1) >> 5) & 1) ? w1 : w0)[16 * T1 * k] = half ? x[k].y : x[k].x;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const Real f = ((k & 1) ? r_odd : r_even)[16 * k];
                if (half)
                    x[k].y = f;
                else
                    x[k].x = f;
            }
            if (half == 0)
                __syncthreads();
        }
    }
    fft32_dif<REV, false, 10 - L1>(x, w1k, 0); // the last log2 T1 stages on 32 / T1 groups of T1 consecutive rows

    // row 32 u + k holds Y[k1], k1 = bit_reverse_L1(32 u + k) = (bit_reverse5(k) << (L1 - 5)) | bit_reverse(u)
    const uint32_t bu = brev_bits<L1 - 5>(u);
    const C pw = twiddle_n<L, REV>(w1k, n2 * bu);
    const C *const qcol = qtab + c;
    C *dst_tile = ws_x + (size_t)tile * (N1 * kTile); // [tile][k1][c]
    const uint32_t soff = (bu * 16 + c) * (uint32_t)sizeof(C);
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if ((k & 7) == 0)
            __builtin_amdgcn_sched_barrier(0);
        const int j = (int)(__brev((uint32_t)k) >> 27);
        const C v = cmul(x[k], cmul(pw, qcol[16 * j]));
        if constexpr (WT)
            handoff::wt_store(at(dst_tile + (size_t)j * (16 << (L1 - 5)), soff), v);
        else
            *at(dst_tile + (size_t)j * (16 << (L1 - 5)), soff) = v;
    }
}
template <typename C> __device__ __forceinline__ void stage_w1k2p(C *w1k, const C *tw_1024, uint32_t threads)
{
    // 16-byte copies: two f32 twiddles / one f64 twiddle each
    constexpr uint32_t n16 = 1024 * sizeof(C) / 16;
    for (uint32_t i = threadIdx.x; i < n16; i += threads)
        reinterpret_cast<float4 *>(w1k)[i] = reinterpret_cast<const float4 *>(tw_1024)[i];
}
template <int L, int L1, bool REV, typename C>
__global__ __launch_bounds__(kTile * (1 << (L1 - 5))) void sdsp_fft2p_cols(const C *__restrict__ in, C *__restrict__ ws,
                                                                          const C *__restrict__ tw_1024)
{
    using Real = typename w32<C>::real;
    constexpr int L2 = L - L1, N1 = 1 << L1, N2 = 1 << L2, T1 = N1 / 32, THREADS = kTile * T1, TILES = N2 / kTile;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft2p_smem[];
    Real *plane = reinterpret_cast<Real *>(sdsp_fft2p_smem);
    C *w1k = reinterpret_cast<C *>(sdsp_fft2p_smem + (size_t)N1 * kTile * sizeof(Real));
    stage_w1k2p(w1k, tw_1024, THREADS);
    const size_t xoff = (size_t)(blockIdx.x / TILES) << L;
    cols_tile2p<L, L1, REV, C>(in + xoff, ws + xoff, blockIdx.x % TILES, plane, w1k, w1k + 1024, threadIdx.x);
}

// ---- pass 2: 16 rows of one transform, written transposed; 16 * T2 threads -------------------------------------
// LDS: plane 16 x N2 floats; wrow = pass 2's thread twiddles [stage][lane] (5 x 32 float2, staged by the caller)
// HM: the fused convolution's forward transform -- every output X[k] leaves multiplied by h_x[k] (same index arithmetic as the store), which
// saves the composition's separate multiply pass over HBM
template <int L, int L1, bool REV, typename C, bool HM = false>
__device__ __forceinline__ void rows_tile2p(const C *ws_x, C *out_x, uint32_t tile, typename w32<C>::real *plane, const C *wrow,
                                            typename w32<C>::real scale, uint32_t t, const C *h_x = nullptr)
{
    using Real = typename w32<C>::real;
    constexpr uint32_t ES = (uint32_t)sizeof(C);
    constexpr int L2 = L - L1, N1 = 1 << L1, N2 = 1 << L2, T2 = N2 / 32;

    // first register pass: T2 lanes run along a row; element n2 = ua + T2 k of row k1 = 16 tile + ra lives at
    // [(n2 >> 4)][k1][n2 & 15] of the intermediate
    const uint32_t ra = t / T2, ua = t % T2;
    const C *src = ws_x + (size_t)tile * (kTile * kTile);
    C x[32];
    if constexpr (T2 == 32) {
        const uint32_t aoff = ((ua >> 4) * (N1 * kTile) + ra * 16 + (ua & 15)) * ES;
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k] = *at(src + (size_t)2 * k * (N1 * kTile), aoff);
    } else if constexpr (T2 == 16) {
        const uint32_t aoff = (ra * 16 + ua) * ES;
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k] = *at(src + (size_t)k * (N1 * kTile), aoff);
    } else {
        static_assert(T2 == 8, "row length 256, 512 or 1024");
        const uint32_t aoff = (ra * 16 + ua) * ES;
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k] = *at(src + (size_t)(k >> 1) * (N1 * kTile) + 8 * (k & 1), aoff);
    }
    __syncthreads(); // wrow staged
    fft32_dif<REV, true, 0, true>(x, wrow + ua, 32);

    // exchange, and switch the thread mapping so that 16 lanes run across the 16 rows:
    //   slot(row, pos) = row * N2 + (pos ^ (mask(row) | (((pos >> 5) & 1) << 4))), mask(row) = row (rows of 256: row rotated right by one)
    // write pos = ua + T2 k of row ra; read pos = 32 ub + k of row rb
    const uint32_t rb = t & 15, ub = t >> 4; // ub < T2
    {
        Real *wb0, *wb1; // bases of the writes; the per-k part is a compile-time offset
        if constexpr (T2 == 32) { // low 5 bits of pos = ua; bit 5 = k & 1
            wb0 = plane + ra * N2 + (ua ^ ra);
            wb1 = plane + ra * N2 + (ua ^ ra ^ 16);
        } else if constexpr (T2 == 16) { // low 4 bits ua, bit 4 = k & 1, bit 5 = (k >> 1) & 1
            wb0 = wb1 = plane + ra * N2 + (ua ^ ra);
        } else { // low 3 bits ua, bit 3 = k & 1, bit 4 = (k >> 1) & 1, bit 5 = (k >> 2) & 1
            // rows of 256: a write instruction's lanes span 2 (b64) / 4 (b32) consecutive rows of 8 positions each, so the row's XOR
            // mask is its index ROTATED (row >> 1 | (row & 1) << 3): rows 2m and 2m + 1 then differ in bit 3 and fill different
            // halves of the 16 slots -- conflict-free ds_write_b64 (was 2-way: 38 % LDS conflict cycles in double,
            // profiles/r03_lds_bank_conflicts.md), 2-way = free ds_write_b32 (was 4-way); the reads only need 16 distinct masks
            wb0 = plane + ra * N2 + (ua ^ (ra >> 1)) + 8 * (ra & 1);       // k even: bit 3 = 0 ^ (ra & 1)
            wb1 = plane + ra * N2 + (ua ^ (ra >> 1)) + 8 * (1 - (ra & 1)); // k odd:  bit 3 = 1 ^ (ra & 1)
        }
        const Real *const r_base = plane + rb * N2 + 32 * ub;
        const uint32_t rmask = T2 == 8 ? ((rb >> 1) | ((rb & 1) << 3)) : rb;
        const uint32_t rx = rmask | ((ub & 1) << 4);
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const Real f = half ? x[k].y : x[k].x;
                if constexpr (T2 == 32)
                    ((k & 1) ? wb1 : wb0)[32 * k] = f;
                else if constexpr (T2 == 16)
                    wb0[16 * ((k ^ (k >> 1)) & 1) + 32 * (k >> 1)] = f;
                else
                    ((k & 1) ? wb1 : wb0)[16 * (((k >> 1) ^ (k >> 2)) & 1) + 32 * (k >> 2)] = f;
            }
            __syncthreads();
            {
                uint32_t q = rx;
                asm volatile("" : "+v"(q)); // the 32 XOR'ed addresses are rebuilt, not kept in registers
#pragma unroll
                for (int k = 0; k < 32; k++) {
                    const Real f = r_base[k ^ q];
                    if (half)
                        x[k].y = f;
                    else
                        x[k].x = f;
                }
            }
            if (half == 0)
                __syncthreads();
        }
    }
    fft32_dif<REV, false, 10 - L2>(x, wrow, 0);

    // position 32 ub + k of row k1 holds X[k1 + N1 k2], k2 = (bit_reverse5(k) << (L2 - 5)) | bit_reverse(ub): 16 lanes write
    // 128 contiguous bytes (streaming store of the final result)
    C *dst_tile = out_x + tile * kTile;
    const uint32_t bub = brev_bits<L2 - 5>(ub);
    const uint32_t boff = (bub * N1 + rb) * ES;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if ((k & 7) == 0)
            __builtin_amdgcn_sched_barrier(0);
        C v = x[k];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            v.x *= scale;
            v.y *= scale;
        }
        if constexpr (HM)
            v = cmul(v, *at(h_x + tile * kTile + (size_t)(__brev((uint32_t)k) >> 27) * ((size_t)N1 << (L2 - 5)), boff));
        nt_store(at(dst_tile + (size_t)(__brev((uint32_t)k) >> 27) * ((size_t)N1 << (L2 - 5)), boff), v);
    }
}
// [stage][lane] = W_N2^(lane << stage) = W_1024^((lane 1024/N2) << stage)
template <int L2, typename C> __device__ __forceinline__ void stage_wrow2p(C *wrow, const C *tw_1024, uint32_t threads)
{
    using Real = typename w32<C>::real;
    constexpr int N2 = 1 << L2, T2 = N2 / 32;
    for (uint32_t i = threadIdx.x; i < 5 * 32; i += threads)
        wrow[i] = (i & 31u) < (uint32_t)T2 ? tw_1024[((i & 31u) * (1024 / N2)) << (i >> 5)] : C{ Real(1), Real(0) };
}
template <int L, int L1, bool REV, typename C, bool HM = false>
__global__ __launch_bounds__(kTile * (1 << (L - L1 - 5))) void sdsp_fft2p_rows(const C *__restrict__ ws, C *__restrict__ out,
                                                                              const C *__restrict__ tw_1024, typename w32<C>::real scale,
                                                                              const C *__restrict__ hmul)
{
    using Real = typename w32<C>::real;
    constexpr int L2 = L - L1, N1 = 1 << L1, N2 = 1 << L2, T2 = N2 / 32, THREADS = kTile * T2, TILES = N1 / kTile;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft2p_smem[];
    Real *plane = reinterpret_cast<Real *>(sdsp_fft2p_smem);
    C *wrow = reinterpret_cast<C *>(sdsp_fft2p_smem + (size_t)N2 * kTile * sizeof(Real));
    stage_wrow2p<L2>(wrow, tw_1024, THREADS);
    const size_t xoff = (size_t)(blockIdx.x / TILES) << L;
    rows_tile2p<L, L1, REV, C, HM>(ws + xoff, out + xoff, blockIdx.x % TILES, plane, wrow, scale, threadIdx.x, hmul);
}

template <int L, int L1, bool REV, typename C> int launch_pair(const fft_2pass_args &a, hipStream_t s)
{
    using Real = typename w32<C>::real;
    constexpr int L2 = L - L1, N1 = 1 << L1, N2 = 1 << L2;
    constexpr size_t lds_cols = (size_t)N1 * kTile * sizeof(Real) + 1024 * sizeof(C) + 32 * kTile * sizeof(C);
    constexpr size_t lds_rows = (size_t)N2 * kTile * sizeof(Real) + 5 * 32 * sizeof(C);
    static std::atomic<uint64_t> done_c{ 0 }, done_r{ 0 };
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sdsp_fft2p_cols<L, L1, REV, C>), lds_cols, done_c))
        return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sdsp_fft2p_rows<L, L1, REV, C>), lds_rows, done_r))
        return rc;
    if constexpr (!REV) {
        static std::atomic<uint64_t> done_h{ 0 };
        if (a.hmul)
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sdsp_fft2p_rows<L, L1, false, C, true>), lds_rows, done_h))
                return rc;
    }
    const uint64_t blocks_c = a.count * (N2 / kTile), blocks_r = a.count * (N1 / kTile);
    if (blocks_c > 0x7fffffffull || blocks_r > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "chunk too large for one launch");
    C *d = reinterpret_cast<C *>(a.data), *ws = reinterpret_cast<C *>(a.workspace);
    const C *tw = reinterpret_cast<const C *>(a.tw_1024);
    const Real scale = sizeof(Real) == 8 ? (Real)a.scale_d : (Real)a.scale;
    hipLaunchKernelGGL((sdsp_fft2p_cols<L, L1, REV, C>), dim3((uint32_t)blocks_c), dim3(kTile * (N1 / 32)), lds_cols, s, d, ws, tw);
    const C *hm = reinterpret_cast<const C *>(a.hmul);
    if constexpr (!REV) {
        if (hm)
            hipLaunchKernelGGL((sdsp_fft2p_rows<L, L1, false, C, true>), dim3((uint32_t)blocks_r), dim3(kTile * (N2 / 32)), lds_rows, s, ws, d, tw, scale, hm);
        else
            hipLaunchKernelGGL((sdsp_fft2p_rows<L, L1, REV, C>), dim3((uint32_t)blocks_r), dim3(kTile * (N2 / 32)), lds_rows, s, ws, d, tw, scale, hm);
    } else {
        hipLaunchKernelGGL((sdsp_fft2p_rows<L, L1, REV, C>), dim3((uint32_t)blocks_r), dim3(kTile * (N2 / 32)), lds_rows, s, ws, d, tw, scale, hm);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_2pass launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

// ================================================================================================================
// Round 3: a factor of 2048 -- N = 2^21 = 1024 x 2048 and N = 2^22 = 2048 x 2048 in TWO passes (were four / five nested ones).
// A 2048-point sequence does not fit "32 points per thread, two register passes" (eleven stages); it is held by 32 threads
// with SIXTY-FOUR points each: one top stage (span 1024) pairs register k with register k + 32 and owes the lower output
// W_2048^(u + 32 k) = the thread's W_2048^u x the literal W_64^k; the two halves are then 1024-point sequences in exactly the
// T = 32 layout of the kernels above (fft32_dif with thread twiddles W_1024^(u << s)); ONE exchange regroups to 64
// consecutive positions per thread, and two constants-only fft32_dif finish.  128 data VGPRs, 512-thread workgroups (16
// sequences), a 128 KiB plane: one workgroup per CU.  Index arithmetic replayed in numpy first: tools/model_fft2pass.py (run64).
__device__ constexpr float kC64[32] = { 1.00000000000000000000f, 0.99518472667219692873f, 0.98078528040323043058f, 0.95694033573220882438f, 0.92387953251128673848f, 0.88192126434835504956f, 0.83146961230254523567f, 0.77301045336273699338f, 0.70710678118654757274f, 0.63439328416364548779f, 0.55557023301960228867f, 0.47139673682599780857f, 0.38268343236508983729f, 0.29028467725446233105f, 0.19509032201612833135f, 0.09801714032956077016f, 0.00000000000000006123f, -0.09801714032956064526f, -0.19509032201612819257f, -0.29028467725446216452f, -0.38268343236508972627f, -0.47139673682599769755f, -0.55557023301960195560f, -0.63439328416364537677f, -0.70710678118654746172f, -0.77301045336273699338f, -0.83146961230254534669f, -0.88192126434835493853f, -0.92387953251128673848f, -0.95694033573220882438f, -0.98078528040323043058f, -0.99518472667219681771f };
__device__ constexpr float kS64[32] = { 0.00000000000000000000f, 0.09801714032956060363f, 0.19509032201612824808f, 0.29028467725446233105f, 0.38268343236508978178f, 0.47139673682599764204f, 0.55557023301960217765f, 0.63439328416364548779f, 0.70710678118654746172f, 0.77301045336273699338f, 0.83146961230254523567f, 0.88192126434835493853f, 0.92387953251128673848f, 0.95694033573220893540f, 0.98078528040323043058f, 0.99518472667219681771f, 1.00000000000000000000f, 0.99518472667219692873f, 0.98078528040323043058f, 0.95694033573220893540f, 0.92387953251128673848f, 0.88192126434835504956f, 0.83146961230254545772f, 0.77301045336273710440f, 0.70710678118654757274f, 0.63439328416364548779f, 0.55557023301960217765f, 0.47139673682599786408f, 0.38268343236508989280f, 0.29028467725446238656f, 0.19509032201612860891f, 0.09801714032956082567f };
constexpr float kC2048 = 0.99999529380957617151f, kS2048 = 0.00306795676296597627f; // cos / sin of 2 pi / 2048

// W_2048^u from the table W_1024^j (direction-folded): W_1024^(u >> 1), times W_2048^1 for odd u
template <bool REV> __device__ __forceinline__ float2 w2048(const float2 *w1k, uint32_t u)
{
    const float2 w = w1k[u >> 1];
    const float2 odd = cmul(w, float2{ kC2048, REV ? kS2048 : -kS2048 });
    return (u & 1u) ? odd : w;
}

template <bool REV> __device__ __forceinline__ void top_stage64(float2 (&lo)[32], float2 (&hi)[32], float2 wu)
{
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const float2 a = lo[k], b = hi[k];
        lo[k] = a + b;
        float2 d = a - b;
        if (k == 16) {
            d = REV ? float2{ -d.y, d.x } : float2{ d.y, -d.x };
        } else if (k != 0) {
            const float cr = kC64[k], ci = REV ? kS64[k] : -kS64[k];
            d = float2{ d.x * cr - d.y * ci, d.x * ci + d.y * cr };
        }
        hi[k] = cmul(d, wu);
    }
}

// ---- pass 1 with N1 = 2048: 16 columns, 512 threads.  LDS: plane 2048 x 16 floats (slot = (row * 16 + col) ^ (((row >> 6) & 1) << 4)),
// w1k = W_1024 (8 KiB), qtab 64 x 16 float2 (8 KiB)
template <int L, bool REV, bool WT = false>
__device__ __forceinline__ void cols64_tile2p(const float2 *in_x, float2 *ws_x, uint32_t tile, float *plane, const float2 *w1k, float2 *qtab)
{
    constexpr int L1 = 11, L2 = L - L1, N1 = 1 << L1, N2 = 1 << L2;
    const uint32_t t = threadIdx.x;
    const uint32_t c = t & 15, u = t >> 4; // u < 32
    const uint32_t n2 = tile * kTile + c;

    const float2 *src_tile = in_x + tile * kTile;
    const uint32_t toff = (u * N2 + c) * 8u;
    float2 lo[32], hi[32]; // rows u + 32 k and 1024 + u + 32 k
#pragma unroll
    for (int k = 0; k < 32; k++)
        lo[k] = nt_load(at(src_tile + (size_t)32 * N2 * k, toff));
#pragma unroll
    for (int k = 0; k < 32; k++)
        hi[k] = nt_load(at(src_tile + (size_t)32 * N2 * (k + 32), toff));
    __syncthreads(); // w1k staged

    // the column part of the inter-pass twiddle: W_N^(n2 * jj * 32), jj < 64; two entries per thread
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t jj = u + 32 * i;
        qtab[jj * 16 + c] = twiddle_n<L, REV>(w1k, (n2 * jj) << 5);
    }
    top_stage64<REV>(lo, hi, w2048<REV>(w1k, u));
    fft32_dif<REV, true>(lo, w1k, u);
    fft32_dif<REV, true>(hi, w1k, u);

    // exchange rows {1024 h + u + 32 k} -> {64 u + j}
    {
        float *const w0 = plane + (u * 16 + c);
        float *const w1 = plane + ((u * 16 + c) ^ 16);
        const int flip = (int)(u & 1) * 16;
        const float *const r_even = plane + (1024 * u + c) + flip;
        const float *const r_odd = plane + (1024 * u + c) - flip;
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int k = 0; k < 32; k++) { // bit 6 of the row = bit 1 of k
                (((k >> 1) & 1) ? w1 : w0)[512 * k] = half ? lo[k].y : lo[k].x;
                (((k >> 1) & 1) ? w1 : w0)[512 * k + 16384] = half ? hi[k].y : hi[k].x;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const float f = ((j & 1) ? r_odd : r_even)[16 * j];
                const float g = ((j & 1) ? r_odd : r_even)[16 * (j + 32)];
                if (half) {
                    lo[j].y = f;
                    hi[j].y = g;
                } else {
                    lo[j].x = f;
                    hi[j].x = g;
                }
            }
            if (half == 0)
                __syncthreads();
        }
    }
    fft32_dif<REV, false, 0>(lo, w1k, 0);
    fft32_dif<REV, false, 0>(hi, w1k, 0);

    // row 64 u + j holds Y[k1], k1 = bit_reverse11(64 u + j) = 32 * jj + bit_reverse5(u), jj = (bit_reverse5(j & 31) << 1) | (j >> 5)
    const uint32_t bu = brev_bits<5>(u);
    const float2 pw = twiddle_n<L, REV>(w1k, n2 * bu);
    const float2 *const qcol = qtab + c;
    float2 *dst_tile = ws_x + (size_t)tile * (N1 * kTile); // [tile][k1][c]
    const uint32_t soff = (bu * 16 + c) * 8u;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if ((j & 7) == 0)
            __builtin_amdgcn_sched_barrier(0);
        const int jj = (int)(__brev((uint32_t)j) >> 27) << 1;
        const float2 v = cmul(lo[j], cmul(pw, qcol[16 * jj])), w = cmul(hi[j], cmul(pw, qcol[16 * (jj + 1)]));
        if constexpr (WT) {
            handoff::wt_store(at(dst_tile + (size_t)jj * 512, soff), v);
            handoff::wt_store(at(dst_tile + (size_t)(jj + 1) * 512, soff), w);
        } else {
            *at(dst_tile + (size_t)jj * 512, soff) = v;
            *at(dst_tile + (size_t)(jj + 1) * 512, soff) = w;
        }
    }
}
template <int L, bool REV>
__global__ __launch_bounds__(512) void sdsp_fft2p_cols64(const float2 *__restrict__ in, float2 *__restrict__ ws, const float2 *__restrict__ tw_1024)
{
    constexpr int N1 = 2048, N2 = 1 << (L - 11), TILES = N2 / kTile;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft2p_smem[];
    float *plane = reinterpret_cast<float *>(sdsp_fft2p_smem);
    float2 *w1k = reinterpret_cast<float2 *>(sdsp_fft2p_smem + (size_t)N1 * kTile * sizeof(float));
    stage_w1k2p(w1k, tw_1024, 512);
    const size_t xoff = (size_t)(blockIdx.x / TILES) << L;
    cols64_tile2p<L, REV>(in + xoff, ws + xoff, blockIdx.x % TILES, plane, w1k, w1k + 1024);
}

// ---- pass 2 with N2 = 2048: 16 rows, written transposed; 512 threads.  LDS: plane 16 x 2048 floats
// (slot(row, pos) = row * 2048 + (pos ^ (row | (((pos >> 6) & 1) << 4)))), wrow = [stage < 5][lane] W_1024^(lane << stage), [5][lane] W_2048^lane
template <int L, int L1, bool REV, bool HM = false>
__device__ __forceinline__ void rows64_tile2p(const float2 *ws_x, float2 *out_x, uint32_t tile, float *plane, const float2 *wrow, float scale,
                                              const float2 *h_x = nullptr)
{
    constexpr int N1 = 1 << L1, N2 = 2048;
    static_assert(L - L1 == 11, "rows of 2048");
    const uint32_t t = threadIdx.x;
    const uint32_t ra = t >> 5, ua = t & 31;
    const float2 *src = ws_x + (size_t)tile * (kTile * kTile);
    // element n2 = ua + 32 k of row k1 = 16 tile + ra lives at [(n2 >> 4)][k1][n2 & 15] of the intermediate
    const uint32_t aoff = ((ua >> 4) * (N1 * kTile) + ra * 16 + (ua & 15)) * 8u;
    float2 lo[32], hi[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        lo[k] = *at(src + (size_t)2 * k * (N1 * kTile), aoff);
#pragma unroll
    for (int k = 0; k < 32; k++)
        hi[k] = *at(src + (size_t)2 * (k + 32) * (N1 * kTile), aoff);
    __syncthreads(); // wrow staged
    top_stage64<REV>(lo, hi, wrow[5 * 32 + ua]);
    fft32_dif<REV, true, 0, true>(lo, wrow + ua, 32);
    fft32_dif<REV, true, 0, true>(hi, wrow + ua, 32);

    // exchange, and switch the thread mapping so that 16 lanes run across the 16 rows: write pos = 1024 h + ua + 32 k of row
    // ra; read pos = 64 ub + j of row rb
    const uint32_t rb = t & 15, ub = t >> 4; // ub < 32
    {
        float *const wb0 = plane + ra * N2 + (ua ^ ra);
        float *const wb1 = plane + ra * N2 + (ua ^ ra ^ 16);
        const float *const r_base = plane + rb * N2 + 64 * ub;
        const uint32_t rx = rb | ((ub & 1) << 4);
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int k = 0; k < 32; k++) { // bit 6 of pos = bit 1 of k
                (((k >> 1) & 1) ? wb1 : wb0)[32 * k] = half ? lo[k].y : lo[k].x;
                (((k >> 1) & 1) ? wb1 : wb0)[32 * k + 1024] = half ? hi[k].y : hi[k].x;
            }
            __syncthreads();
            {
                uint32_t q = rx;
                asm volatile("" : "+v"(q)); // the 64 XOR'ed addresses are rebuilt, not kept in registers
#pragma unroll
                for (int j = 0; j < 32; j++) {
                    const float f = r_base[j ^ q];
                    const float g = r_base[(j + 32) ^ q];
                    if (half) {
                        lo[j].y = f;
                        hi[j].y = g;
                    } else {
                        lo[j].x = f;
                        hi[j].x = g;
                    }
                }
            }
            if (half == 0)
                __syncthreads();
        }
    }
    fft32_dif<REV, false, 0>(lo, wrow, 0);
    fft32_dif<REV, false, 0>(hi, wrow, 0);

    // position 64 ub + j of row k1 holds X[k1 + N1 k2], k2 = (bit_reverse5(j & 31) << 6) | ((j >> 5) << 5) | bit_reverse5(ub): 16 lanes
    // write 128 contiguous bytes (streaming store of the final result)
    float2 *dst_tile = out_x + tile * kTile;
    const uint32_t bub = brev_bits<5>(ub);
    const uint32_t boff = (bub * N1 + rb) * 8u;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if ((j & 7) == 0)
            __builtin_amdgcn_sched_barrier(0);
        float2 v = lo[j], w = hi[j];
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            v.x *= scale;
            v.y *= scale;
            w.x *= scale;
            w.y *= scale;
        }
        const size_t k2hi = (size_t)(__brev((uint32_t)j) >> 27) * 64;
        if constexpr (HM) {
            v = cmul(v, *at(h_x + tile * kTile + k2hi * N1, boff));
            w = cmul(w, *at(h_x + tile * kTile + (k2hi + 32) * N1, boff));
        }
        nt_store(at(dst_tile + k2hi * N1, boff), v);
        nt_store(at(dst_tile + (k2hi + 32) * N1, boff), w);
    }
}
template <int L, int L1, bool REV, bool HM = false>
__global__ __launch_bounds__(512) void sdsp_fft2p_rows64(const float2 *__restrict__ ws, float2 *__restrict__ out, const float2 *__restrict__ tw_1024,
                                                        float scale, const float2 *__restrict__ hmul)
{
    constexpr int N1 = 1 << L1, N2 = 2048, TILES = N1 / kTile;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft2p_smem[];
    float *plane = reinterpret_cast<float *>(sdsp_fft2p_smem);
    float2 *wrow = reinterpret_cast<float2 *>(sdsp_fft2p_smem + (size_t)N2 * kTile * sizeof(float));
    for (uint32_t i = threadIdx.x; i < 6 * 32; i += 512) // [stage][lane] = W_1024^(lane << stage); [5][lane] = W_2048^lane
        wrow[i] = i < 5 * 32 ? tw_1024[(i & 31u) << (i >> 5)] : w2048<REV>(tw_1024, i & 31u);
    const size_t xoff = (size_t)(blockIdx.x / TILES) << L;
    rows64_tile2p<L, L1, REV, HM>(ws + xoff, out + xoff, blockIdx.x % TILES, plane, wrow, scale, hmul);
}

// N = 2^21: the 32-point column pass above (N1 = 1024) + rows of 2048; N = 2^22: both passes of 2048
template <int L, bool REV> int launch_pair64(const fft_2pass_args &a, hipStream_t s)
{
    constexpr int L1 = L == 21 ? 10 : 11, N1 = 1 << L1, N2 = 2048;
    constexpr size_t lds_cols = L1 == 10 ? (size_t)N1 * kTile * 4 + 1024 * 8 + 32 * kTile * 8 : (size_t)2048 * kTile * 4 + 1024 * 8 + 64 * kTile * 8;
    constexpr size_t lds_rows = (size_t)N2 * kTile * 4 + 6 * 32 * 8;
    static std::atomic<uint64_t> done_c{ 0 }, done_r{ 0 };
    const void *kc = L1 == 10 ? reinterpret_cast<const void *>(sdsp_fft2p_cols<L, 10, REV, float2>) : reinterpret_cast<const void *>(sdsp_fft2p_cols64<22, REV>);
    if (int rc = ensure_dynamic_lds(kc, lds_cols, done_c))
        return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sdsp_fft2p_rows64<L, L1, REV>), lds_rows, done_r))
        return rc;
    if constexpr (!REV) {
        static std::atomic<uint64_t> done_h{ 0 };
        if (a.hmul)
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sdsp_fft2p_rows64<L, L1, false, true>), lds_rows, done_h))
                return rc;
    }
    const uint64_t blocks_c = a.count * (N2 / kTile), blocks_r = a.count * (N1 / kTile);
    if (blocks_c > 0x7fffffffull || blocks_r > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "chunk too large for one launch");
    float2 *d = reinterpret_cast<float2 *>(a.data), *ws = reinterpret_cast<float2 *>(a.workspace);
    const float2 *tw = reinterpret_cast<const float2 *>(a.tw_1024);
    if constexpr (L1 == 10)
        hipLaunchKernelGGL((sdsp_fft2p_cols<L, 10, REV, float2>), dim3((uint32_t)blocks_c), dim3(kTile * (N1 / 32)), lds_cols, s, d, ws, tw);
    else
        hipLaunchKernelGGL((sdsp_fft2p_cols64<22, REV>), dim3((uint32_t)blocks_c), dim3(512), lds_cols, s, d, ws, tw);
    const float2 *hm = reinterpret_cast<const float2 *>(a.hmul);
    if constexpr (!REV) {
        if (hm)
            hipLaunchKernelGGL((sdsp_fft2p_rows64<L, L1, false, true>), dim3((uint32_t)blocks_r), dim3(512), lds_rows, s, ws, d, tw, a.scale, hm);
        else
            hipLaunchKernelGGL((sdsp_fft2p_rows64<L, L1, REV>), dim3((uint32_t)blocks_r), dim3(512), lds_rows, s, ws, d, tw, a.scale, hm);
    } else {
        hipLaunchKernelGGL((sdsp_fft2p_rows64<L, L1, REV>), dim3((uint32_t)blocks_r), dim3(512), lds_rows, s, ws, d, tw, a.scale, hm);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_2pass launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
// ================================================================================================================
// Round 3, second part: the two passes in ONE persistent, ticketed launch -- the schedule of sdsp_fft1m_fused
// (fft1m_kernels.h) over the tiles above, for every two-pass size.  A ticket step covers one UNIT = `unit` consecutive
// transforms (8 MiB of data where a transform is smaller than that): its pass-1 items, then the pass-2 items of the unit
// `lag` steps behind in the same queue.  A unit's intermediates live in slot (q * ring + i % ring) of the workspace, so the
// intermediates in flight stay inside the Infinity Cache (queues x ring x unit bytes <= 256 MiB) and are consumed a few
// microseconds after they were produced, instead of after a whole chunk's pass 1 (fft2p_chunk: 256 MiB written, then read).
// Where the two passes have different thread counts (N1 != N2) the workgroup has the larger one and the shorter pass runs
// G tiles side by side (whole waves each; the barriers inside a tile function are workgroup barriers either way).
// Hand-off, synchronisation words, bounded polls, abort / sticky words: handoff.h, exactly as in sdsp_fft1m_fused.
struct fused2p_kargs {
    void *data;          // count transforms, in place
    void *ws;            // queues x ring x unit transforms
    const void *tw_1024;
    unsigned *sync;
    unsigned *sticky;    // nullable
    uint32_t count, unit, ring, lag, queues;
    uint32_t flags;      // 8 = fault injection (tests): every hand-off wait gives up at once
    float scale;
    double scale_d;
    unsigned long long spin_limit;
    const void *hmul;    // HM instances: the frequency response the forward transform's outputs are multiplied by (fused convolution)
};

// the thread index as a value the compiler cannot prove loop-invariant: the per-thread addresses of BOTH tile functions would
// otherwise be hoisted out of the persistent loop and stay live across every tile (16 - 64 bytes of scratch per lane in f32)
__device__ __forceinline__ uint32_t fresh_tid()
{
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
// what the persistent kernel needs to know about a size: both passes on 32 points per thread (cols_tile2p / rows_tile2p)
// MIN_THREADS: the least workgroup size -- f32 N = 2^16 (two 128-thread passes) takes 256, i.e. two tiles side by side in BOTH passes:
// half as many items, each as large as N = 2^17's
template <int L_, int L1_, typename C_, int MIN_THREADS = 64> struct shape32 {
    using C = C_;
    using Real = typename w32<C>::real;
    static constexpr int L = L_, L1 = L1_, L2 = L - L1, N1 = 1 << L1, N2 = 1 << L2;
    static constexpr int THREADS_C = kTile * (N1 / 32), THREADS_R = kTile * (N2 / 32);
    static constexpr int THREADS_P = THREADS_C > THREADS_R ? THREADS_C : THREADS_R;
    static constexpr int THREADS = THREADS_P > MIN_THREADS ? THREADS_P : MIN_THREADS;
    static constexpr int GC = THREADS / THREADS_C, GR = THREADS / THREADS_R; // tiles side by side in one item
    static constexpr int MIN_WAVES = sizeof(Real) == 4 ? 4 : 2; // per SIMD: <= 128 VGPRs in f32 (as the two-launch kernels), <= 256 in double
    static constexpr int ITEMS_C = (N2 / kTile) / GC, ITEMS_R = (N1 / kTile) / GR; // per transform
    static constexpr size_t PLANE = (size_t)32 * THREADS * sizeof(Real); // G tiles of 32 points per thread, either pass
    // LDS: plane | W_1024 | qtab (pass 1's column twiddles, G tiles).  Pass 2's thread twiddles (1.25 / 2.5 KiB) are rebuilt from
    // the W_1024 copy into the qtab area by every pass-2 item (qtab is dead then; the barrier inside rows_tile2p covers it), and
    // the ticket mailbox lies over the plane's first 16 bytes -- the plane is dead between items, a tile function's own barrier
    // separates the workgroup's read of the ticket from the first plane write, and the barrier that ends every item separates
    // the last plane read from the next ticket's write.  That keeps N = 2^19 f32 at 80 KiB (two workgroups per CU) and fits
    // N = 2^19 in double into the CU's 160 KiB.
    static constexpr size_t W1K = PLANE, QTAB = W1K + 1024 * sizeof(C), WROW = QTAB, MAIL = 0;
    static constexpr size_t LDS = QTAB + (size_t)GC * 32 * kTile * sizeof(C);
    static_assert(5 * 32 * sizeof(C) <= (size_t)GC * 32 * kTile * sizeof(C), "pass 2's thread twiddles fit the qtab area");
    template <bool REV> static __device__ __forceinline__ void stage(unsigned char *smem, const C *tw_1024)
    {
        stage_w1k2p(reinterpret_cast<C *>(smem + W1K), tw_1024, THREADS);
    }
    template <bool REV> static __device__ __forceinline__ void cols(const C *in_x, C *ws_x, uint32_t item, unsigned char *smem)
    {
        const uint32_t tid = fresh_tid();
        const uint32_t g = tid / THREADS_C, t = tid % THREADS_C;
        cols_tile2p<L, L1, REV, C, true>(in_x, ws_x, item * GC + g, reinterpret_cast<Real *>(smem) + (size_t)g * N1 * kTile,
                                         reinterpret_cast<const C *>(smem + W1K), reinterpret_cast<C *>(smem + QTAB) + g * 32 * kTile, t);
    }
    template <bool REV, bool HM>
    static __device__ __forceinline__ void rows(const C *ws_x, C *out_x, uint32_t item, unsigned char *smem, Real scale, const C *h_x)
    {
        const uint32_t tid = fresh_tid();
        const uint32_t g = tid / THREADS_R, t = tid % THREADS_R;
        stage_wrow2p<L2>(reinterpret_cast<C *>(smem + WROW), reinterpret_cast<const C *>(smem + W1K), THREADS);
        rows_tile2p<L, L1, REV, C, HM>(ws_x, out_x, item * GR + g, reinterpret_cast<Real *>(smem) + (size_t)g * N2 * kTile,
                                       reinterpret_cast<const C *>(smem + WROW), scale, t, h_x);
    }
};
// N = 2^21 (1024 x 2048) and 2^22 (2048 x 2048), f32: the 64-points-per-thread passes
template <int L_> struct shape64 {
    using C = float2;
    using Real = float;
    static constexpr int L = L_, L1 = L == 21 ? 10 : 11, N1 = 1 << L1, N2 = 2048;
    static constexpr int THREADS = 512, ITEMS_C = N2 / kTile, ITEMS_R = N1 / kTile, MIN_WAVES = 2;
    static constexpr size_t PLANE = (size_t)2048 * kTile * 4;
    static constexpr size_t W1K = PLANE, QTAB = W1K + 1024 * 8, WROW = QTAB + (size_t)64 * kTile * 8;
    static constexpr size_t MAIL = WROW + 6 * 32 * 8, LDS = MAIL + 16;
    template <bool REV> static __device__ __forceinline__ void stage(unsigned char *smem, const float2 *tw_1024)
    {
        stage_w1k2p(reinterpret_cast<float2 *>(smem + W1K), tw_1024, THREADS);
        float2 *wrow = reinterpret_cast<float2 *>(smem + WROW);
        for (uint32_t i = threadIdx.x; i < 6 * 32; i += THREADS)
            wrow[i] = i < 5 * 32 ? tw_1024[(i & 31u) << (i >> 5)] : w2048<REV>(tw_1024, i & 31u);
    }
    template <bool REV> static __device__ __forceinline__ void cols(const float2 *in_x, float2 *ws_x, uint32_t item, unsigned char *smem)
    {
        float *plane = reinterpret_cast<float *>(smem);
        const float2 *w1k = reinterpret_cast<const float2 *>(smem + W1K);
        float2 *qtab = reinterpret_cast<float2 *>(smem + QTAB);
        if constexpr (L1 == 10)
            cols_tile2p<L, 10, REV, float2, true>(in_x, ws_x, item, plane, w1k, qtab, fresh_tid());
        else
            cols64_tile2p<L, REV, true>(in_x, ws_x, item, plane, w1k, qtab);
    }
    template <bool REV, bool HM>
    static __device__ __forceinline__ void rows(const float2 *ws_x, float2 *out_x, uint32_t item, unsigned char *smem, float scale, const float2 *h_x)
    {
        rows64_tile2p<L, L1, REV, HM>(ws_x, out_x, item, reinterpret_cast<float *>(smem), reinterpret_cast<const float2 *>(smem + WROW), scale, h_x);
    }
};

template <class S, bool REV, bool HM = false> __global__ __launch_bounds__(S::THREADS, S::MIN_WAVES) void sdsp_fft2p_fused(fused2p_kargs a)
{
    using C = typename S::C;
    using Real = typename S::Real;
    using handoff::ld_relaxed;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft2p_smem[];
    // mailbox: [0], [1] the item's ticket, alternating per iteration (an iteration without work has no barrier between the
    // other waves' read of its ticket and lane 0's write of the next one); [2] go / abort of the item's wait
    unsigned *mail = reinterpret_cast<unsigned *>(sdsp_fft2p_smem + S::MAIL);

    const uint32_t n_units = (a.count + a.unit - 1) / a.unit;
    const uint32_t q = blockIdx.x % a.queues;
    if (q >= n_units)
        return;
    const uint32_t n_q = (n_units - q + a.queues - 1) / a.queues; // units of this queue
    unsigned *const ticket_ctr = a.sync + 32 * q, *const abort_flag = a.sync + 32 * a.queues;
    unsigned *const done = a.sync + 32 * (a.queues + 1);
    const unsigned items_c = a.unit * S::ITEMS_C, items_r = a.unit * S::ITEMS_R, per_step = items_c + items_r;
    const unsigned n_tickets = (n_q + a.lag) * per_step;

    S::template stage<REV>(sdsp_fft2p_smem, reinterpret_cast<const C *>(a.tw_1024));
    unsigned next = 0;
    if (threadIdx.x == 0)
        next = __hip_atomic_fetch_add(ticket_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    for (unsigned it = 0;; it++) {
        // ---- this item's ticket (drawn one item ahead, so the atomic's latency hides behind the previous tile).  A workgroup
        // only ever waits for LOWER tickets of its queue, each of which is held by a running workgroup or done: the grid drains
        // whatever its size and the dispatch order.  (Drawing blocks of 2 / 4 / 8 consecutive tickets per atomic measured 1 - 2 /
        // 5 - 7 / 10 - 14 points slower at every size: a block keeps its later tickets waiting behind its first.)
        if (threadIdx.x == 0)
            mail[it & 1] = next;
        __syncthreads(); // also: every wave has finished the previous item (planes / tables are free), tables are staged
        const unsigned ticket = mail[it & 1];
        if (ticket >= n_tickets)
            break;
        if (threadIdx.x == 0)
            next = __hip_atomic_fetch_add(ticket_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned step = ticket / per_step, sub = ticket % per_step;
        const bool first = sub < items_c;
        const unsigned i = first ? step : step - a.lag; // wraps for the leading pass-2 slots: filtered below
        if (i >= n_q)
            continue; // ramp-up / ramp-down slot without work (uniform)
        const unsigned item = first ? sub : sub - items_c;
        const unsigned x_in = item / (first ? S::ITEMS_C : S::ITEMS_R), item_x = item % (first ? S::ITEMS_C : S::ITEMS_R);
        const unsigned unit_id = q + i * a.queues;
        const unsigned xf = unit_id * a.unit + x_in;
        if (xf >= a.count)
            continue; // the batch's last unit may be short (its arrival target below counts what exists)
        const unsigned cnt = min(a.unit, a.count - unit_id * a.unit);
        C *const ws_x = reinterpret_cast<C *>(a.ws) + (((size_t)(q * a.ring + i % a.ring) * a.unit + x_in) << S::L);
        C *const data_x = reinterpret_cast<C *>(a.data) + ((size_t)xf << S::L);

        // ---- wait for what this item depends on (ONE lane polls; ONE acquire for the workgroup)
        unsigned *wait_word = nullptr;
        unsigned target = 0;
        if (!first) {
            wait_word = done + 32 * unit_id; // the whole intermediate of this unit
            target = cnt * S::ITEMS_C;
        } else if (i >= a.ring) {
            wait_word = done + 32 * (unit_id - a.ring * a.queues) + 16; // the ring slot's previous tenant (a full unit) has been read
            target = a.unit * S::ITEMS_R;
        }
        if (wait_word) {
            if (threadIdx.x == 0) {
                bool ok;
                if (a.flags & 8u) { // injected fault: behave exactly like a poll whose bound expired
                    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (a.sticky)
                        __hip_atomic_store(a.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                } else {
                    ok = handoff::poll_geq(wait_word, target, abort_flag, a.sticky, a.spin_limit, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                mail[2] = ok ? 1u : 0u;
            }
            __syncthreads();
            if (mail[2] == 0u)
                break; // aborted: drain (uniform)
        }

        if (first)
            S::template cols<REV>(data_x, ws_x, item_x, sdsp_fft2p_smem);
        else
            S::template rows<REV, HM>(ws_x, data_x, item_x, sdsp_fft2p_smem, sizeof(Real) == 8 ? (Real)a.scale_d : (Real)a.scale,
                                      reinterpret_cast<const C *>(a.hmul));

        // ---- publish: pass 1 hands its output to other workgroups (write-through stores: drained = at the fabric); pass 2
        // frees the ring slot -- its loads of the slot have returned (their values fed the butterflies)
        if (first) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_fetch_add(done + 32 * unit_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_fetch_add(done + 32 * unit_id + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <class S, bool REV, bool HM = false> int launch_fused_s(const fft_2pass_fused_args &a, hipStream_t s)
{
    auto kern = sdsp_fft2p_fused<S, REV, HM>;
    static std::atomic<uint64_t> lds_done{ 0 };
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), S::LDS, lds_done))
        return rc;
    // as many workgroups as are resident (a larger grid would be correct too: tickets are drawn by running workgroups only);
    // per device: a node may mix parts with different CU counts
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, "hipGetDevice failed");
    int grid = cached[dev & 63].load();
    if (!grid) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, S::THREADS, S::LDS) != hipSuccess || per_cu < 1 ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return fail(SDSP_HIP_ERR_HIP, "fft_2pass fused: occupancy query failed");
        grid = per_cu * prop.multiProcessorCount;
        cached[dev & 63].store(grid);
    }
    const uint64_t units = (a.count + a.unit - 1) / a.unit;
    fused2p_kargs k;
    k.data = a.data;
    k.ws = a.workspace;
    k.tw_1024 = a.tw_1024;
    k.sync = reinterpret_cast<unsigned *>(a.sync);
    k.sticky = reinterpret_cast<unsigned *>(a.sticky);
    k.count = (uint32_t)a.count;
    k.unit = a.unit;
    k.ring = a.ring;
    k.lag = a.lag;
    k.queues = a.queues;
    k.flags = a.spin_limit == 0 ? 8u : 0u;
    k.scale = a.scale;
    k.scale_d = a.scale_d;
    k.spin_limit = a.spin_limit;
    k.hmul = a.hmul;
    hipError_t e = hipMemsetAsync(a.sync, 0, fft_2pass_sync_bytes(units, a.queues), s);
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_2pass fused memset: ") + hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(S::THREADS), S::LDS, s, k);
    e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft_2pass fused launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
template <class S> int launch_fused_dir(const fft_2pass_fused_args &a, hipStream_t s)
{
    if (a.hmul && a.reverse)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "fft_2pass: the multiply rides on the forward transform");
    return a.reverse ? launch_fused_s<S, true>(a, s) : a.hmul ? launch_fused_s<S, false, true>(a, s) : launch_fused_s<S, false>(a, s);
}

template <int L, int L1, typename C> int launch_dir(const fft_2pass_args &a, hipStream_t s)
{
    return a.reverse ? launch_pair<L, L1, true, C>(a, s) : launch_pair<L, L1, false, C>(a, s);
}
} // namespace

// f32: N = 2^16 .. 2^19 and 2^21, 2^22 (2^20 has the persistent kernel of fft1m.hip); f64: N = 2^15 .. 2^20 (2^15 = 128 x 256: columns of
// 128 points on FOUR threads each -- one wave per tile; f32 N = 2^15 is a single-pass size, fft_big.hip)
bool fft_2pass_supports(uint32_t n, int precision)
{
    if (!sdsp_hip_is_power_of_2(n) || n < (1u << 15))
        return false;
    if (precision == SDSP_HIP_F64)
        return n <= (1u << 20);
    if (n < (1u << 16))
        return false;
    return n <= (1u << 19) || n == (1u << 21) || n == (1u << 22);
}

// both passes over one chunk of `count` transforms (the workspace holds `count` intermediates)
int launch_fft_2pass(int precision, const fft_2pass_args &a, void *stream)
{
    if (a.count == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (precision == SDSP_HIP_F64) {
        switch (a.n) {
        case 1u << 15: return launch_dir<15, 7, double2>(a, s);  //  128 x  256
        case 1u << 16: return launch_dir<16, 8, double2>(a, s);  //  256 x  256
        case 1u << 17: return launch_dir<17, 8, double2>(a, s);  //  256 x  512
        case 1u << 18: return launch_dir<18, 9, double2>(a, s);  //  512 x  512
        case 1u << 19: return launch_dir<19, 9, double2>(a, s);  //  512 x 1024
        case 1u << 20: return launch_dir<20, 10, double2>(a, s); // 1024 x 1024
        default: break;
        }
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the two-pass kernels");
    }
    switch (a.n) {
    // N1 x N2: the longer factor goes to pass 2, whose exchange writes are conflict-free at N2 = 1024 only
    case 1u << 16: return launch_dir<16, 8, float2>(a, s);  //  256 x  256
    case 1u << 17: return launch_dir<17, 8, float2>(a, s);  //  256 x  512
    case 1u << 18: return launch_dir<18, 9, float2>(a, s);  //  512 x  512
    case 1u << 19: return launch_dir<19, 9, float2>(a, s);  //  512 x 1024
    case 1u << 21: return a.reverse ? launch_pair64<21, true>(a, s) : launch_pair64<21, false>(a, s); // 1024 x 2048
    case 1u << 22: return a.reverse ? launch_pair64<22, true>(a, s) : launch_pair64<22, false>(a, s); // 2048 x 2048
    default: break;
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the two-pass kernels");
}
size_t fft_2pass_sync_bytes(uint64_t units, uint32_t queues) { return handoff::sync_words((uint32_t)units, queues) * sizeof(unsigned); }

// the default schedule of a size: unit = transforms per ticket step (8 MiB of data, or one transform where it is larger), and
// queues x ring units of intermediate = 256 MiB (the Infinity Cache), pass 2 trailing pass 1 by ring - 2 steps
void fft_2pass_fused_shape(uint32_t n, int precision, uint32_t *unit, uint32_t *queues, uint32_t *ring, uint32_t *lag)
{
    const uint64_t bytes = (uint64_t)n * (precision == SDSP_HIP_F64 ? 16 : 8);
    *unit = *queues = *ring = *lag = 0;
    if (!fft_2pass_supports(n, precision))
        return;
    *unit = (uint32_t)std::max<uint64_t>(1, (8ull << 20) / bytes);
    const uint64_t unit_bytes = *unit * bytes;
    *ring = 4;
    *lag = 2;
    *queues = (uint32_t)std::max<uint64_t>(2, (256ull << 20) / (unit_bytes * *ring));
}

// Persistent launch against two launches per chunk, same call, % of HBM peak on the compulsory bytes at 1 GiB / 2 GiB batches
// (tools/lab_fft2p_fused.py, profiles/r03_fft2p_fused_lab.txt; p = persistent, c = chunked):
//   f32  2^16: p 30.8 c 37.5 (2 GiB)      2^17: 35.7 v 35.6 / 40.0 v 36.9     2^18: 36.0 v 37.0 / 39.6 v 38.0     2^19: 37.2 v 36.0 / 40.7 v 37.7
//        2^21: 32.7 v 32.3 / 35.0 v 33.1   2^22: 32.0 v 29.2 / 33.8 v 30.3
//   f64  2^16: 35.4 v 36.2 / 37.9 v 37.1   2^17: 35.7 v 36.2 / 38.0 v 37.0     2^18: 35.5 v 33.8 / 37.9 v 34.6     2^19: 36.0 v 32.6 / 38.0 v 33.7
//        2^20: 36.4 v 34.1 / 38.2 v 34.3
// The persistent launch pays per item (two barriers, a drained store queue before the hand-off), which small tiles feel
// (2^16: 32 KiB items), and a ramp at either end of the batch, which short batches feel.  Sustained (bench.py --workload fft, 100 steps of
// 1 GiB, tools/ab_two_pass_bench.sh): f32 2^16 30.9 v 38.0, 2^17 40.3 v 37.5, 2^18 40.1 v 37.9; f64 2^15 36.8 v 36.4, 2^16 36.5 v 37.6,
// 2^17 37.5 v 37.3.
// Since the small-tile sizes run two tiles side by side in both passes (MIN_THREADS = 256: f32 2^16 40.0 v 38.4, f64 2^15 38.5 v 36.8,
// f64 2^16 38.3 v 37.7 sustained) the persistent launch is at least level at every size: it is every two-pass plan's default
// (capi.hip: select_kernel; the two launches per chunk are variant 3).

// both passes over `count` transforms in ONE persistent launch (the workspace holds queues x ring x unit intermediates)
int launch_fft_2pass_fused(int precision, const fft_2pass_fused_args &a, void *stream)
{
    if (a.count == 0)
        return SDSP_HIP_OK;
    if (a.ring == 0 || a.lag >= a.ring || a.queues == 0 || a.unit == 0 || a.count > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "fft_2pass fused: need lag < ring, queues > 0, unit > 0 and a sane count");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (precision == SDSP_HIP_F64) {
        switch (a.n) {
        case 1u << 15: return launch_fused_dir<shape32<15, 7, double2, 256>>(a, s);
        case 1u << 16: return launch_fused_dir<shape32<16, 8, double2, 256>>(a, s);
        case 1u << 17: return launch_fused_dir<shape32<17, 8, double2>>(a, s);
        case 1u << 18: return launch_fused_dir<shape32<18, 9, double2>>(a, s);
        case 1u << 19: return launch_fused_dir<shape32<19, 9, double2>>(a, s);
        case 1u << 20: return launch_fused_dir<shape32<20, 10, double2>>(a, s);
        default: break;
        }
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the two-pass kernels");
    }
    switch (a.n) {
    case 1u << 16: return launch_fused_dir<shape32<16, 8, float2, 256>>(a, s);
    case 1u << 17: return launch_fused_dir<shape32<17, 8, float2>>(a, s);
    case 1u << 18: return launch_fused_dir<shape32<18, 9, float2>>(a, s);
    case 1u << 19: return launch_fused_dir<shape32<19, 9, float2>>(a, s);
    case 1u << 20: return launch_fused_dir<shape32<20, 10, float2>>(a, s); // (variant 2 of the N = 2^20 plans; their default: fft1m.hip)
    case 1u << 21: return launch_fused_dir<shape64<21>>(a, s);
    case 1u << 22: return launch_fused_dir<shape64<22>>(a, s);
    default: break;
    }
    return fail(SDSP_HIP_ERR_UNSUPPORTED, "size not covered by the two-pass kernels");
}
} // namespace sdsp_hip

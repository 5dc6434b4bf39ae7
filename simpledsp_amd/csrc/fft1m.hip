// fft1m.hip -- batched N = 2^20 radix-2 complex f32 FFT for gfx950 (BASELINE config 3).
//
// sdsp::fft_radix2<T, 2^20> (fft.h:258-299) cannot even be compiled in the reference (its table
// would be 320 MiB of constexpr data); here the 20 radix-2 butterfly stages run as a four-step
// decomposition N = 1024 x 1024 with the transform viewed as a row-major [n1][n2] matrix:
//
//   pass 1 (sdsp_fft1m_cols)  for 16 adjacent columns n2: ten radix-2 stages over n1 (stride 1024),
//                             times the inter-pass twiddle W_N^(n2*k1), written to the workspace
//   pass 2 (sdsp_fft1m_rows)  for 16 adjacent rows k1: ten radix-2 stages over n2 (contiguous),
//                             written transposed, X[k1 + 1024*k2], back into the caller's buffer
//
// Both passes use the same building block: 512 threads = 16 sequences x 32 threads, 32 points per
// thread in registers, two register passes of five radix-2 DIF stages each, ONE exchange through
// LDS.  The exchange moves the real and the imaginary plane separately, so a 1024 x 16 tile costs
// 64 KiB instead of 128 KiB and two workgroups fit a CU (one loads while the other computes).
// Twiddles: stage s of the first register pass needs W_1024^(2^s * u) for the thread's fixed u
// (five values, fetched once) times compile-time W_32 constants; the second register pass needs
// constants only.  Every global access is a 128-byte (pass 1, pass 2 stores) or 256-byte (pass 2
// loads) contiguous segment.  LDS planes are XOR-swizzled so all ds_read/ds_write_b32 are
// bank-conflict free.
//
// The host launches the two passes for a CHUNK of transforms at a time (see capi.hip) so that the
// chunk's intermediate matrix is still resident in the 256 MiB Infinity Cache when pass 2 reads it:
// HBM then sees little more than the compulsory 16 MiB per transform although the algorithm makes
// two passes.  Streaming input / final output use non-temporal accesses to stay out of that cache;
// the intermediate uses the default policy to stay in it.
#include <hip/hip_runtime.h>

#include "fft32.h"
#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
using namespace fft32;

constexpr int kTile = 16;     // sequences per workgroup
constexpr int kThreads = 512; // 16 sequences x 32 threads

// ---- pass 1: 16 columns of one transform ------------------------------------------------------
template <bool REV>
__global__ __launch_bounds__(kThreads, 4) void sdsp_fft1m_cols(const float2 *__restrict__ in,
                                                               float2 *__restrict__ ws,
                                                               const float2 *__restrict__ tw_n,    // W_N^j (j < 1024 used)
                                                               const float2 *__restrict__ tw_1024, // W_1024^j
                                                               uint32_t tiles_per_transform)
{
    // dynamic LDS (76 KiB > the 64 KiB static limit): one real plane [row][col] with rows pair-swapped
    // by row bit 5, then W_1024^j staged in LDS, then the column part of the inter-pass twiddle
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft1m_smem[];
    float *plane = reinterpret_cast<float *>(sdsp_fft1m_smem);
    float2 *w1k = reinterpret_cast<float2 *>(sdsp_fft1m_smem + 1024 * kTile * sizeof(float));
    float2 *qtab = w1k + 1024; // [j][column]: W_N^(32 * n2 * j), the part of the inter-pass twiddle a column shares
    const uint32_t t = threadIdx.x;
    const uint32_t c = t & 15, u = t >> 4;
    reinterpret_cast<float4 *>(w1k)[t] = reinterpret_cast<const float4 *>(tw_1024)[t];
    const uint32_t tile = blockIdx.x % tiles_per_transform;
    const uint64_t xform = blockIdx.x / tiles_per_transform;
    const uint32_t n2 = tile * kTile + c;
    // addresses = wave-uniform base (SGPRs; the per-k part is a compile-time constant) + ONE 32-bit
    // per-thread offset, so the 32 loads / stores share a single offset register
    const float2 *src_tile = in + xform * (1ull << 20) + tile * kTile;
    float2 *dst_tile = ws + xform * (1ull << 20) + tile * kTile;
    const uint32_t toff = (u * 1024 + c) * 8u; // bytes

    float2 x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = nt_load(at(src_tile + 32768 * k, toff));

    __syncthreads();
    // W_N^m = W_1024^(m >> 10) * W_N^(m & 1023).  The coarse factor comes from the LDS table; the fine factor
    // has an angle below 2*pi/1024 = 0.0062 rad, where cos = 1 - t^2/2 and sin = t - t^3/6 are exact to fp32
    // rounding (next terms < 6e-11): no second gather.
    auto twiddle = [&](uint32_t m) {
        const float th = (float)(m & 1023) * 5.9921124526782858e-06f; // 2*pi / 2^20
        const float th2 = th * th;
        const float sn = th - th * th2 * 0.16666667f;
        const float2 fine = float2{ 1.0f - 0.5f * th2, REV ? sn : -sn };
        return cmul(w1k[m >> 10], fine);
    };
    // The inter-pass twiddle of output k1 = 32 j + bu of column n2 is W_N^(n2 bu) * W_N^(32 n2 j): the first
    // factor is one value per thread, the second is shared by the 32 threads of a column -- 16 x 32 values per
    // workgroup, one per thread, parked in LDS (read back after the exchange barriers below).  That replaces a
    // polynomial and a conflict-prone table gather per ELEMENT by one conflict-free LDS read and one multiply.
    qtab[u * 16 + c] = twiddle(32u * n2 * u); // thread (c, u) computes j = u
    fft32_dif<REV, true>(x, w1k, u); // stages with row strides 512 .. 32; twiddles W_1024^(2^s u)

    // exchange rows {u + 32k} -> {32u + k}.  slot(row, col) = (row*16 + col) ^ (((row >> 5) & 1) << 4).
    // Written with two base registers + compile-time offsets (per-element XOR'd addresses would
    // cost 64 VGPRs): writes flip bit 4 for odd k; reads use slot 512u + c + 16*(k ^ (u&1)).
    {
        float *const w_even = plane + (u * 16 + c);
        float *const w_odd = plane + ((u * 16 + c) ^ 16);
        const int flip = (int)(u & 1) * 16;
        const float *const r_even = plane + (512 * u + c) + flip;
        const float *const r_odd = plane + (512 * u + c) - flip;
#pragma unroll
        for (int k = 0; k < 32; k++)
            ((k & 1) ? w_odd : w_even)[512 * k] = x[k].x;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k].x = ((k & 1) ? r_odd : r_even)[16 * k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k++)
            ((k & 1) ? w_odd : w_even)[512 * k] = x[k].y;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k].y = ((k & 1) ? r_odd : r_even)[16 * k];
    }
    fft32_dif<REV, false>(x, w1k, u); // row strides 16 .. 1

    // position 32u + k now holds Y[k1], k1 = bit_reverse10(32u + k); times W_N^(n2*k1), stored at
    // row k1 of the intermediate matrix (default cache policy: it should stay in the Infinity Cache)
    const uint32_t bu = brev5(u);
    const uint32_t soff = (bu * 1024 + c) * 8u; // bytes
    const float2 pw = twiddle(n2 * bu); // W_N^(n2 bu)
    const float2 *const qcol = qtab + c;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if ((k & 7) == 0) // keep at most 8 elements' table reads in flight (register budget)
            __builtin_amdgcn_sched_barrier(0);
        const float2 tw = cmul(pw, qcol[16 * (int)(__brev((uint32_t)k) >> 27)]); // k1 = 32 * bit_reverse5(k) + bu
        // default cache policy on purpose: a streaming (nt) store here measured 13 % slower overall,
        // the intermediate is re-read from the Infinity Cache by pass 2
        *at(dst_tile + 32768 * (int)(__brev((uint32_t)k) >> 27), soff) = cmul(x[k], tw);
    }
}

// ---- pass 2: 16 rows of one transform, written transposed ------------------------------------------
template <bool REV>
__global__ __launch_bounds__(kThreads, 4) void sdsp_fft1m_rows(const float2 *__restrict__ ws,
                                                               float2 *__restrict__ out,
                                                               const float2 *__restrict__ tw_1024,
                                                               uint32_t tiles_per_transform, float scale)
{
    __shared__ float plane[kTile * 1024]; // [row][pos ^ (row | (((pos >> 5) & 1) << 4))]
    const uint32_t t = threadIdx.x;
    const uint32_t tile = blockIdx.x % tiles_per_transform;
    const uint64_t xform = blockIdx.x / tiles_per_transform;

    // first register pass: 32 lanes run along a row (256 contiguous bytes per half wave)
    const uint32_t ra = t >> 5, ua = t & 31;
    const float2 *src_tile = ws + xform * (1ull << 20) + (uint64_t)tile * kTile * 1024;
    const uint32_t aoff = (ra * 1024 + ua) * 8u; // bytes
    float2 x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = *at(src_tile + 32 * k, aoff);
    fft32_dif<REV, true>(x, tw_1024, ua);

    // exchange, and switch the thread mapping so that 16 lanes run across the 16 rows
    const uint32_t rb = t & 15, ub = t >> 4;
    // write slot ra*1024 + ((ua + 32k) ^ (ra | ((k&1) << 4))): the XOR touches the low 5 bits only
    //   -> bases (ua ^ ra) and (ua ^ ra ^ 16) + 32k;
    // read slot rb*1024 + ((32ub + k) ^ (rb | ((ub&1) << 4))) = rb*1024 + 32ub + (k ^ rb ^ 16(ub&1)):
    //   the register index is XOR'ed with a run-time value, so those 32 addresses are rebuilt from an
    //   opaque value with one v_xor each instead of living in registers across the butterflies.
    {
        float *const w_even = plane + ra * 1024 + (ua ^ ra);
        float *const w_odd = plane + ra * 1024 + (ua ^ ra ^ 16);
        const float *const r_base = plane + rb * 1024 + 32 * ub;
        const uint32_t rx = rb | ((ub & 1) << 4);
#pragma unroll
        for (int k = 0; k < 32; k++)
            ((k & 1) ? w_odd : w_even)[32 * k] = x[k].x;
        __syncthreads();
        {
            uint32_t q = rx;
            asm volatile("" : "+v"(q));
#pragma unroll
            for (int k = 0; k < 32; k++)
                x[k].x = r_base[k ^ q];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k++)
            ((k & 1) ? w_odd : w_even)[32 * k] = x[k].y;
        __syncthreads();
        {
            uint32_t q = rx;
            asm volatile("" : "+v"(q));
#pragma unroll
            for (int k = 0; k < 32; k++)
                x[k].y = r_base[k ^ q];
        }
    }
    fft32_dif<REV, false>(x, tw_1024, ua);

    // position 32ub + k of row k1 holds X[k1 + 1024*k2], k2 = bit_reverse10(32ub + k): 16 lanes write
    // 128 contiguous bytes.  Streaming (non-temporal) store of the final result.
    float2 *dst_tile = out + xform * (1ull << 20) + tile * kTile;
    const uint32_t bu = brev5(ub);
    const uint32_t boff = (bu * 1024 + rb) * 8u; // bytes
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if ((k & 7) == 0)
            __builtin_amdgcn_sched_barrier(0);
        float2 v = x[k]; // k2 = bit_reverse5(k)*32 + bit_reverse5(ub)
        if constexpr (REV) { // reverse_fft::ScaleValues, fft.h:128-132
            v.x *= scale;
            v.y *= scale;
        }
        nt_store(at(dst_tile + 32768 * (int)(__brev((uint32_t)k) >> 27), boff), v);
    }
}
} // namespace

// One pass over one chunk of `count` transforms: which = 1 columns (data -> workspace),
// which = 2 rows (workspace -> data).  The workspace holds `count` matrices.
int launch_fft1m_pass(const fft1m_args &a, int which, void *stream)
{
    if (a.count == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const uint32_t tiles = 1024 / kTile;
    const uint64_t blocks = a.count * tiles;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "chunk too large for one launch");
    const float2 *in = reinterpret_cast<const float2 *>(a.data);
    float2 *out = reinterpret_cast<float2 *>(a.data);
    float2 *ws = reinterpret_cast<float2 *>(a.workspace);
    const float2 *twn = reinterpret_cast<const float2 *>(a.tw_n);
    const float2 *tw1k = reinterpret_cast<const float2 *>(a.tw_1024);
    constexpr size_t kColsLds = 1024 * kTile * sizeof(float) + 1024 * sizeof(float2) + 32 * kTile * sizeof(float2);
    static std::atomic<uint64_t> attr_done_rev{ 0 }, attr_done_fwd{ 0 };
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sdsp_fft1m_cols<true>), kColsLds, attr_done_rev))
        return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sdsp_fft1m_cols<false>), kColsLds, attr_done_fwd))
        return rc;
    const dim3 grid((uint32_t)blocks), block(kThreads);
    if (which == 1) {
        if (a.reverse)
            hipLaunchKernelGGL(sdsp_fft1m_cols<true>, grid, block, kColsLds, s, in, ws, twn, tw1k, tiles);
        else
            hipLaunchKernelGGL(sdsp_fft1m_cols<false>, grid, block, kColsLds, s, in, ws, twn, tw1k, tiles);
    } else {
        if (a.reverse)
            hipLaunchKernelGGL(sdsp_fft1m_rows<true>, grid, block, 0, s, ws, out, tw1k, tiles, a.scale);
        else
            hipLaunchKernelGGL(sdsp_fft1m_rows<false>, grid, block, 0, s, ws, out, tw1k, tiles, a.scale);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft1m launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

int launch_fft1m_r2_f32(const fft1m_args &a, void *stream)
{
    if (int rc = launch_fft1m_pass(a, 1, stream))
        return rc;
    return launch_fft1m_pass(a, 2, stream);
}
} // namespace sdsp_hip

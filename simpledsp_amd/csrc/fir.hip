// fir.hip -- batched direct-form FIR filter bank for MI355X (gfx950).  SURVEY 8(f)-4: the reference
// lists "FIR filter" as a TODO (README.md:16) and has no code for it; the recurrence implemented here
// is the textbook y[n] = sum_{k=0}^{T-1} h[k] x[n-k], accumulated in ascending k (acc = h[0] x[n], then
// one multiply and one add per tap -- unfused in f64 so the result is bit-identical to the CPU oracle,
// fused multiply-add in f32).  Layout, ownership and streaming semantics follow the IIR bank
// (casc_2o_iir.h:36-80 as batched in iir.hip): channel-major rows filtered in place, per-channel history
// read at entry and written at exit.
//
// HBM-bound: 8 B (f32) / 16 B (f64) per sample.  One group of `tpr` threads owns one channel row and
// walks it block by block (16*tpr samples).  A block is loaded with 16-byte accesses, a wave covering
// 1 KiB of the row per instruction, into a padded LDS line that also keeps the previous T-1 inputs in
// front of it; each thread then produces 16 CONSECUTIVE outputs from a sliding register window
// (one 16-sample LDS block read per 16 taps), the outputs go back through the same LDS line and leave
// with the same coalesced shape.  Because a row is owned by one group, in-place operation needs no
// second buffer.
#include "sdsp_hip_internal.h"

#include <hip/hip_runtime.h>

namespace sdsp_hip
{
namespace
{
constexpr int kThreads = 256; // the largest workgroup (long filters, f32 up to 16 taps); TH = 128 otherwise (launch_fir)
constexpr int kOut = 16; // outputs per thread = LDS block length

template <typename R> struct vec_of;
template <> struct vec_of<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int lanes = 4;
    static constexpr int pad = 4; // elements of padding after every 16-element block (80-B pitch)
};
template <> struct vec_of<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    static constexpr int lanes = 2;
    static constexpr int pad = 2; // 144-B pitch
};

template <typename R> __device__ __forceinline__ R mul_add(R h, R x, R acc);
template <> __device__ __forceinline__ float mul_add<float>(float h, float x, float acc) { return __builtin_fmaf(h, x, acc); }
template <> __device__ __forceinline__ double mul_add<double>(double h, double x, double acc)
{
    return acc + h * x; // built with -ffp-contract=off: separate multiply and add, as the oracle does
}

// padded LDS position of element p of a line (p counts from the start of the history region)
template <typename R> __device__ __forceinline__ uint32_t slot(uint32_t p) { return p + (p >> 4) * vec_of<R>::pad; }

template <typename R> __device__ __forceinline__ void read_block(const R *line, uint32_t blk, R (&dst)[kOut])
{
    using V = typename vec_of<R>::type;
    constexpr int L = vec_of<R>::lanes;
    const V *src = reinterpret_cast<const V *>(line + blk * (kOut + vec_of<R>::pad));
#pragma unroll
    for (int i = 0; i < kOut / L; i++) {
        V v = src[i];
#pragma unroll
        for (int j = 0; j < L; j++)
            dst[i * L + j] = v[j];
    }
}

// 16 taps h[k0 .. k0+16) applied to 16 outputs.  win[0..16) = the block 16 samples earlier,
// win[16..32) = the block the oldest of these taps starts in.  `count` < 16 only in the last chunk.
template <typename R, bool FIRST, bool PARTIAL>
__device__ __forceinline__ void taps16(const R *__restrict__ h, uint32_t k0, uint32_t count, const R (&win)[2 * kOut], R (&acc)[kOut])
{
#pragma unroll
    for (int kk = 0; kk < kOut; kk++) {
        if (!PARTIAL || static_cast<uint32_t>(kk) < count) { // wave-uniform
            const R hk = h[k0 + kk];
#pragma unroll
            for (int r = 0; r < kOut; r++) {
                if (FIRST && kk == 0)
                    acc[r] = hk * win[kOut + r];
                else
                    acc[r] = mul_add<R>(hk, win[kOut + r - kk], acc[r]);
            }
        }
    }
}

// One thread's 16 outputs for the block `mine` of its line.  Tap chunk kc (16 taps) needs the LDS blocks
// mine - kc and mine - kc - 1; the window slides through registers.
template <typename R, bool PACKED> struct fir_block {
    // generic form (used for f64): one output per accumulator, multiply then add per tap
    static __device__ __forceinline__ void run(const R *line, uint32_t mine, const R *__restrict__ h, uint32_t taps, R (&acc)[kOut])
    {
        const uint32_t nfull = taps / kOut, rem = taps % kOut;
        R win[2 * kOut];
        {
            R cur[kOut];
            read_block<R>(line, mine, cur);
#pragma unroll
            for (int i = 0; i < kOut; i++)
                win[kOut + i] = cur[i];
        }
        uint32_t kc = 0;
        if (nfull) {
            R prev[kOut];
            read_block<R>(line, mine - 1, prev);
#pragma unroll
            for (int i = 0; i < kOut; i++)
                win[i] = prev[i];
            taps16<R, true, false>(h, 0, kOut, win, acc);
            for (kc = 1; kc < nfull; kc++) {
#pragma unroll
                for (int i = 0; i < kOut; i++)
                    win[kOut + i] = win[i];
                read_block<R>(line, mine - kc - 1, prev);
#pragma unroll
                for (int i = 0; i < kOut; i++)
                    win[i] = prev[i];
                taps16<R, false, false>(h, kc * kOut, kOut, win, acc);
            }
            if (rem) {
#pragma unroll
                for (int i = 0; i < kOut; i++)
                    win[kOut + i] = win[i];
            }
        }
        if (rem) {
            R prev[kOut];
            read_block<R>(line, mine - kc - 1, prev);
#pragma unroll
            for (int i = 0; i < kOut; i++)
                win[i] = prev[i];
            if (nfull)
                taps16<R, false, true>(h, kc * kOut, rem, win, acc);
            else
                taps16<R, true, true>(h, 0, rem, win, acc);
        }
    }
};

// f32: two outputs per v_pk_fma_f32.  Outputs (2p, 2p+1) share an accumulator pair; at step k the low half
// takes tap k and the high half tap k+1 -- both multiply the SAME sample x[n_2p - k], so the instruction is
// (h[k], h[k+1]) * broadcast(x) + acc.  The high half takes tap 0 on its own first and the low half the last
// tap on its own at the end, so every output still accumulates its taps in ascending order (same values
// as the one-FMA-per-tap form).  Non-packed f32 FMA tops out at 78 TFLOP/s on this chip; 64 taps need 90.
template <> struct fir_block<float, true> {
    typedef float f2 __attribute__((ext_vector_type(2)));
    template <bool TAIL>
    static __device__ __forceinline__ void chunk(const float *__restrict__ h, uint32_t k0, uint32_t count, const float (&win)[2 * kOut],
                                                 f2 (&acc)[kOut / 2])
    {
#pragma unroll
        for (int kk = 0; kk < kOut; kk++) {
            if (!TAIL || static_cast<uint32_t>(kk) + 1 < count) { // wave-uniform
                const f2 hv = { h[k0 + kk], h[k0 + kk + 1] };
#pragma unroll
                for (int p = 0; p < kOut / 2; p++) {
                    const float x = win[kOut + 2 * p - kk];
                    acc[p] = __builtin_elementwise_fma(hv, (f2){ x, x }, acc[p]);
                }
            } else if (static_cast<uint32_t>(kk) + 1 == count) { // the filter's last tap: low halves only
                const float hk = h[k0 + kk];
#pragma unroll
                for (int p = 0; p < kOut / 2; p++)
                    acc[p].x = __builtin_fmaf(hk, win[kOut + 2 * p - kk], acc[p].x);
            }
        }
    }
    static __device__ __forceinline__ void run(const float *line, uint32_t mine, const float *__restrict__ h, uint32_t taps,
                                               float (&out)[kOut])
    {
        const uint32_t nch = (taps + kOut - 1) / kOut;
        float win[2 * kOut];
        {
            float cur[kOut];
            read_block<float>(line, mine, cur);
#pragma unroll
            for (int i = 0; i < kOut; i++)
                win[kOut + i] = cur[i];
        }
        f2 acc[kOut / 2];
        {
            const float h0 = h[0];
#pragma unroll
            for (int p = 0; p < kOut / 2; p++)
                acc[p] = (f2){ 0.f, h0 * win[kOut + 2 * p + 1] };
        }
        float prev[kOut];
        uint32_t kc = 0;
        for (; kc + 1 < nch; kc++) {
            read_block<float>(line, mine - kc - 1, prev);
#pragma unroll
            for (int i = 0; i < kOut; i++)
                win[i] = prev[i];
            chunk<false>(h, kc * kOut, kOut, win, acc);
#pragma unroll
            for (int i = 0; i < kOut; i++)
                win[kOut + i] = win[i];
        }
        read_block<float>(line, mine - kc - 1, prev);
#pragma unroll
        for (int i = 0; i < kOut; i++)
            win[i] = prev[i];
        chunk<true>(h, kc * kOut, taps - kc * kOut, win, acc);
#pragma unroll
        for (int p = 0; p < kOut / 2; p++) {
            out[2 * p] = acc[p].x;
            out[2 * p + 1] = acc[p].y;
        }
    }
};

struct fir_kargs {
    void *data;
    void *state;
    uint64_t channels, samples, stride;
    uint32_t taps;
    uint32_t tpr_log2; // threads per row
    uint32_t hist;     // history region length in elements: 16 * ceil(taps / 16)
    uint32_t vec_ok;   // rows 16-byte aligned
};

// h is its own __restrict__ parameter so that the coefficient reads become scalar (s_load) instructions
template <typename R, bool NT, bool PACKED, int OCC, int TH = kThreads>
__global__ __launch_bounds__(TH, OCC) void sdsp_fir_kernel(fir_kargs a, const R *__restrict__ h)
{
    using V = typename vec_of<R>::type;
    constexpr int L = vec_of<R>::lanes;
    constexpr int VPT = kOut / L; // vectors per thread per block
    extern __shared__ __align__(16) unsigned char lds_raw[];

    const uint32_t tpr = 1u << a.tpr_log2;
    const uint32_t row = threadIdx.x >> a.tpr_log2;
    const uint32_t t = threadIdx.x & (tpr - 1);
    const uint32_t rows_per_wg = (uint32_t)TH >> a.tpr_log2;
    const uint64_t ch = static_cast<uint64_t>(blockIdx.x) * rows_per_wg + row;
    const bool live = ch < a.channels;
    const uint32_t block_len = tpr * kOut;
    const uint32_t line_elems = slot<R>(a.hist + block_len);
    R *line = reinterpret_cast<R *>(lds_raw) + static_cast<size_t>(row) * line_elems;
    R *rowp = static_cast<R *>(a.data) + (live ? ch : 0) * a.stride;
    const uint32_t T1 = a.taps - 1;
    R *statep = a.state ? static_cast<R *>(a.state) + (live ? ch : 0) * T1 : nullptr;

    // history: element e = -1-j of the stream is state[j]; it sits at line position hist-1-j
    for (uint32_t j = t; j < a.hist; j += tpr) {
        R v = R(0);
        if (live && statep && j < T1)
            v = statep[j];
        line[slot<R>(a.hist - 1 - j)] = v;
    }

    for (uint64_t s0 = 0; s0 < a.samples; s0 += block_len) {
        const uint64_t left = a.samples - s0;
        const uint32_t len = left < block_len ? static_cast<uint32_t>(left) : block_len;
        const bool whole = len == block_len && a.vec_ok;
        R *blk = rowp + s0;

        // ---- load the block: vector i of thread t is elements (i*tpr + t)*L ...
        if (live) {
            if (whole) {
                V v[VPT];
#pragma unroll
                for (int i = 0; i < VPT; i++) {
                    const V *src = reinterpret_cast<const V *>(blk) + (i * tpr + t);
                    v[i] = NT ? __builtin_nontemporal_load(src) : *src;
                }
#pragma unroll
                for (int i = 0; i < VPT; i++)
                    *reinterpret_cast<V *>(line + slot<R>(a.hist + (i * tpr + t) * L)) = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < VPT; i++)
#pragma unroll
                    for (int j = 0; j < L; j++) {
                        const uint32_t e = (i * tpr + t) * L + j;
                        line[slot<R>(a.hist + e)] = e < len ? blk[e] : R(0);
                    }
            }
        }
        __syncthreads();

        // ---- 16 consecutive outputs per thread
        R acc[kOut];
        fir_block<R, PACKED>::run(line, (a.hist >> 4) + t, h, a.taps, acc);
        __syncthreads();

        // ---- carry the newest inputs: to the state buffer after the last block, else to the history region
        const bool last = s0 + block_len >= a.samples;
        if (last) {
            if (live && statep)
                for (uint32_t j = t; j < T1; j += tpr)
                    statep[j] = line[slot<R>(a.hist + len - 1 - j)]; // len-1-j >= -hist always
        } else {
            // block_len >= hist (host guarantees), so source and destination do not overlap
            for (uint32_t j = t; j < a.hist; j += tpr)
                line[slot<R>(j)] = line[slot<R>(block_len + j)];
        }
        __syncthreads();

        // ---- outputs back through the line (block region), then out with the load's shape
        {
            V *dst = reinterpret_cast<V *>(line + slot<R>(a.hist + t * kOut));
#pragma unroll
            for (int i = 0; i < VPT; i++) {
                V v;
#pragma unroll
                for (int j = 0; j < L; j++)
                    v[j] = acc[i * L + j];
                dst[i] = v;
            }
        }
        __syncthreads();
        if (live) {
            if (whole) {
#pragma unroll
                for (int i = 0; i < VPT; i++) {
                    const V v = *reinterpret_cast<const V *>(line + slot<R>(a.hist + (i * tpr + t) * L));
                    V *dstg = reinterpret_cast<V *>(blk) + (i * tpr + t);
                    if (NT)
                        __builtin_nontemporal_store(v, dstg);
                    else
                        *dstg = v;
                }
            } else {
#pragma unroll
                for (int i = 0; i < VPT; i++)
#pragma unroll
                    for (int j = 0; j < L; j++) {
                        const uint32_t e = (i * tpr + t) * L + j;
                        if (e < len)
                            blk[e] = line[slot<R>(a.hist + e)];
                    }
            }
        }
        // the next block's LDS writes touch exactly the positions this thread just read; the history
        // region was written before the barrier above
    }
}

uint32_t ceil_log2(uint64_t v)
{
    uint32_t l = 0;
    while ((1ull << l) < v)
        l++;
    return l;
}
} // namespace

size_t fir_lds_bytes(int precision, uint32_t taps, uint32_t tpr_log2, uint32_t threads)
{
    const uint32_t hist = kOut * ((taps + kOut - 1) / kOut);
    const uint32_t elems = hist + (kOut << tpr_log2);
    const uint32_t pad = precision == SDSP_HIP_F64 ? vec_of<double>::pad : vec_of<float>::pad;
    const size_t line = elems + (elems >> 4) * pad;
    return line * (threads >> tpr_log2) * (precision == SDSP_HIP_F64 ? 8 : 4);
}

int launch_fir(int precision, const fir_args &fa, int variant, void *stream_v)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const size_t rs = precision == SDSP_HIP_F64 ? 8 : 4;
    fir_kargs k{};
    k.data = fa.data;
    k.state = fa.state;
    k.channels = fa.channels;
    k.samples = fa.samples;
    k.stride = fa.stride;
    k.taps = fa.taps;
    k.hist = kOut * ((fa.taps + kOut - 1) / kOut);
    // threads per row: enough for the row (up to 256), never fewer than the history needs
    uint32_t l = ceil_log2((fa.samples + kOut - 1) / kOut);
    const uint32_t lmin = ceil_log2(k.hist / kOut);
    if (l < lmin)
        l = lmin;
    if (lmin > 8)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "too many taps");
    // Workgroups of 128 threads (rows in blocks of at most 2048 samples, half the LDS, twice the workgroups per CU) where they measured
    // faster -- the register-pass FFT families' finding (fft_reg.hip), one call, 1M channels x 4096 samples: f32 32 / 64 taps 70.4 / 53.0 %
    // of HBM peak against 63.3 / 49.9 % with 256 threads, f64 16 / 32 taps 72.3 / 60.5 against 69 / 52 %; f32 with up to 16 taps keeps 256
    // (74.4 against 71.3 %), and so do filters whose history needs more than 128 threads per row (> 2048 taps)
    const bool small = lmin <= 7 && (precision == SDSP_HIP_F64 || fa.taps > 16);
    const uint32_t lmax = small ? 7 : 8, threads = small ? 128 : 256;
    if (l > lmax)
        l = lmax;
    k.tpr_log2 = l;
    k.vec_ok = (reinterpret_cast<uintptr_t>(fa.data) % 16 == 0 && (fa.stride * rs) % 16 == 0) ? 1 : 0;
    const size_t lds = fir_lds_bytes(precision, fa.taps, l, threads);
    const uint32_t rows_per_wg = threads >> l;
    const uint64_t grid = (fa.channels + rows_per_wg - 1) / rows_per_wg;
    if (grid > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "too many channels for one launch");
    auto run = [&](auto kernel, auto hp) -> int {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(lds));
            if (e != hipSuccess)
                return fail(SDSP_HIP_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(grid)), dim3(threads), lds, stream, k, hp);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return fail(SDSP_HIP_ERR_HIP, std::string("fir launch: ") + hipGetErrorString(e));
        return SDSP_HIP_OK;
    };
    if (precision == SDSP_HIP_F64) {
        const double *hp = static_cast<const double *>(fa.h);
        // f64: multiply + add per tap (bit-exact order); 209 VGPRs, two waves per SIMD
        if (small)
            return variant == 1 ? run(sdsp_fir_kernel<double, false, false, 2, 128>, hp) : run(sdsp_fir_kernel<double, true, false, 2, 128>, hp);
        return variant == 1 ? run(sdsp_fir_kernel<double, false, false, 2>, hp) : run(sdsp_fir_kernel<double, true, false, 2>, hp);
    }
    const float *hp = static_cast<const float *>(fa.h);
    // f32 (measured, 1M channels x 4096 samples, % of 8 TB/s): one FMA per tap at 120 VGPRs / 4 waves per
    // SIMD: 16 taps 76 %, 32 taps 66 %, 64 taps 46 %; packed FMAs at 128 VGPRs: 74 / 65 / 50 %
    if (small) {
        switch (variant) {
        case 1: return run(sdsp_fir_kernel<float, false, false, 4, 128>, hp); // default cache policy
        case 2: return run(sdsp_fir_kernel<float, true, true, 4, 128>, hp);
        case 3: return run(sdsp_fir_kernel<float, true, false, 4, 128>, hp);
        default:
            return fa.taps < 48 ? run(sdsp_fir_kernel<float, true, false, 4, 128>, hp) : run(sdsp_fir_kernel<float, true, true, 4, 128>, hp);
        }
    }
    switch (variant) {
    case 1: return run(sdsp_fir_kernel<float, false, false, 4>, hp); // default cache policy
    case 2: return run(sdsp_fir_kernel<float, true, true, 4>, hp);
    case 3: return run(sdsp_fir_kernel<float, true, false, 4>, hp);
    default:
        return fa.taps < 48 ? run(sdsp_fir_kernel<float, true, false, 4>, hp) : run(sdsp_fir_kernel<float, true, true, 4>, hp);
    }
}
} // namespace sdsp_hip

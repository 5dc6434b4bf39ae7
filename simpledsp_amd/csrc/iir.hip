// iir.hip -- banks of cascaded second-order sections on gfx950.
//
// Batched form of sdsp::casc_2o_iir<m_t>::process (casc_2o_iir.h:36-80) and of the
// numerator-folded casc_2o_iir_{lp,hp,bp}::process_spec (:286-295, :344-353, :402-411): many
// independent channels, shared coefficients, per-channel state.  One lane owns one channel and
// runs the reference's Direct-Form-I recurrence.  The f64 kernels keep the reference's exact
// operation order (this file is compiled with -ffp-contract=off), so they reproduce the reference's
// doubles bit for bit; the f32 kernels (parity by tolerance) fuse each multiply-subtract pair.  In
// both, block-by-block streaming is bit-identical to one long call.  Built with -fno-slp-vectorize:
// packing this scalar recurrence into v_pk_*_f32 cost ~1000 v_mov per kernel for no gain on SIMD-32.
//
// The path is HBM-bound (8 B per f32 sample, 33 flop): the work is in the data movement.  Each
// channel's samples are contiguous in memory, so a lane reading "its" channel would touch one
// cache line per lane.  Instead a wave moves a [64 channels x T samples] tile with 16-byte
// per-lane accesses in which 128..256 consecutive bytes of one channel are covered by
// neighbouring lanes, parks it in a padded (bank-conflict-free) LDS tile, and every lane then
// reads its own row.  Results go back the same way.  The next tile's loads are issued before the
// current tile is computed.  Coefficients live in SGPRs (kernel arguments).
#include <hip/hip_runtime.h>

#include <atomic>
#include <string>
#include <utility>

#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
// S: the type the samples are stored in; R: the type the recurrence (state, coefficients, arithmetic) runs in.
// (float, float), (double, double), or -- the mixed mode, SDSP_HIP_F32_F64STATE -- (float, double): 8 bytes of HBM traffic per
// sample with the double-precision recurrence's accuracy (f32 state loses 1e-4 at f0/fs = 0.005).
template <typename S, typename R, int M> struct iir_dev_args {
    S *data;
    R *state; // nullable; state[(3*j + age) * channels + c]
    uint64_t channels, samples, stride;
    R gain;
    R a1[M], a2[M], b1[M], b2[M];
};

// a kernel's pair of types: S samples in memory, R recurrence
template <typename S_, typename R_> struct prec {
    using S = S_;
    using R = R_;
    static constexpr bool fused = sizeof(S_) == 4; // f32 and mixed: parity by tolerance; f64: bit-exact operation order
};
using prec_f32 = prec<float, float>;
using prec_f64 = prec<double, double>;
using prec_mix = prec<float, double>;

template <typename R> struct vec16;
template <> struct vec16<float> {
    using type = float4;
    using native = float __attribute__((ext_vector_type(4)));
    static constexpr int n = 4;
};
template <> struct vec16<double> {
    using type = double2;
    using native = double __attribute__((ext_vector_type(2)));
    static constexpr int n = 2;
};

// 16-byte global accesses; NT = streaming (non-temporal) policy: every sample is touched exactly
// once each way, and keeping it out of the L2 / Infinity-Cache replacement state is worth ~10 % on
// in-place streams (tools/membench.hip)
template <typename R, bool NT> __device__ __forceinline__ typename vec16<R>::type gload16(const R *p)
{
    using V = typename vec16<R>::type;
    using N = typename vec16<R>::native;
    if constexpr (NT) {
        const N v = __builtin_nontemporal_load(reinterpret_cast<const N *>(p));
        V out;
        __builtin_memcpy(&out, &v, 16);
        return out;
    } else {
        return *reinterpret_cast<const V *>(p);
    }
}
template <typename R, bool NT> __device__ __forceinline__ void gstore16(R *p, typename vec16<R>::type a)
{
    using V = typename vec16<R>::type;
    using N = typename vec16<R>::native;
    if constexpr (NT) {
        N v;
        __builtin_memcpy(&v, &a, 16);
        __builtin_nontemporal_store(v, reinterpret_cast<N *>(p));
    } else {
        *reinterpret_cast<V *>(p) = a;
    }
}

// One sample through the cascade.  y1[j] / y2[j] are level j's values one / two samples ago
// (level 0 = gain-scaled input, level M = output): the reference's m_mem ring (casc_2o_iir.h:15)
// with the ring index resolved at compile time.
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// FUSED: parity by tolerance (f32 and mixed mode) -- each "x*b - y*a" pair is a multiply and an FMA; !FUSED (f64): the
// reference's exact operation order, bit for bit.
template <typename R, int KIND, int M, bool FUSED, typename ARGS>
__device__ __forceinline__ R cascade_step(R x, const ARGS &p, R (&y1)[M + 1], R (&y2)[M + 1], R (&y3)[M + 1])
{
    R cur[M + 1];
    cur[0] = x * p.gain; // :52 / :242
#pragma unroll
    for (int j = 0; j < M; j++) {
        R acc = cur[j];
        if constexpr (FUSED) {
            // parity by tolerance (1e-6): same terms, grouped as the reference groups them, but
            // each "x*b - y*a" pair costs a multiply and an FMA instead of two multiplies and a subtract
            if constexpr (KIND == SDSP_HIP_IIR_GENERIC) {
                acc += fma_t(y1[j], p.b1[j], -(y1[j + 1] * p.a1[j]));
                acc += fma_t(y2[j], p.b2[j], -(y2[j + 1] * p.a2[j]));
            } else if constexpr (KIND == SDSP_HIP_IIR_LP) {
                acc += fma_t(-y1[j + 1], p.a1[j], y1[j] + y1[j]);
                acc += fma_t(-y2[j + 1], p.a2[j], y2[j]);
            } else if constexpr (KIND == SDSP_HIP_IIR_HP) {
                acc += fma_t(-y1[j + 1], p.a1[j], -y1[j] - y1[j]);
                acc += fma_t(-y2[j + 1], p.a2[j], y2[j]);
            } else {
                acc += -y1[j + 1] * p.a1[j];
                acc += fma_t(-y2[j + 1], p.a2[j], -y2[j]);
            }
        } else if constexpr (KIND == SDSP_HIP_IIR_GENERIC) { // :67-68
            acc += y1[j] * p.b1[j] - y1[j + 1] * p.a1[j];
            acc += y2[j] * p.b2[j] - y2[j + 1] * p.a2[j];
        } else if constexpr (KIND == SDSP_HIP_IIR_LP) { // :292-293
            acc += y1[j] + y1[j] - y1[j + 1] * p.a1[j];
            acc += y2[j] - y2[j + 1] * p.a2[j];
        } else if constexpr (KIND == SDSP_HIP_IIR_HP) { // :350-351
            acc += -y1[j] - y1[j] - y1[j + 1] * p.a1[j];
            acc += y2[j] - y2[j + 1] * p.a2[j];
        } else { // band pass :408-409
            acc += -y1[j + 1] * p.a1[j];
            acc += -y2[j] - y2[j + 1] * p.a2[j];
        }
        cur[j + 1] = acc;
    }
#pragma unroll
    for (int j = 0; j <= M; j++) {
        y3[j] = y2[j];
        y2[j] = y1[j];
        y1[j] = cur[j];
    }
    return cur[M]; // :71 / :254
}

template <typename R, int M, typename ARGS>
__device__ __forceinline__ void load_state(const ARGS &p, uint64_t c, R (&y1)[M + 1],
                                           R (&y2)[M + 1], R (&y3)[M + 1])
{
#pragma unroll
    for (int j = 0; j <= M; j++) {
        y1[j] = y2[j] = y3[j] = R(0);
    }
    if (p.state) {
#pragma unroll
        for (int j = 0; j <= M; j++) {
            y1[j] = p.state[(uint64_t)(3 * j + 0) * p.channels + c];
            y2[j] = p.state[(uint64_t)(3 * j + 1) * p.channels + c];
            y3[j] = p.state[(uint64_t)(3 * j + 2) * p.channels + c];
        }
    }
}

template <typename R, int M, typename ARGS>
__device__ __forceinline__ void store_state(const ARGS &p, uint64_t c, const R (&y1)[M + 1],
                                            const R (&y2)[M + 1], const R (&y3)[M + 1])
{
    if (p.state) {
#pragma unroll
        for (int j = 0; j <= M; j++) {
            p.state[(uint64_t)(3 * j + 0) * p.channels + c] = y1[j];
            p.state[(uint64_t)(3 * j + 1) * p.channels + c] = y2[j];
            p.state[(uint64_t)(3 * j + 2) * p.channels + c] = y3[j];
        }
    }
}

// ---- direct variant: lane = channel, scalar global accesses.  Any alignment, any length.
template <typename P, int KIND, int M>
__global__ __launch_bounds__(256) void sdsp_iir_direct_kernel(iir_dev_args<typename P::S, typename P::R, M> p)
{
    using S = typename P::S;
    using R = typename P::R;
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.channels)
        return;
    R y1[M + 1], y2[M + 1], y3[M + 1];
    load_state<R, M>(p, c, y1, y2, y3);
    S *row = p.data + c * p.stride;
    for (uint64_t s = 0; s < p.samples; s++)
        row[s] = (S)cascade_step<R, KIND, M, P::fused>((R)row[s], p, y1, y2, y3);
    store_state<R, M>(p, c, y1, y2, y3);
}

// ---- super-tile variant (default).  One wave per workgroup owns 64 channels.  Per step it moves a
// [64 channels x 512 bytes] super-tile: 32 sixteen-byte accesses per lane, issued ROW-GROUP-MAJOR --
// the four consecutive 128-byte pieces of the same 8 channels in four back-to-back instructions -- so
// that DRAM sees 512 contiguous bytes per channel per burst.  That shape streams at 5.66 TB/s in
// place where one 128-byte piece per step stops at 4.9 (tools/membench.hip, gpurun_out round 1).
// The super-tile stays in registers (128 VGPRs); its four 128-byte sub-tiles go through a padded
// 9 KiB LDS transpose one after the other, and the filtered samples return to the same registers,
// so the stores have the same burst shape as the loads.
template <typename P, int KIND, int M, bool NT, int SUBS>
__global__ __launch_bounds__(64) void sdsp_iir_supertile_kernel(iir_dev_args<typename P::S, typename P::R, M> p)
{
    using S = typename P::S;
    using R = typename P::R;
    using V = typename vec16<S>::type;
    constexpr int EPV = vec16<S>::n;          // elements per 16-byte vector
    constexpr int ROWB = 128;                 // bytes of one channel per sub-tile
    constexpr int T = ROWB / (int)sizeof(S);  // samples per sub-tile
    constexpr int NV = ROWB / 16;             // = 8 vectors per sub-row = lanes per row
    constexpr int RPI = 64 / NV;              // = 8 rows per wave-wide access
    // SUBS = sub-tiles per super-tile: 4 -> 512 B per channel per burst, 128 data VGPRs
    constexpr int PITCH = ROWB + 16;

    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_iir_smem[];
    unsigned char *tile = sdsp_iir_smem;
    const int lane = threadIdx.x;
    const uint64_t ch0 = (uint64_t)blockIdx.x * 64;
    const uint64_t my_ch = ch0 + lane;
    const bool have_ch = my_ch < p.channels;

    R y1[M + 1], y2[M + 1], y3[M + 1];
    load_state<R, M>(p, have_ch ? my_ch : 0, y1, y2, y3);

    const int piece = lane % NV, sub = lane / NV;
    const uint64_t n_super = (p.samples + SUBS * T - 1) / (SUBS * T);
    // interior workgroups (all 64 channels exist) take an unguarded path for their full super-tiles:
    // one per-lane base pointer + wave-uniform offsets, no per-access predicates
    const bool interior = ch0 + 64 <= p.channels;
    S *const lane_base = p.data + (ch0 + sub) * p.stride + (uint64_t)piece * EPV;
    const uint64_t group_step = (uint64_t)RPI * p.stride; // elements between row groups
    for (uint64_t st = 0; st < n_super; st++) {
        V stage[SUBS * NV]; // register SUBS*i + j: rows 8i..8i+7, sub-tile j
        const bool full = interior && (st + 1) * SUBS * T <= p.samples;
        S *const tile_base = lane_base + st * SUBS * T;
        if (full) {
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int j = 0; j < SUBS; j++)
                    stage[SUBS * i + j] = gload16<S, NT>(tile_base + i * group_step + j * T);
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const uint64_t ch = ch0 + (uint64_t)(i * RPI + sub);
#pragma unroll
                for (int j = 0; j < SUBS; j++) {
                    const uint64_t s0 = (st * SUBS + j) * T + (uint64_t)piece * EPV;
                    stage[SUBS * i + j] = V{};
                    if (ch < p.channels && s0 < p.samples)
                        stage[SUBS * i + j] = gload16<S, NT>(p.data + ch * p.stride + s0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SUBS; j++) {
            const uint64_t first = (st * SUBS + j) * T;
            if (first >= p.samples)
                break;
            const uint64_t left = p.samples - first;
            const int valid = left < (uint64_t)T ? (int)left : T;
#pragma unroll
            for (int i = 0; i < NV; i++)
                *reinterpret_cast<V *>(tile + (i * RPI + sub) * PITCH + piece * 16) = stage[SUBS * i + j];
            __syncthreads();
            V *myrow = reinterpret_cast<V *>(tile + lane * PITCH);
            if (valid == T) {
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    V x = myrow[v];
                    S *xe = reinterpret_cast<S *>(&x);
#pragma unroll
                    for (int e = 0; e < EPV; e++)
                        xe[e] = (S)cascade_step<R, KIND, M, P::fused>((R)xe[e], p, y1, y2, y3);
                    myrow[v] = x;
                }
            } else {
                for (int v = 0; v * EPV < valid; v++) {
                    V x = myrow[v];
                    S *xe = reinterpret_cast<S *>(&x);
#pragma unroll
                    for (int e = 0; e < EPV; e++)
                        xe[e] = (S)cascade_step<R, KIND, M, P::fused>((R)xe[e], p, y1, y2, y3);
                    myrow[v] = x;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NV; i++)
                stage[SUBS * i + j] = *reinterpret_cast<const V *>(tile + (i * RPI + sub) * PITCH + piece * 16);
            __syncthreads();
        }
        if (full) {
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int j = 0; j < SUBS; j++)
                    gstore16<S, NT>(tile_base + i * group_step + j * T, stage[SUBS * i + j]);
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const uint64_t ch = ch0 + (uint64_t)(i * RPI + sub);
#pragma unroll
                for (int j = 0; j < SUBS; j++) {
                    const uint64_t s0 = (st * SUBS + j) * T + (uint64_t)piece * EPV;
                    if (ch < p.channels && s0 < p.samples)
                        gstore16<S, NT>(p.data + ch * p.stride + s0, stage[SUBS * i + j]);
                }
            }
        }
    }
    if (have_ch && p.samples)
        store_state<R, M>(p, my_ch, y1, y2, y3);
}

// ---- wide super-tile variant.  The same [64 channels x 512 bytes] super-tile in 128 VGPRs, but every load / store
// instruction covers 512 CONTIGUOUS bytes of two channels (lanes 0-31 one row, lanes 32-63 the next) instead of 128 bytes of
// eight: as a bare in-place copy that shape streams at 5.98 TB/s where four back-to-back 128-byte pieces stop at 5.66
// (tools/membench.hip, round 1) -- and 5.66 is what the kernel above reaches.  The price is in the LDS transpose: a row's
// 512 bytes are spread over 32 lanes, so the super-tile passes through LDS in two halves of 256 bytes per channel (17 KiB
// per wave), each half written and read back by the half of the lanes that holds it (exec-masked ds_write/read_b128).
template <typename P, int KIND, int M, bool NT>
__global__ __launch_bounds__(64) void sdsp_iir_wide_kernel(iir_dev_args<typename P::S, typename P::R, M> p)
{
    // whole super-tiles only: channels a multiple of 64, samples a multiple of 512 bytes (the launcher sends every other
    // shape to the kernel above -- predicated edge paths in here cost 70 more registers and the second wave per SIMD)
    using S = typename P::S;
    using R = typename P::R;
    using V = typename vec16<S>::type;
    constexpr int EPV = vec16<S>::n;
    constexpr int ROWB = 512, HALFB = 256;
    constexpr int T = ROWB / (int)sizeof(S); // samples per super-tile
    constexpr int NVH = HALFB / 16;          // vectors per half row
    constexpr int PITCH = HALFB + 16;

    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_iir_smem[];
    unsigned char *tile = sdsp_iir_smem;
    const int lane = threadIdx.x;
    const uint64_t ch0 = (uint64_t)blockIdx.x * 64;
    const uint64_t my_ch = ch0 + lane;

    R y1[M + 1], y2[M + 1], y3[M + 1];
    load_state<R, M>(p, my_ch, y1, y2, y3);

    const int rsel = lane >> 5, col = lane & 31; // row of the pair, 16-byte column of the 512-byte segment
    const int hsel = col >> 4;                   // which LDS pass holds this lane's column
    unsigned char *const xchg = tile + rsel * PITCH + (col & 15) * 16;
    // addresses = uniform base of the row pair (SGPRs) + ONE 32-bit per-lane byte offset (the global_load "saddr" form)
    const uint32_t lane_off = (uint32_t)((rsel * p.stride + (uint64_t)col * EPV) * sizeof(S)); // the host checks the range
    const uint64_t pair_step = 2 * p.stride; // elements between row pairs
    auto at_lane = [&](S *uniform_base) { return reinterpret_cast<S *>(reinterpret_cast<char *>(uniform_base) + lane_off); };
    const uint64_t n_super = p.samples / T;
    for (uint64_t st = 0; st < n_super; st++) {
        V stage[32]; // register i: rows 2i, 2i+1
        S *const tile_base = p.data + ch0 * p.stride + st * T; // uniform
#pragma unroll
        for (int i = 0; i < 32; i++)
            stage[i] = gload16<S, NT>(at_lane(tile_base + i * pair_step));
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (hsel == h) {
#pragma unroll
                for (int i = 0; i < 32; i++)
                    *reinterpret_cast<V *>(xchg + 2 * i * PITCH) = stage[i];
            }
            __syncthreads();
            V *myrow = reinterpret_cast<V *>(tile + lane * PITCH);
#pragma unroll
            for (int v = 0; v < NVH; v++) {
                V x = myrow[v];
                S *xe = reinterpret_cast<S *>(&x);
#pragma unroll
                for (int e = 0; e < EPV; e++)
                    xe[e] = (S)cascade_step<R, KIND, M, P::fused>((R)xe[e], p, y1, y2, y3);
                myrow[v] = x;
            }
            __syncthreads();
            if (hsel == h) {
#pragma unroll
                for (int i = 0; i < 32; i++)
                    stage[i] = *reinterpret_cast<const V *>(xchg + 2 * i * PITCH);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 32; i++)
            gstore16<S, NT>(at_lane(tile_base + i * pair_step), stage[i]);
    }
    if (p.samples)
        store_state<R, M>(p, my_ch, y1, y2, y3);
}

// ---- LDS-DMA ring variant (round 3).  One wave per workgroup owns 64 channels, as above, but the tile never passes
// through the register file on its way in: `global_load_lds_dwordx4` writes a [64 channels x RB bytes] tile straight into
// one of SLOTS LDS slots (1 KiB of LDS per wave instruction: 64 / CPR rows x RB contiguous bytes of each), the lane filters
// its own row IN PLACE in LDS, and only the finished tile is read back (lane-linear, conflict-free) for the streaming
// stores.  So the loads of tile t + SLOTS - 1 are in flight while tile t is being filtered: the wave's memory pipeline has
// no empty phase, and the registers hold one row's worth of samples instead of the 128-VGPR staging tile.
//
// LDS image: an LDS-DMA instruction's destination is M0 + 16 * lane, i.e. lane-linear, and only the SOURCE address is
// per-lane (cdna_hip_programming.md, rule 21).  The 16-byte chunk at (row r, position c) of a slot therefore holds the
// row's chunk c ^ f(r): the swizzle sits on the source address of the fill, on the lane's own row reads / writes, and on
// the destination address of the stores.  f makes the row-per-lane ds_read_b128 (lane groups {0-3,12-15,20-27}, ... of
// MI355X_MICROARCH.md's LDS table; bank = (a / 4) mod 64) and the ds_write_b128 (8 contiguous lanes, mod 32) conflict-free:
// f(r) = r mod 16 for rows of 256 / 512 bytes, (r / 2) mod 8 for rows of 128 bytes.
//
// Ordering (hipcc does not count asm loads): a slot is read only behind an explicit `s_waitcnt vmcnt(W)`, W = the
// vector-memory operations issued after that slot's fill (later fills and the stores of the tiles in between; all counted
// in issue order); a slot is refilled only after its read-back has been consumed by the stores in front of the refill.
template <int RB> __device__ __forceinline__ constexpr int iir_dma_swz(int r)
{
    return RB >= 256 ? (r & 15) : ((r >> 1) & 7);
}

template <bool NT> __device__ __forceinline__ void glds16(const void *uniform_base, uint32_t lane_off, uint32_t lds_dst)
{
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane_off), "s"(uniform_base), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane_off), "s"(uniform_base), "s"(lds_dst)
                     : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most `groups` groups of NI operations are outstanding (capped at the counter's 63)
template <int NI, int MAXG> __device__ __forceinline__ void wait_groups(int groups)
{
    if constexpr (MAXG == 0) {
        wait_vmcnt<0>();
    } else {
        if (groups >= MAXG)
            wait_vmcnt<(NI * MAXG > 63 ? 63 / NI * NI : NI * MAXG)>();
        else
            wait_groups<NI, MAXG - 1>(groups);
    }
}

// hipcc does not count asm loads: its own `s_waitcnt vmcnt(0)` for the state loads must sit in front of the first fill, not
// at the state's first use inside the tile loop (where it would drain the fill in flight, every iteration)
template <typename R, int M> __device__ __forceinline__ void settle_state(R (&y1)[M + 1], R (&y2)[M + 1], R (&y3)[M + 1])
{
#pragma unroll
    for (int j = 0; j <= M; j++)
        asm volatile("" : "+v"(y1[j]), "+v"(y2[j]), "+v"(y3[j]));
}

template <typename P, int KIND, int M, int RB, int SLOTS, bool NT, bool PASS>
__global__ __launch_bounds__(64) void sdsp_iir_dma_kernel(iir_dev_args<typename P::S, typename P::R, M> p)
{
    // whole tiles only: channels a multiple of 64, samples a multiple of RB bytes (the launcher sends other shapes to the
    // super-tile kernel: identical arithmetic)
    using S = typename P::S;
    using R = typename P::R;
    using V = typename vec16<S>::type;
    constexpr int EPV = vec16<S>::n;
    constexpr int CPR = RB / 16;  // 16-byte chunks per row of a tile = fills per tile
    constexpr int RPI = 64 / CPR; // rows one fill instruction covers
    constexpr int NI = CPR;
    constexpr int TILE = 64 * RB;
    constexpr int T = RB / (int)sizeof(S); // samples per tile row

    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_iir_smem[];
    const int lane = threadIdx.x;
    const uint64_t ch0 = (uint64_t)blockIdx.x * 64;
    const uint64_t my_ch = ch0 + lane;

    R y1[M + 1], y2[M + 1], y3[M + 1];
    load_state<R, M>(p, my_ch, y1, y2, y3);
    settle_state<R, M>(y1, y2, y3);

    const int q = lane / CPR, c = lane % CPR;
    const uint64_t row_bytes = p.stride * sizeof(S);
    const uint32_t lane_row = (uint32_t)(q * row_bytes); // the host checks the range
    const char *const wg_base = reinterpret_cast<const char *>(p.data + ch0 * p.stride);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)sdsp_iir_smem;
    const uint64_t n_tiles = p.samples / T;
    const int fl = iir_dma_swz<RB>(lane);

    // byte offset of this lane's chunk of fill / store instruction i from the instruction's uniform row base
    auto lane_off = [&](int i) { return lane_row + 16u * (uint32_t)(c ^ iir_dma_swz<RB>(RPI * i + q)); };
    auto fill = [&](uint64_t t, int slot) {
        const char *tb = wg_base + t * RB;
#pragma unroll
        for (int i = 0; i < NI; i++)
            glds16<NT>(tb + (uint64_t)i * RPI * row_bytes, lane_off(i), lds0 + (uint32_t)(slot * TILE + i * 1024));
    };

    for (int s = 0; s < SLOTS; s++)
        if ((uint64_t)s < n_tiles)
            fill(s, s);
    int slot = 0;
    for (uint64_t t = 0; t < n_tiles; t++) {
        // operations issued after tile t's fill: the stores of the tiles since (at most SLOTS - 1) and the later fills
        const uint64_t st = t < (uint64_t)(SLOTS - 1) ? t : (uint64_t)(SLOTS - 1);
        const uint64_t fl_after = n_tiles - 1 - t < (uint64_t)(SLOTS - 1) ? n_tiles - 1 - t : (uint64_t)(SLOTS - 1);
        wait_groups<NI, 2 * (SLOTS - 1)>((int)(st + fl_after));
        unsigned char *const sl = sdsp_iir_smem + slot * TILE;
        if constexpr (!PASS) {
            unsigned char *const row = sl + lane * RB;
            V x[CPR];
#pragma unroll
            for (int k = 0; k < CPR; k++)
                x[k] = *reinterpret_cast<const V *>(row + 16 * (k ^ fl));
#pragma unroll
            for (int k = 0; k < CPR; k++) {
                S *xe = reinterpret_cast<S *>(&x[k]);
#pragma unroll
                for (int e = 0; e < EPV; e++)
                    xe[e] = (S)cascade_step<R, KIND, M, P::fused>((R)xe[e], p, y1, y2, y3);
                *reinterpret_cast<V *>(row + 16 * (k ^ fl)) = x[k];
            }
        }
        char *const tb = const_cast<char *>(wg_base) + t * RB;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const V v = *reinterpret_cast<const V *>(sl + (i * 64 + lane) * 16);
            gstore16<S, true>(reinterpret_cast<S *>(tb + (uint64_t)i * RPI * row_bytes + lane_off(i)), v);
        }
        if (t + SLOTS < n_tiles) // (the refill in FRONT of these stores measured the same: 59.0 against 56.9-59.1 %)
            fill(t + SLOTS, slot);
        slot = slot + 1 == SLOTS ? 0 : slot + 1;
    }
    wait_vmcnt<0>();
    if (p.samples)
        store_state<R, M>(p, my_ch, y1, y2, y3);
}

// ---- landing-slot variant (round 3): the LDS-DMA fill of the kernel above with the super-tile kernel's burst shape.
// What the ring kernel's lab runs established (profiles/r03_iir_lab.md): with 128- or 256-byte row segments per fill the
// transport alone (no recurrence) stops at 59-61 % of HBM peak whatever the ring depth, occupancy, swizzle or issue order
// -- the figure of the "one 128-byte piece per step" copy of DESIGN.md section 5: DRAM wants 512 contiguous bytes per
// channel per burst.  A ring of 512-byte-row slots does not fit (2 x 32 KiB per wave = two waves per CU).  So this kernel
// keeps ONE 32 KiB landing slot per wave: tile t's rows move from the slot into registers (row-per-lane, 128 VGPRs), the
// slot is refilled with tile t + 1 AT ONCE -- it lands during the ~5 us recurrence of tile t -- and the results leave the
// way the super-tile kernel's do: 128-byte pieces through an 8 KiB transpose buffer back into the same registers, then
// row-group-major streaming stores (512 contiguous bytes per channel per burst).  40 KiB of LDS per one-wave workgroup:
// four per CU, one per SIMD.
template <typename P, int KIND, int M, bool NT, bool PASS>
__global__ __launch_bounds__(64) void sdsp_iir_landing_kernel(iir_dev_args<typename P::S, typename P::R, M> p)
{
    using S = typename P::S;
    using R = typename P::R;
    using V = typename vec16<S>::type;
    constexpr int EPV = vec16<S>::n;
    constexpr int RB = 512, CPR = RB / 16, RPI = 64 / CPR, NI = CPR, TILE = 64 * RB;
    constexpr int T = RB / (int)sizeof(S); // samples per tile row
    constexpr int PB = 128, PCH = PB / 16, NP = RB / PB; // output pieces: 128 bytes of a row, 8 chunks, 4 per tile

    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_iir_smem[];
    unsigned char *const land = sdsp_iir_smem;
    // 64 rows x 128 bytes, chunk k of row r at position k ^ (r & 7): the row-per-lane ds_write_b128 (8 contiguous lanes per
    // LDS cycle, 32 banks) then hits eight distinct 16-byte columns; the read-back is lane-linear.  (The first version used the ring
    // kernel's 128-byte-row map (r >> 1) & 7, which is made for row-per-lane READS: 33 % LDS conflict cycles, r03_lds_bank_conflicts.md)
    unsigned char *const xp = sdsp_iir_smem + TILE;
    const int lane = threadIdx.x;
    const uint64_t ch0 = (uint64_t)blockIdx.x * 64;
    const uint64_t my_ch = ch0 + lane;

    R y1[M + 1], y2[M + 1], y3[M + 1];
    load_state<R, M>(p, my_ch, y1, y2, y3);
    settle_state<R, M>(y1, y2, y3);

    const int q = lane / CPR, c = lane % CPR;
    const uint64_t row_bytes = p.stride * sizeof(S);
    const uint32_t lane_row = (uint32_t)(q * row_bytes); // the host checks the range
    char *const wg_base = reinterpret_cast<char *>(p.data + ch0 * p.stride);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)sdsp_iir_smem;
    const uint64_t n_tiles = p.samples / T;
    const int fl = iir_dma_swz<RB>(lane);

    auto fill = [&](uint64_t t) {
        const char *tb = wg_base + t * RB;
#pragma unroll
        for (int i = 0; i < NI; i++)
            glds16<NT>(tb + (uint64_t)i * RPI * row_bytes, lane_row + 16u * (uint32_t)(c ^ iir_dma_swz<RB>(RPI * i + q)),
                       lds0 + (uint32_t)(i * 1024));
    };
    // output side: store instruction (i, j) covers rows 8 i .. 8 i + 7, piece j; this lane: row 8 i + sub, position pc
    const int sub = lane / PCH, pc = lane % PCH;
    const uint32_t out_row = (uint32_t)(sub * row_bytes);
    auto out_off = [&](int) { return out_row + 16u * (uint32_t)(pc ^ sub); }; // row 8 i + sub: (8 i + sub) & 7 = sub
    const int fo = lane & 7;

    if (n_tiles)
        fill(0);
    for (uint64_t t = 0; t < n_tiles; t++) {
        // behind tile t's fill: the stores of tile t - 1 (NP * 8 = 32 operations)
        if (t == 0)
            wait_vmcnt<0>();
        else
            wait_vmcnt<NP * 8>();
        V x[CPR];
        {
            const unsigned char *const row = land + lane * RB;
#pragma unroll
            for (int k = 0; k < CPR; k++)
                x[k] = *reinterpret_cast<const V *>(row + 16 * (k ^ fl));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the slot has been read: tile t + 1 may land
        if (t + 1 < n_tiles)
            fill(t + 1);
        // (one fill instruction behind every chunk's arithmetic instead of 32 in a block here measured the same: 70.8-71.1 %
        // against 70.5-71.0 %, profiles/r03_iir_lab.md)
        if constexpr (!PASS) {
#pragma unroll
            for (int k = 0; k < CPR; k++) {
                S *xe = reinterpret_cast<S *>(&x[k]);
#pragma unroll
                for (int e = 0; e < EPV; e++)
                    xe[e] = (S)cascade_step<R, KIND, M, P::fused>((R)xe[e], p, y1, y2, y3);
            }
        }
        // transpose piece by piece: x[8 j + k] (row = lane, chunk k of piece j)  ->  x[8 j + i] (row 8 i + sub, position pc)
#pragma unroll
        for (int j = 0; j < NP; j++) {
            unsigned char *const row = xp + lane * PB;
#pragma unroll
            for (int k = 0; k < PCH; k++)
                *reinterpret_cast<V *>(row + 16 * (k ^ fo)) = x[PCH * j + k];
#pragma unroll
            for (int i = 0; i < 8; i++)
                x[PCH * j + i] = *reinterpret_cast<const V *>(xp + (i * 64 + lane) * 16);
        }
        char *const tb = wg_base + t * RB;
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < NP; j++)
                gstore16<S, true>(reinterpret_cast<S *>(tb + (uint64_t)i * 8 * row_bytes + j * PB + out_off(i)), x[PCH * j + i]);
    }
    wait_vmcnt<0>();
    if (p.samples)
        store_state<R, M>(p, my_ch, y1, y2, y3);
}

// ---- interleaved ("wire") layout, SURVEY 8(f)-2: data[s * stride + c], channels contiguous.  No
// transpose at all: a lane owns VEC ADJACENT channels and runs their recurrences side by side
// (independent dependency chains), a wave moves 64*VEC*sizeof(R) contiguous bytes of one sample row
// per instruction, and U rows are fetched ahead of the U rows being filtered.
template <typename R, int VEC> struct alignas(sizeof(R) * VEC) chan_vec {
    R v[VEC];
};
template <typename R, int VEC, bool NT> __device__ __forceinline__ chan_vec<R, VEC> gload_cv(const R *p)
{
    using N = R __attribute__((ext_vector_type(VEC)));
    chan_vec<R, VEC> out;
    if constexpr (VEC == 1) {
        out.v[0] = NT ? __builtin_nontemporal_load(p) : *p;
    } else if constexpr (NT) {
        const N v = __builtin_nontemporal_load(reinterpret_cast<const N *>(p));
        __builtin_memcpy(&out, &v, sizeof(out));
    } else {
        out = *reinterpret_cast<const chan_vec<R, VEC> *>(p);
    }
    return out;
}
template <typename R, int VEC, bool NT> __device__ __forceinline__ void gstore_cv(R *p, const chan_vec<R, VEC> &a)
{
    using N = R __attribute__((ext_vector_type(VEC)));
    if constexpr (VEC == 1) {
        if constexpr (NT)
            __builtin_nontemporal_store(a.v[0], p);
        else
            *p = a.v[0];
    } else if constexpr (NT) {
        N v;
        __builtin_memcpy(&v, &a, sizeof(a));
        __builtin_nontemporal_store(v, reinterpret_cast<N *>(p));
    } else {
        *reinterpret_cast<chan_vec<R, VEC> *>(p) = a;
    }
}

// U rows are being filtered while the next U are in flight (double buffer).  Measured with 16-byte lanes on the
// BASELINE config-4 shape: U = 2: 51 %, 4: 63 %, 8: 66 % of HBM peak; a single ring of 8 / 16 row registers that
// is refilled row by row (same rows in flight, half the registers): 62 % / 61 %.
template <typename P, int KIND, int M, bool NT, int VEC, int U>
__global__ __launch_bounds__(64) void sdsp_iir_interleaved_kernel(iir_dev_args<typename P::S, typename P::R, M> p)
{
    using S = typename P::S;
    using R = typename P::R;
    using V = chan_vec<S, VEC>;
    const uint64_t c0 = ((uint64_t)blockIdx.x * 64 + threadIdx.x) * VEC;
    if (c0 >= p.channels)
        return; // channels is a multiple of VEC (checked by the host)

    R y1[VEC][M + 1], y2[VEC][M + 1], y3[VEC][M + 1];
#pragma unroll
    for (int e = 0; e < VEC; e++)
        load_state<R, M>(p, c0 + e, y1[e], y2[e], y3[e]);

    S *col = p.data + c0;
    V cur[U], nxt[U];
    auto fetch = [&](V (&dst)[U], uint64_t s0) {
        if (s0 + U <= p.samples) { // whole batch inside the stream: no per-row predicates
            S *row = col + s0 * p.stride;
#pragma unroll
            for (int u = 0; u < U; u++)
                dst[u] = gload_cv<S, VEC, NT>(row + u * p.stride);
            return;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            dst[u] = V{};
            if (s0 + u < p.samples)
                dst[u] = gload_cv<S, VEC, NT>(col + (s0 + u) * p.stride);
        }
    };
    fetch(cur, 0);
    for (uint64_t s0 = 0; s0 < p.samples; s0 += U) {
        if (s0 + U < p.samples)
            fetch(nxt, s0 + U);
        if (s0 + U <= p.samples) {
            S *row = col + s0 * p.stride;
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int e = 0; e < VEC; e++)
                    cur[u].v[e] = (S)cascade_step<R, KIND, M, P::fused>((R)cur[u].v[e], p, y1[e], y2[e], y3[e]);
                gstore_cv<S, VEC, NT>(row + u * p.stride, cur[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (s0 + u < p.samples) {
#pragma unroll
                    for (int e = 0; e < VEC; e++)
                        cur[u].v[e] = (S)cascade_step<R, KIND, M, P::fused>((R)cur[u].v[e], p, y1[e], y2[e], y3[e]);
                    gstore_cv<S, VEC, NT>(col + (s0 + u) * p.stride, cur[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            cur[u] = nxt[u];
    }
#pragma unroll
    for (int e = 0; e < VEC; e++)
        store_state<R, M>(p, c0 + e, y1[e], y2[e], y3[e]);
}

template <typename P, int M> iir_dev_args<typename P::S, typename P::R, M> make_args(const iir_args &a)
{
    using R = typename P::R;
    iir_dev_args<typename P::S, R, M> p;
    p.data = reinterpret_cast<typename P::S *>(a.data);
    p.state = reinterpret_cast<R *>(a.state);
    p.channels = a.channels;
    p.samples = a.samples;
    p.stride = a.stride;
    p.gain = (R)a.gain;
    for (int j = 0; j < M; j++) {
        p.a1[j] = (R)a.a1[j];
        p.a2[j] = (R)a.a2[j];
        p.b1[j] = (R)a.b1[j];
        p.b2[j] = (R)a.b2[j];
    }
    return p;
}

// the LDS-DMA ring kernel on whole tiles; SLOTS x 64 x RB bytes of LDS per one-wave workgroup
template <typename P, int KIND, int M, int RB, int SLOTS, bool NT, bool PASS>
int launch_dma(const iir_args &a, hipStream_t stream)
{
    const auto p = make_args<P, M>(a);
    const uint64_t blocks = a.channels / 64;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "too many channels for one launch");
    constexpr size_t lds = (size_t)SLOTS * 64 * RB;
    static std::atomic<uint64_t> lds_ok{ 0 };
    auto kern = sdsp_iir_dma_kernel<P, KIND, M, RB, SLOTS, NT, PASS>;
    if (lds > 64 * 1024)
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, lds_ok))
            return rc;
    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(64), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("iir launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

// the landing-slot kernel on whole tiles; 40 KiB of LDS per one-wave workgroup
template <typename P, int KIND, int M, bool PASS> int launch_landing(const iir_args &a, hipStream_t stream)
{
    const auto p = make_args<P, M>(a);
    const uint64_t blocks = a.channels / 64;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "too many channels for one launch");
    constexpr size_t lds = 64 * 512 + 64 * 128;
    hipLaunchKernelGGL((sdsp_iir_landing_kernel<P, KIND, M, true, PASS>), dim3((uint32_t)blocks), dim3(64), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("iir launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

// ---- which kernel serves (precision, sections, shape, variant): ONE function, used by the launcher and by the name query
// (sdsp_hip_iir_plan_kernel) that bench.py / the profiles match rocprofv3 rows with.
//   variant 0  default: the landing-slot kernel where it measured faster -- f32 recurrences of m_t <= 4 on whole tiles
//              (70.6-71.1 % against 69.9-70.1 % for the super-tile kernel, cfg 4, one call, three times; in double and in the
//              mixed mode its one wave per SIMD is VALU-bound: 68.3 / 64.6 % against 71.8 / 67.5 %) -- else the super-tile kernel
//   variant 1  wide super-tile (512 contiguous bytes per instruction); whole super-tiles only, else as variant 3
//   variant 2  direct kernel (lane = channel, scalar accesses): any alignment, any length
//   variant 3  super-tile kernel (round 2's default)
//   variants 10 / 19 / 20 (lab, f32 casc_2o_iir<4> only): LDS-DMA ring 256 B x 2; landing slot and ring without the recurrence
// Shapes the vector kernels cannot address (not 16-byte aligned) run the direct kernel whatever the variant.
enum iir_kernel_id { IIR_K_SUPERTILE, IIR_K_WIDE, IIR_K_DIRECT, IIR_K_LANDING, IIR_K_RING_LAB, IIR_K_LANDING_COPY_LAB, IIR_K_RING_COPY_LAB, IIR_K_BAD };

const char *iir_kernel_name(iir_kernel_id id)
{
    switch (id) {
    case IIR_K_SUPERTILE: return "sdsp_iir_supertile_kernel";
    case IIR_K_WIDE: return "sdsp_iir_wide_kernel";
    case IIR_K_DIRECT: return "sdsp_iir_direct_kernel";
    case IIR_K_LANDING:
    case IIR_K_LANDING_COPY_LAB: return "sdsp_iir_landing_kernel";
    case IIR_K_RING_LAB:
    case IIR_K_RING_COPY_LAB: return "sdsp_iir_dma_kernel";
    default: return "none";
    }
}

iir_kernel_id iir_select(int precision, const iir_args &a, int variant)
{
    const size_t ss = precision == SDSP_HIP_F64 ? 8 : 4; // sample size (the mixed mode stores floats)
    if (a.sections > 8)
        return IIR_K_DIRECT; // m_t = 10 .. 16: correct, not tuned
    const bool aligned = ((uintptr_t)a.data % 16 == 0) && ((a.stride * ss) % 16 == 0) && ((a.samples * ss) % 16 == 0);
    if (variant == 2 || !aligned)
        return IIR_K_DIRECT;
    // whole [64 channels x 512 bytes] tiles, per-lane row offsets within 32 bits
    const bool tiles = a.channels % 64 == 0 && (a.samples * ss) % 512 == 0 && a.stride * ss * 8 < (1ull << 32);
    const bool lab = precision == SDSP_HIP_F32 && a.kind == SDSP_HIP_IIR_GENERIC && a.sections == 4 && tiles;
    switch (variant) {
    case 0: return (precision == SDSP_HIP_F32 && a.sections <= 4 && tiles) ? IIR_K_LANDING : IIR_K_SUPERTILE;
    case 1: return (tiles && (a.stride + 128) * ss < (1ull << 32)) ? IIR_K_WIDE : IIR_K_SUPERTILE;
    case 3: return IIR_K_SUPERTILE;
    case 10: return lab ? IIR_K_RING_LAB : IIR_K_BAD;
    case 19: return lab ? IIR_K_LANDING_COPY_LAB : IIR_K_BAD;
    case 20: return lab ? IIR_K_RING_COPY_LAB : IIR_K_BAD;
    default: return IIR_K_BAD;
    }
}

template <typename P, int KIND, int M> int launch_km(const iir_args &a, iir_kernel_id id, hipStream_t stream)
{
    const auto p = make_args<P, M>(a);
    constexpr bool f32 = sizeof(typename P::S) == 4 && sizeof(typename P::R) == 4;
    const uint64_t blocks = id == IIR_K_DIRECT ? (a.channels + 255) / 256 : (a.channels + 63) / 64;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "too many channels for one launch");
    switch (id) {
    case IIR_K_DIRECT:
        hipLaunchKernelGGL((sdsp_iir_direct_kernel<P, KIND, M>), dim3((uint32_t)blocks), dim3(256), 0, stream, p);
        break;
    case IIR_K_SUPERTILE:
        hipLaunchKernelGGL((sdsp_iir_supertile_kernel<P, KIND, M, true, 4>), dim3((uint32_t)blocks), dim3(64), 64 * (128 + 16), stream, p);
        break;
    case IIR_K_WIDE:
        hipLaunchKernelGGL((sdsp_iir_wide_kernel<P, KIND, M, true>), dim3((uint32_t)blocks), dim3(64), 64 * (256 + 16), stream, p);
        break;
    case IIR_K_LANDING:
        if constexpr (f32 && M <= 4)
            return launch_landing<P, KIND, M, false>(a, stream);
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "the landing-slot kernel is built for f32, m_t <= 4");
    case IIR_K_RING_LAB:
    case IIR_K_RING_COPY_LAB:
    case IIR_K_LANDING_COPY_LAB:
        if constexpr (f32 && M == 4 && KIND == SDSP_HIP_IIR_GENERIC) {
            if (id == IIR_K_RING_LAB)
                return launch_dma<P, KIND, M, 256, 2, true, false>(a, stream);
            if (id == IIR_K_RING_COPY_LAB)
                return launch_dma<P, KIND, M, 256, 2, true, true>(a, stream);
            return launch_landing<P, KIND, M, true>(a, stream);
        }
        return fail(SDSP_HIP_ERR_INVALID_ARG, "lab variants exist for f32 casc_2o_iir<4> only");
    default:
        return fail(SDSP_HIP_ERR_INVALID_ARG, "unknown IIR kernel variant");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("iir launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

// m_t = 10 .. 16 (the reference accepts any even M, casc_2o_iir.h:25): correct through the direct kernel, not tuned
template <typename P, int KIND, int M> int launch_direct(const iir_args &a, hipStream_t stream)
{
    const auto p = make_args<P, M>(a);
    const uint64_t blocks = (a.channels + 255) / 256;
    hipLaunchKernelGGL((sdsp_iir_direct_kernel<P, KIND, M>), dim3((uint32_t)blocks), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("iir launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

template <typename P, int KIND> int launch_k(const iir_args &a, iir_kernel_id id, hipStream_t stream)
{
    switch (a.sections) {
    case 2: return launch_km<P, KIND, 2>(a, id, stream);
    case 4: return launch_km<P, KIND, 4>(a, id, stream);
    case 6: return launch_km<P, KIND, 6>(a, id, stream);
    case 8: return launch_km<P, KIND, 8>(a, id, stream);
    case 10: return launch_direct<P, KIND, 10>(a, stream);
    case 12: return launch_direct<P, KIND, 12>(a, stream);
    case 14: return launch_direct<P, KIND, 14>(a, stream);
    case 16: return launch_direct<P, KIND, 16>(a, stream);
    default: return fail(SDSP_HIP_ERR_UNSUPPORTED, "sections must be even and at most 16");
    }
}

template <typename P> int launch_r(const iir_args &a, iir_kernel_id id, hipStream_t stream)
{
    switch (a.kind) {
    case SDSP_HIP_IIR_GENERIC: return launch_k<P, SDSP_HIP_IIR_GENERIC>(a, id, stream);
    case SDSP_HIP_IIR_LP: return launch_k<P, SDSP_HIP_IIR_LP>(a, id, stream);
    case SDSP_HIP_IIR_HP: return launch_k<P, SDSP_HIP_IIR_HP>(a, id, stream);
    case SDSP_HIP_IIR_BP: return launch_k<P, SDSP_HIP_IIR_BP>(a, id, stream);
    default: return fail(SDSP_HIP_ERR_INVALID_ARG, "unknown IIR kind");
    }
}
} // namespace

namespace
{
template <typename P, int KIND, int M, int VEC, int U>
int launch_il_v(const iir_args &a, bool nt, hipStream_t stream)
{
    using S = typename P::S;
    if (((uintptr_t)a.data % (sizeof(S) * VEC)) || ((a.stride * sizeof(S)) % (sizeof(S) * VEC)) || (a.channels % VEC))
        return fail(SDSP_HIP_ERR_INVALID_ARG, "interleaved layout: data/stride/channels must be multiples of the lane width");
    const auto p = make_args<P, M>(a);
    const uint64_t blocks = (a.channels / VEC + 63) / 64;
    if (blocks > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "too many channels for one launch");
    if (nt)
        hipLaunchKernelGGL((sdsp_iir_interleaved_kernel<P, KIND, M, true, VEC, U>), dim3((uint32_t)blocks), dim3(64), 0, stream, p);
    else
        hipLaunchKernelGGL((sdsp_iir_interleaved_kernel<P, KIND, M, false, VEC, U>), dim3((uint32_t)blocks), dim3(64), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("iir interleaved launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
// variants: 0 = 16-byte lanes (4 f32 / 2 f64 channels per lane) when the shape allows, streaming, eight rows in flight
// (65.9 % of HBM peak, round 1); 1 = 8-byte lanes (58 %); 2 = 4-byte lanes, f32 only (58 %).  Narrower shapes fall through.
// Measured and dropped: four rows in flight 63.3 %, 8-byte lanes with the default cache policy 57 %.
template <typename P, int KIND, int M> int launch_il_km(const iir_args &a, int variant, hipStream_t stream)
{
    using S = typename P::S;
    constexpr int V8 = 8 / (int)sizeof(S), V16 = 16 / (int)sizeof(S);
    const bool a16 = !((uintptr_t)a.data % 16) && !((a.stride * sizeof(S)) % 16) && !(a.channels % V16);
    const bool a8 = !((uintptr_t)a.data % 8) && !((a.stride * sizeof(S)) % 8) && !(a.channels % V8);
    if (variant < 0 || variant > 2)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "unknown IIR kernel variant");
    if (variant == 0 && a16)
        return launch_il_v<P, KIND, M, V16, 8>(a, true, stream);
    if constexpr (sizeof(S) == 4) {
        if (variant == 2 || !a8)
            return launch_il_v<P, KIND, M, 1, 8>(a, true, stream);
    }
    if (a8)
        return launch_il_v<P, KIND, M, V8, 8>(a, true, stream);
    return fail(SDSP_HIP_ERR_INVALID_ARG, "interleaved layout needs 8-byte aligned rows");
}
template <typename P, int KIND> int launch_il_k(const iir_args &a, int variant, hipStream_t stream)
{
    switch (a.sections) {
    case 2: return launch_il_km<P, KIND, 2>(a, variant, stream);
    case 4: return launch_il_km<P, KIND, 4>(a, variant, stream);
    case 6: return launch_il_km<P, KIND, 6>(a, variant, stream);
    case 8: return launch_il_km<P, KIND, 8>(a, variant, stream);
    // m_t = 10 .. 16: one channel per lane, four rows in flight (correct, not tuned)
    case 10: return launch_il_v<P, KIND, 10, 1, 4>(a, true, stream);
    case 12: return launch_il_v<P, KIND, 12, 1, 4>(a, true, stream);
    case 14: return launch_il_v<P, KIND, 14, 1, 4>(a, true, stream);
    case 16: return launch_il_v<P, KIND, 16, 1, 4>(a, true, stream);
    default: return fail(SDSP_HIP_ERR_UNSUPPORTED, "sections must be even and at most 16");
    }
}
template <typename P> int launch_il_r(const iir_args &a, int variant, hipStream_t stream)
{
    switch (a.kind) {
    case SDSP_HIP_IIR_GENERIC: return launch_il_k<P, SDSP_HIP_IIR_GENERIC>(a, variant, stream);
    case SDSP_HIP_IIR_LP: return launch_il_k<P, SDSP_HIP_IIR_LP>(a, variant, stream);
    case SDSP_HIP_IIR_HP: return launch_il_k<P, SDSP_HIP_IIR_HP>(a, variant, stream);
    case SDSP_HIP_IIR_BP: return launch_il_k<P, SDSP_HIP_IIR_BP>(a, variant, stream);
    default: return fail(SDSP_HIP_ERR_INVALID_ARG, "unknown IIR kind");
    }
}
} // namespace

// data[s * stride + c]: sample-major ("interleaved") layout
int launch_iir_interleaved(int precision, const iir_args &a, int variant, void *stream)
{
    if (a.channels == 0 || a.samples == 0)
        return SDSP_HIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (precision == SDSP_HIP_F32_F64STATE)
        return launch_il_r<prec_mix>(a, variant, s);
    return precision == SDSP_HIP_F64 ? launch_il_r<prec_f64>(a, variant, s) : launch_il_r<prec_f32>(a, variant, s);
}

int launch_iir(int precision, const iir_args &a, int variant, void *stream)
{
    if (a.channels == 0 || a.samples == 0)
        return SDSP_HIP_OK;
    if ((a.channels + 255) / 256 > 0x7fffffffull)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "too many channels for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const iir_kernel_id id = iir_select(precision, a, variant);
    if (id == IIR_K_BAD)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "unknown IIR kernel variant");
    if (precision == SDSP_HIP_F32_F64STATE)
        return launch_r<prec_mix>(a, id, s);
    return precision == SDSP_HIP_F64 ? launch_r<prec_f64>(a, id, s) : launch_r<prec_f32>(a, id, s);
}

const char *iir_kernel_for(int precision, const iir_args &a, int variant) { return iir_kernel_name(iir_select(precision, a, variant)); }
} // namespace sdsp_hip

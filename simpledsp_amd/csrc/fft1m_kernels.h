// fft1m_kernels.h -- device code of the batched N = 2^20 radix-2 complex f32 FFT (BASELINE config 3) for gfx950.
// Included by fft1m.hip (the product) and by tools/lab_fft1m.hip (the measurement harness, which also builds the
// MATH = false data movers with the same access patterns).
//
// sdsp::fft_radix2<T, 2^20> (fft.h:258-299) cannot even be compiled in the reference (its table would be 320 MiB
// of constexpr data); here the 20 radix-2 butterfly stages run as a four-step decomposition N = 1024 x 1024 with
// the transform viewed as a row-major [n1][n2] matrix:
//
//   pass 1 (cols_tile)  for 16 adjacent columns n2: ten radix-2 stages over n1 (stride 1024), times the inter-pass
//                       twiddle W_N^(n2*k1), written to the intermediate
//   pass 2 (rows_tile)  for 16 adjacent rows k1: ten radix-2 stages over n2, written transposed,
//                       X[k1 + 1024*k2], back into the caller's buffer
//
// Both passes use the same building block: 512 threads = 16 sequences x 32 threads, 32 points per thread in
// registers, two register passes of five radix-2 DIF stages each (fft32.h), ONE exchange through LDS.  The exchange
// moves the real and the imaginary plane separately, so a 1024 x 16 tile costs 64 KiB instead of 128 KiB and two
// workgroups fit a CU.  Stage twiddles: five per-thread values W_1024^(2^s u) (from an LDS copy of W_1024) times
// compile-time W_32 constants.  LDS planes are XOR-swizzled so all ds_read/ds_write_b32 are bank-conflict free.
//
// Two schedules over these tiles:
//   * sdsp_fft1m_fused: ONE persistent launch per batch.  Workgroups draw tickets from a global counter; the ticket
//     order interleaves pass-1 tiles of transform t + D with pass-2 tiles of transform t, so the two passes share the
//     CUs all the time, the intermediate of a transform is consumed a few microseconds after it was produced (a ring
//     of R << 32 transforms that stays in the 256 MiB Infinity Cache) and HBM sees the compulsory read and write
//     once each.  Hand-off between workgroups: agent-scope release / acquire on per-transform arrival counters.
//   * sdsp_fft1m_cols + sdsp_fft1m_rows: two launches per chunk of 32 transforms (round 1's schedule; kept as the
//     plan's variant 1 for A/B measurements).
#pragma once

#include <hip/hip_runtime.h>

#include "fft32.h"
#include "handoff.h"

namespace sdsp_hip
{
namespace fft1m
{
using namespace fft32;

constexpr int kTile = 16;      // sequences per tile
constexpr int kThreads = 512;  // 16 sequences x 32 threads
constexpr int kTiles = 1024 / kTile;
// dynamic LDS: one real plane [row][col] (64 KiB), W_1024 (8 KiB), the column part of the inter-pass twiddle (4 KiB),
// the fused kernel's ticket mailbox (16 B), pass 2's thread twiddles [stage][lane] = W_1024^(lane << stage) (1.25 KiB:
// gathering them from the W_1024 copy is a 2- to 16-way bank conflict, 7.8 % of the kernel's LDS cycles)
constexpr size_t kPlaneBytes = 1024 * kTile * sizeof(float);
constexpr size_t kMailOffset = kPlaneBytes + 1024 * sizeof(float2) + 32 * kTile * sizeof(float2);
constexpr size_t kRowTwOffset = kMailOffset + 16;
constexpr size_t kLdsBytes = kRowTwOffset + 5 * 32 * sizeof(float2);

// Intermediate layouts.  ROWS: the [k1][n2] matrix itself (pass 1 writes 128-B segments 8 KiB apart, pass 2 reads
// whole 8 KiB rows).  BLOCKED: [n2 / 16][k1][n2 % 16] -- a pass-1 tile's output is ONE contiguous 128 KiB block,
// pass 2 gathers 2 KiB pieces (16 rows x 128 B) from each of the 64 blocks.
enum ws_layout { WS_ROWS = 0, WS_BLOCKED = 1 };
// OR-ed into LAYOUT: the intermediate is stored write-through (sc1: straight to the fabric, the line is not kept in the
// XCD's L2) -- the hand-off form of cdna_hip_programming.md Guideline 16 (R1) that needs no release fence.  (16-byte
// write-through stores, two columns per lane after a DPP lane swap, measured no faster: 40.9 / 41.7 % against 41.4 / 42.2 %.)
enum ws_access { WS_SC1_STORES = 2 };
__device__ __forceinline__ void ws_store_sc1(float2 *p, float2 v) { handoff::wt_store(p, v); }
// What a tile does.  MODE_FFT is the product.  The other two exist for tools/lab_fft1m.hip only: the same loads and
// stores without the butterflies (MODE_MOVE), and without the intermediate's traffic either (MODE_HBM_ONLY: what the
// HBM-facing halves of the two passes cost on their own).
enum tile_mode { MODE_MOVE = 0, MODE_FFT = 1, MODE_HBM_ONLY = 2 };

// ---- pass 1: 16 columns of one transform ------------------------------------------------------
// in_x / ws_x: the transform's input matrix / its intermediate.  w1k (LDS): W_1024^j, staged by the caller.
template <bool REV, int MODE, int LAYOUT, bool NT_IN>
__device__ __forceinline__ void cols_tile(const float2 *in_x, float2 *ws_x, uint32_t tile, float *plane, const float2 *w1k,
                                          float2 *qtab)
{
    constexpr bool MATH = MODE == MODE_FFT;
    const uint32_t t = threadIdx.x;
    const uint32_t c = t & 15, u = t >> 4;
    const uint32_t n2 = tile * kTile + c;
    // addresses = wave-uniform base (SGPRs; the per-k part is a compile-time constant) + ONE 32-bit
    // per-thread offset, so the 32 loads / stores share a single offset register
    const float2 *src_tile = in_x + tile * kTile;
    const uint32_t toff = (u * 1024 + c) * 8u; // bytes

    float2 x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = NT_IN ? nt_load(at(src_tile + 32768 * k, toff)) : *at(src_tile + 32768 * k, toff);

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
    const uint32_t bu = brev5(u);
    float2 pw = float2{ 1.0f, 0.0f };
    if constexpr (MATH) {
        // The inter-pass twiddle of output k1 = 32 j + bu of column n2 is W_N^(n2 bu) * W_N^(32 n2 j): the first
        // factor is one value per thread, the second is shared by the 32 threads of a column -- 16 x 32 values per
        // workgroup, one per thread, parked in LDS (read back after the exchange barriers below).  That replaces a
        // polynomial and a conflict-prone table gather per ELEMENT by one conflict-free LDS read and one multiply.
        qtab[u * 16 + c] = twiddle(32u * n2 * u); // thread (c, u) computes j = u
        fft32_dif<REV, true>(x, w1k, u);          // stages with row strides 512 .. 32; twiddles W_1024^(2^s u)

        // exchange rows {u + 32k} -> {32u + k}.  slot(row, col) = (row*16 + col) ^ (((row >> 5) & 1) << 4).
        // Written with two base registers + compile-time offsets (per-element XOR'd addresses would
        // cost 64 VGPRs): writes flip bit 4 for odd k; reads use slot 512u + c + 16*(k ^ (u&1)).
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
        fft32_dif<REV, false>(x, w1k, u); // row strides 16 .. 1
        pw = twiddle(n2 * bu);            // W_N^(n2 bu)
    }

    // position 32u + k now holds Y[k1], k1 = bit_reverse10(32u + k) = 32 * bit_reverse5(k) + bu; times W_N^(n2*k1),
    // stored at row k1 of the intermediate (default cache policy: it is re-read within microseconds)
    const float2 *const qcol = qtab + c;
    float2 *dst_tile;
    uint32_t soff;
    if constexpr ((LAYOUT & 1) == WS_BLOCKED) {
        dst_tile = ws_x + (size_t)tile * (1024 * kTile); // [tile][k1][c]
        soff = (bu * 16 + c) * 8u;
    } else {
        dst_tile = ws_x + tile * kTile; // [k1][n2]
        soff = (bu * 1024 + c) * 8u;
    }
    constexpr int kStep = (LAYOUT & 1) == WS_BLOCKED ? 32 * 16 : 32 * 1024; // float2 elements between k1 and k1 + 32
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if ((k & 7) == 0) // keep at most 8 elements' table reads in flight (register budget)
            __builtin_amdgcn_sched_barrier(0);
        float2 v = x[k];
        if constexpr (MATH)
            v = cmul(v, cmul(pw, qcol[16 * (int)(__brev((uint32_t)k) >> 27)]));
        if constexpr (MODE == MODE_HBM_ONLY) {
            if (v.x == 1.2345e-30f) // never true for the lab's data: keeps the loads alive without the stores
                *at(dst_tile + kStep * (int)(__brev((uint32_t)k) >> 27), soff) = v;
        } else if constexpr ((LAYOUT & WS_SC1_STORES) != 0) {
            ws_store_sc1(at(dst_tile + kStep * (int)(__brev((uint32_t)k) >> 27), soff), v);
        } else {
            *at(dst_tile + kStep * (int)(__brev((uint32_t)k) >> 27), soff) = v;
        }
    }
}

// ---- pass 2: 16 rows of one transform, written transposed ------------------------------------------
template <bool REV, int MODE, int LAYOUT, bool NT_OUT>
__device__ __forceinline__ void rows_tile(const float2 *ws_x, float2 *out_x, uint32_t tile, float *plane, const float2 *w1k,
                                          const float2 *wrow, float scale)
{
    constexpr bool MATH = MODE == MODE_FFT;
    const uint32_t t = threadIdx.x;
    // first register pass: 32 lanes run along a row
    const uint32_t ra = t >> 5, ua = t & 31;
    float2 x[32];
    if constexpr (MODE == MODE_HBM_ONLY) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k] = float2{ (float)t, (float)k };
    } else if constexpr ((LAYOUT & 1) == WS_BLOCKED) {
        // element n2 = ua + 32k of row k1 = 16 tile + ra lives at [(n2 >> 4)][k1][n2 & 15]
        const float2 *src = ws_x + (size_t)tile * (kTile * kTile);
        const uint32_t aoff = ((ua >> 4) * (1024 * kTile) + ra * 16 + (ua & 15)) * 8u;
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k] = *at(src + 2 * k * (1024 * kTile), aoff);
    } else {
        const float2 *src = ws_x + (size_t)tile * kTile * 1024;
        const uint32_t aoff = (ra * 1024 + ua) * 8u; // bytes
#pragma unroll
        for (int k = 0; k < 32; k++)
            x[k] = *at(src + 32 * k, aoff);
    }
    // exchange, and switch the thread mapping so that 16 lanes run across the 16 rows
    const uint32_t rb = t & 15, ub = t >> 4;
    if constexpr (MATH) {
        fft32_dif<REV, true, 0, true>(x, wrow + ua, 32); // stage s: wrow[32 s + ua] = W_1024^(ua << s)
        // write slot ra*1024 + ((ua + 32k) ^ (ra | ((k&1) << 4))): the XOR touches the low 5 bits only
        //   -> bases (ua ^ ra) and (ua ^ ra ^ 16) + 32k;
        // read slot rb*1024 + ((32ub + k) ^ (rb | ((ub&1) << 4))) = rb*1024 + 32ub + (k ^ rb ^ 16(ub&1)):
        //   the register index is XOR'ed with a run-time value, so those 32 addresses are rebuilt from an
        //   opaque value with one v_xor each instead of living in registers across the butterflies.
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
        fft32_dif<REV, false>(x, w1k, ua);
    }

    // position 32ub + k of row k1 holds X[k1 + 1024*k2], k2 = bit_reverse10(32ub + k): 16 lanes write
    // 128 contiguous bytes.  Streaming (non-temporal) store of the final result.
    float2 *dst_tile = out_x + tile * kTile;
    const uint32_t bu = brev5(ub);
    const uint32_t boff = (bu * 1024 + rb) * 8u; // bytes
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if ((k & 7) == 0)
            __builtin_amdgcn_sched_barrier(0);
        float2 v = x[k]; // k2 = bit_reverse5(k)*32 + bit_reverse5(ub)
        if constexpr (REV && MATH) { // reverse_fft::ScaleValues, fft.h:128-132
            v.x *= scale;
            v.y *= scale;
        }
        float2 *dst = at(dst_tile + 32768 * (int)(__brev((uint32_t)k) >> 27), boff);
        if constexpr (NT_OUT)
            nt_store(dst, v);
        else
            *dst = v;
    }
}

__device__ __forceinline__ void stage_w1k(float2 *w1k, const float2 *tw_1024)
{
    reinterpret_cast<float4 *>(w1k)[threadIdx.x] = reinterpret_cast<const float4 *>(tw_1024)[threadIdx.x];
    if (threadIdx.x < 160) { // pass 2's thread twiddles, [stage][lane]
        float2 *wrow = reinterpret_cast<float2 *>(reinterpret_cast<unsigned char *>(w1k) - kPlaneBytes + kRowTwOffset);
        wrow[threadIdx.x] = tw_1024[(threadIdx.x & 31u) << (threadIdx.x >> 5)];
    }
}

// ---- two launches per chunk (variant 1) ----------------------------------------------------------------
template <bool REV, int MODE, int LAYOUT>
__global__ __launch_bounds__(kThreads, 4) void sdsp_fft1m_cols(const float2 *__restrict__ in, float2 *__restrict__ ws,
                                                               const float2 *__restrict__ tw_1024)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft1m_smem[];
    float *plane = reinterpret_cast<float *>(sdsp_fft1m_smem);
    float2 *w1k = reinterpret_cast<float2 *>(sdsp_fft1m_smem + kPlaneBytes);
    float2 *qtab = w1k + 1024;
    stage_w1k(w1k, tw_1024);
    __syncthreads();
    const uint32_t tile = blockIdx.x % kTiles;
    const uint64_t xform = blockIdx.x / kTiles;
    cols_tile<REV, MODE, LAYOUT, true>(in + xform * (1ull << 20), ws + xform * (1ull << 20), tile, plane, w1k, qtab);
}

template <bool REV, int MODE, int LAYOUT>
__global__ __launch_bounds__(kThreads, 4) void sdsp_fft1m_rows(const float2 *__restrict__ ws, float2 *__restrict__ out,
                                                               const float2 *__restrict__ tw_1024, float scale)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft1m_smem[];
    float *plane = reinterpret_cast<float *>(sdsp_fft1m_smem);
    float2 *w1k = reinterpret_cast<float2 *>(sdsp_fft1m_smem + kPlaneBytes);
    stage_w1k(w1k, tw_1024);
    __syncthreads();
    const uint32_t tile = blockIdx.x % kTiles;
    const uint64_t xform = blockIdx.x / kTiles;
    const float2 *wrow = reinterpret_cast<const float2 *>(sdsp_fft1m_smem + kRowTwOffset);
    rows_tile<REV, MODE, LAYOUT, true>(ws + xform * (1ull << 20), out + xform * (1ull << 20), tile, plane, w1k, wrow, scale);
}

// ---- two passes of DIFFERENT chunks in one launch (software pipelining across launches) -----------------------------
// Launch i runs pass 1 of chunk i (HBM read, intermediate write) and pass 2 of chunk i - 1 (intermediate read, HBM
// write) side by side: odd / even workgroups alternate between the two, so every CU holds one of each and the two kinds
// of traffic overlap all the time.  All dependencies cross a kernel boundary (pass 2 of a chunk runs one launch after its
// pass 1; the intermediate is double-buffered), so there is no in-kernel synchronisation at all.
template <bool REV, int MODE, int LAYOUT>
__global__ __launch_bounds__(kThreads, 4) void sdsp_fft1m_mixed(const float2 *in1, float2 *ws1, uint32_t n1, const float2 *ws2,
                                                                float2 *out2, uint32_t n2, const float2 *__restrict__ tw_1024,
                                                                float scale)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft1m_smem[];
    float *plane = reinterpret_cast<float *>(sdsp_fft1m_smem);
    float2 *w1k = reinterpret_cast<float2 *>(sdsp_fft1m_smem + kPlaneBytes);
    float2 *qtab = w1k + 1024;
    stage_w1k(w1k, tw_1024);
    __syncthreads();
    const uint32_t t1 = n1 * kTiles, t2 = n2 * kTiles, both = 2 * (t1 < t2 ? t1 : t2);
    const uint32_t b = blockIdx.x;
    bool first;
    uint32_t item;
    if (b < both) {
        first = (b & 1) == 0;
        item = b >> 1;
    } else {
        first = t1 > t2;
        item = b - both / 2;
    }
    const uint32_t tile = item % kTiles;
    const size_t xoff = (size_t)(item / kTiles) << 20;
    if (first)
        cols_tile<REV, MODE, LAYOUT, true>(in1 + xoff, ws1 + xoff, tile, plane, w1k, qtab);
    else
        rows_tile<REV, MODE, LAYOUT, true>(ws2 + xoff, out2 + xoff, tile, plane, w1k,
                                           reinterpret_cast<const float2 *>(sdsp_fft1m_smem + kRowTwOffset), scale);
}

// ---- one persistent launch per batch ----------------------------------------------------------------------
// Workgroups are bound to one of `queues` independent queues (blockIdx % queues); queue q owns the transforms
// t = q, q + queues, ... and its own ticket counter, so the ticket atomics of the whole grid do not serialise on one
// address.  Within a queue, ticket step s holds the 64 pass-1 tiles of the queue's s-th transform followed by the 64
// pass-2 tiles of its (s - lag)-th; a transform's intermediate lives in slot (q * ring + i % ring) of the workspace.
// A workgroup only ever waits for work of LOWER tickets of its own queue (pass 2 of i waits for pass 1 of i: lag >= 0;
// pass 1 of i waits for pass 2 of i - ring: ring > lag), and a ticket is held by a running workgroup or done -- so the
// grid drains whatever its size and whatever the dispatch order.
// Synchronisation words (zeroed by the host before every launch), every hot word on a line of its own:
//   sync[32 q]                          ticket counter of queue q
//   sync[32 queues]                     abort flag (a bounded spin gave up: results invalid, every workgroup drains)
//   sync[32 (queues + 1) + 32 t]        pass-1 tiles of transform t that have published their output   (target kTiles)
//   sync[32 (queues + 1) + 32 t + 16]   pass-2 tiles of transform t that have finished reading it      (target kTiles)
// Hand-off protocol (cdna_hip_programming.md Guideline 16, form R1): the intermediate is stored WRITE-THROUGH (sc1), every
// storing wave drains its stores, workgroup barrier, ONE lane adds to the arrival counter (relaxed, agent scope); the
// consumer's ONE lane polls relaxed, then ONE agent-scope acquire + wait, workgroup barrier, plain (vector) loads.  With
// plain stores instead the publishing lane needs an agent-scope release fence (buffer_wbl2) in front of the add: correct
// too, but each one writes back the XCD's whole L2 and costs the workgroup ~8 us -- 2.4 ms per 256 transforms against
// 1.3 ms (tools/lab_fft1m.hip, profiles/r02_fft1m_lab.md).
struct fused_args {
    float2 *data;       // count x 2^20, in place
    float2 *ws;         // queues x ring x 2^20
    const float2 *tw_1024;
    unsigned *sync;
    unsigned *sticky;   // nullable: set to 1 by a launch that gives up; the host clears it once per API call, not per launch
    uint32_t count, ring, lag, queues;
    uint32_t flags;     // lab only: 1 = no release fence, 2 = no acquire fence, 4 = queue = XCD id; 8 = fault injection (tests):
                        // every hand-off wait gives up at once
    uint32_t sleep;     // s_sleep argument of the polls is fixed; this many extra sleeps per poll iteration
    float scale;
    unsigned long long spin_limit; // wall_clock64 ticks (100 MHz) a poll may take before it gives up
};
__host__ __device__ constexpr size_t fused_sync_words(uint32_t count, uint32_t queues) { return handoff::sync_words(count, queues); }
using handoff::ld_relaxed;
using handoff::poll_geq;

template <bool REV, int MODE, int LAYOUT>
__global__ __launch_bounds__(kThreads, 4) void sdsp_fft1m_fused(fused_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_fft1m_smem[];
    float *plane = reinterpret_cast<float *>(sdsp_fft1m_smem);
    float2 *w1k = reinterpret_cast<float2 *>(sdsp_fft1m_smem + kPlaneBytes);
    float2 *qtab = w1k + 1024;
    // mailbox: [0], [1] the item's ticket, alternating per iteration -- an iteration without work has no barrier between
    // the other waves' read of its ticket and lane 0's write of the next one, so the next one goes to the other word;
    // [2] go / abort of the item's wait
    unsigned *mail = reinterpret_cast<unsigned *>(sdsp_fft1m_smem + kMailOffset);

    // flags & 4 (lab): bind the workgroup to the queue of the XCD it runs on (HW_REG_XCC_ID, bits 3:0) -- a transform's two
    // passes then run on ONE XCD and its intermediate is produced and consumed behind one L2
    const uint32_t q = (a.flags & 4u) ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) % a.queues : blockIdx.x % a.queues;
    if (q >= a.count)
        return;
    const uint32_t n_q = (a.count - q + a.queues - 1) / a.queues; // transforms of this queue
    unsigned *const ticket_ctr = a.sync + 32 * q, *const abort_flag = a.sync + 32 * a.queues;
    unsigned *const done = a.sync + 32 * (a.queues + 1);
    const unsigned n_tickets = (n_q + a.lag) * (2 * kTiles);

    stage_w1k(w1k, a.tw_1024);
    unsigned next = 0;
    if (threadIdx.x == 0)
        next = __hip_atomic_fetch_add(ticket_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    for (unsigned it = 0;; it++) {
        // ---- this item's ticket (drawn one item ahead, so the atomic's latency hides behind the previous tile)
        if (threadIdx.x == 0)
            mail[it & 1] = next;
        __syncthreads(); // also: every wave has finished the previous item (plane / qtab are free)
        const unsigned ticket = mail[it & 1];
        if (ticket >= n_tickets)
            break;
        if (threadIdx.x == 0)
            next = __hip_atomic_fetch_add(ticket_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned step = ticket / (2 * kTiles), sub = ticket % (2 * kTiles);
        const bool first = sub < kTiles;
        const unsigned i = first ? step : step - a.lag; // wraps for the leading pass-2 slots: filtered below
        if (i >= n_q)
            continue; // ramp-up / ramp-down slot without work (uniform)
        const unsigned tile = first ? sub : sub - kTiles;
        const unsigned xf = q + i * a.queues;
        float2 *const ws_x = a.ws + (size_t)(q * a.ring + i % a.ring) * (1ull << 20);
        float2 *const data_x = a.data + (size_t)xf * (1ull << 20);

        // ---- wait for what this item depends on (ONE lane polls; ONE acquire for the workgroup)
        unsigned *wait_word = nullptr;
        if (!first)
            wait_word = done + 32 * xf; // the whole intermediate of this transform
        else if (i >= a.ring)
            wait_word = done + 32 * (xf - a.ring * a.queues) + 16; // the ring slot's previous tenant has been read
        if (wait_word) {
            if (threadIdx.x == 0) {
                bool ok;
                if (a.flags & 8u) { // injected fault: behave exactly like a poll whose bound expired
                    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (a.sticky)
                        __hip_atomic_store(a.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                } else {
                    ok = poll_geq(wait_word, kTiles, abort_flag, a.sticky, a.spin_limit, a.sleep);
                }
                if (!(a.flags & 2u)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                mail[2] = ok ? 1u : 0u;
            }
            __syncthreads();
            if (mail[2] == 0u)
                break; // aborted: drain (uniform)
        }

        if (first)
            cols_tile<REV, MODE, LAYOUT, true>(data_x, ws_x, tile, plane, w1k, qtab);
        else
            rows_tile<REV, MODE, LAYOUT, true>(ws_x, data_x, tile, plane, w1k,
                                               reinterpret_cast<const float2 *>(sdsp_fft1m_smem + kRowTwOffset), a.scale);

        // ---- publish: pass 1 hands its output to other workgroups; pass 2 frees the ring slot, and its reads
        // of the slot must be complete (they are: the values were consumed) before the slot's next tenant stores
        if (first) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave
            __syncthreads();
            if (threadIdx.x == 0) {
                // write-through (sc1) stores are at the fabric once drained: only plain stores need the L2 written back
                if (!(LAYOUT & WS_SC1_STORES) && !(a.flags & 1u)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_fetch_add(done + 32 * xf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            __syncthreads(); // every wave's loads of the slot have returned (their values fed the butterflies)
            if (threadIdx.x == 0)
                __hip_atomic_fetch_add(done + 32 * xf + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
} // namespace fft1m
} // namespace sdsp_hip

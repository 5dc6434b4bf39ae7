// handoff.h -- what the persistent, ticketed two-pass kernels (fft1m_kernels.h: N = 2^20 f32; fft_2pass.hip: the other
// two-pass sizes) share: the layout of their synchronisation words, the bounded poll and the write-through store of the
// producer side.  Protocol: cdna_hip_programming.md Guideline 16, form R1 -- the intermediate is stored WRITE-THROUGH (sc1),
// every storing wave drains its stores, workgroup barrier, ONE lane adds to the arrival counter (relaxed, agent scope); the
// consumer's ONE lane polls relaxed, then ONE agent-scope acquire + wait, workgroup barrier, plain vector loads.
#pragma once

#include <hip/hip_runtime.h>

namespace sdsp_hip
{
namespace handoff
{
// Synchronisation words (zeroed by the host before every launch), every hot word on a line of its own:
//   sync[32 q]                          ticket counter of queue q
//   sync[32 queues]                     abort flag (a bounded spin gave up: results invalid, every workgroup drains)
//   sync[32 (queues + 1) + 32 u]        pass-1 items of unit u that have published their output
//   sync[32 (queues + 1) + 32 u + 16]   pass-2 items of unit u that have finished reading it
// (a unit = what one ticket step covers: one transform of N = 2^20, a group of smaller ones)
__host__ __device__ constexpr size_t sync_words(uint32_t units, uint32_t queues) { return 32ull * (queues + 1) + 32ull * units; }

__device__ __forceinline__ unsigned ld_relaxed(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ONE lane waits until *word >= target (or the launch is aborted); returns false when it gave up
__device__ __forceinline__ bool poll_geq(unsigned *word, unsigned target, unsigned *abort_flag, unsigned *sticky,
                                         unsigned long long limit, uint32_t extra_sleep)
{
    if (ld_relaxed(word) >= target)
        return true;
    const unsigned long long t0 = wall_clock64();
    for (unsigned it = 0;; it++) {
        __builtin_amdgcn_s_sleep(8);
        for (uint32_t i = 0; i < extra_sleep; i++)
            __builtin_amdgcn_s_sleep(8);
        if (ld_relaxed(word) >= target)
            return true;
        if ((it & 31) == 31) { // the give-up checks are rare: they must not add traffic to the hot words
            if (ld_relaxed(abort_flag) != 0)
                return false;
            if (wall_clock64() - t0 > limit) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (sticky)
                    __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// write-through (sc1) stores of the intermediate: straight to the fabric, the line is not kept dirty in the XCD's L2, so the
// publishing lane needs no release fence (which would write back the whole L2)
__device__ __forceinline__ void wt_store(float2 *p, float2 v)
{
    unsigned long long bits;
    __builtin_memcpy(&bits, &v, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16 bytes: no 128-bit atomic store exists, so the instruction is spelled out.  The s_nop covers the hazard the compiler cannot
// see inside the asm (a store of more than 64 bits followed by a write to its data registers, profiles/r03_store_hazard.md)
__device__ __forceinline__ void wt_store(double2 *p, double2 v)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 bits;
    __builtin_memcpy(&bits, &v, 16);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(bits) : "memory");
}
} // namespace handoff
} // namespace sdsp_hip

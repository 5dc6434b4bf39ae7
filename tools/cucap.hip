// cucap.hip -- how fast can ONE workgroup per CU move a 256 KiB chunk when the other CUs are busy with something else?
// Persistent grid of G workgroups x 1024 threads (128 KiB of dynamic LDS: one per CU).  Each walks chunks
// c = blockIdx.x, blockIdx.x + G, ...: 32 non-temporal loads of 8 B per lane (512 B per wave instruction, rows 8 KiB apart,
// the N = 32768 kernel's shape), a busy wait of `delay` ticks of 10 ns (the "compute phase"), 32 stores.
// Optional staggered start.  Prints chip-wide GB/s and the memory phase's length per chunk.
//   hipcc --offload-arch=gfx950 -O3 tools/cucap.hip -o build/cucap && build/cucap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int THREADS, int PTS>
__global__ __launch_bounds__(THREADS) void walk(v2f *p, unsigned chunks, unsigned delay, unsigned stag_n, unsigned stag_ticks,
                                                unsigned probe, unsigned stag_shift)
{
    extern __shared__ float pad[];
    if (probe == 0xffffffffu) p[0].x = pad[threadIdx.x];
    if (stag_n) {
        const unsigned long long t0 = wall_clock64();
        const unsigned long long want = (unsigned long long)((blockIdx.x >> stag_shift) % stag_n) * stag_ticks;
        while (wall_clock64() - t0 < want) __builtin_amdgcn_s_sleep(16);
    }
    for (unsigned c = blockIdx.x; c < chunks; c += gridDim.x) {
        v2f *base = p + (size_t)c * THREADS * PTS + threadIdx.x;
        v2f x[PTS];
#pragma unroll
        for (int k = 0; k < PTS; k++)
            x[k] = __builtin_nontemporal_load(base + THREADS * k);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PTS; k++) s += x[k].x;
        if (delay) {
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(8);
        }
        if (s == 123.456f) x[0].y = s;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PTS; k++) {
            v2f o = x[k];
            o.x *= 1.0001f;
            __builtin_nontemporal_store(o, base + THREADS * ((k * 5 + 3) % PTS));
        }
    }
}

template <int THREADS, int PTS> void run(v2f *d, size_t bytes, int grid, unsigned delay, unsigned stag_n, unsigned stag_ticks, size_t lds, unsigned stag_shift = 3)
{
    auto k = walk<THREADS, PTS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned chunks = (unsigned)(bytes / (sizeof(v2f) * THREADS * PTS));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, d, chunks, delay, stag_n, stag_ticks, 0u, stag_shift);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, d, chunks, delay, stag_n, stag_ticks, 0u, stag_shift);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double per_chunk_us = ms * 1e3 / ((double)chunks / grid);
    printf("threads %4d pts %2d (chunk %3zu KiB) grid %4d delay %5.1f us stagger %u x %4.1f us >>%u: %7.3f ms  %6.0f GB/s (%4.1f %%)  cycle %6.2f us, memory phase %6.2f us -> %5.1f GB/s per active CU\n",
           THREADS, PTS, sizeof(v2f) * THREADS * PTS / 1024, grid, delay / 100.0, stag_n, stag_ticks / 100.0, stag_shift, ms, 2.0 * bytes / ms / 1e6,
           2.0 * bytes / ms / 1e6 / 80.0, per_chunk_us, per_chunk_us - delay / 100.0,
           2.0 * sizeof(v2f) * THREADS * PTS / ((per_chunk_us - delay / 100.0) * 1e3));
    fflush(stdout);
}

int main()
{
    const size_t bytes = 4ull << 30;
    v2f *d;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 0x3c, bytes));
    const size_t one_per_cu = 128 * 1024, two_per_cu = 64 * 1024, four_per_cu = 36 * 1024;
    for (unsigned delay : { 800u, 1600u }) {
        run<512, 32>(d, bytes, 512, delay, 0, 0, two_per_cu);
        for (unsigned ticks : { 700u, 1350u, 2000u })
            run<512, 32>(d, bytes, 512, delay, 2, ticks, two_per_cu, 8);
        for (unsigned sn : { 2u, 4u, 8u })
            run<512, 32>(d, bytes, 512, delay, sn, 2700 / sn, two_per_cu, 3);
        for (unsigned sn : { 4u, 8u })
            run<512, 32>(d, bytes, 512, delay, sn, 2700 / sn, two_per_cu, 6);
    }
    for (unsigned delay : { 800u, 1600u }) {
        run<256, 16>(d, bytes, 1024, delay, 0, 0, four_per_cu);
        run<256, 16>(d, bytes, 1024, delay, 4, 400, four_per_cu, 8);
        run<256, 16>(d, bytes, 2048, delay, 0, 0, 18 * 1024);
        run<256, 16>(d, bytes, 2048, delay, 8, 200, 18 * 1024, 8);
    }
    return 0;
}

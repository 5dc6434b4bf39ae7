// delaybench.hip -- what does the time between a workgroup's loads and its stores cost?
// One workgroup per contiguous chunk (THREADS x PTS complex f32 = THREADS*PTS*8 bytes), non-temporal
// loads of the whole chunk, then a busy wait of `delay` clock ticks (s_memtime, 100 MHz), then non-temporal
// stores of the same chunk (permuted rows, like an FFT's reversed output).  In place, 2 GiB.
//   hipcc --offload-arch=gfx950 -O3 tools/delaybench.hip -o build/delaybench && build/delaybench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int THREADS, int PTS>
__global__ __launch_bounds__(THREADS) void chunk_copy(v2f *p, unsigned delay_ticks, unsigned lds_pad_probe)
{
    extern __shared__ float pad[]; // dynamic LDS only limits the workgroups per CU
    if (lds_pad_probe == 0xffffffffu) p[0].x = pad[threadIdx.x];
    v2f *base = p + (size_t)blockIdx.x * THREADS * PTS + threadIdx.x;
    v2f x[PTS];
#pragma unroll
    for (int k = 0; k < PTS; k++)
        x[k] = __builtin_nontemporal_load(base + THREADS * k);
    if (delay_ticks) {
        // make the wait depend on the loaded data having arrived
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PTS; k++) s += x[k].x;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < delay_ticks) { __builtin_amdgcn_s_sleep(8); }
        if (s == 123.456f) x[0].y = s;
    }
#pragma unroll
    for (int k = 0; k < PTS; k++)
        __builtin_nontemporal_store(x[k], base + THREADS * ((k * 5 + 3) % PTS)); // any fixed permutation of the rows
}

// MODE 0: progressive -- store k is issued as soon as load k has landed (the shape that reaches HBM peak)
// MODE 1: the workgroup walks CH consecutive chunks; all loads of a chunk, then all its stores
// MODE 2: as 1, but the stores of chunk c are interleaved one by one with the loads of chunk c+1
template <int THREADS, int PTS, int MODE, int CH>
__global__ __launch_bounds__(THREADS) void chunk_walk(v2f *p, unsigned lds_pad_probe)
{
    extern __shared__ float pad[];
    if (lds_pad_probe == 0xffffffffu) p[0].x = pad[threadIdx.x];
    v2f *base = p + (size_t)blockIdx.x * THREADS * PTS * CH + threadIdx.x;
    v2f x[PTS], y[PTS];
    if constexpr (MODE == 0) {
        for (int c = 0; c < CH; c++) {
            v2f *b = base + (size_t)c * THREADS * PTS;
#pragma unroll
            for (int k = 0; k < PTS; k++)
                x[k] = __builtin_nontemporal_load(b + THREADS * k);
#pragma unroll
            for (int k = 0; k < PTS; k++)
                __builtin_nontemporal_store(x[k], b + THREADS * ((k * 5 + 3) % PTS));
        }
    } else {
#pragma unroll
        for (int k = 0; k < PTS; k++)
            x[k] = __builtin_nontemporal_load(base + THREADS * k);
        for (int c = 0; c < CH; c++) {
            v2f *b = base + (size_t)c * THREADS * PTS;
            // everything of this chunk has to be here before anything of it may leave (a transform's constraint)
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < PTS; k++) s += x[k].x;
            asm volatile("" : "+v"(s));
            if (s == 123.456f) x[0].y = s;
            if (MODE == 2 && c + 1 < CH) {
#pragma unroll
                for (int k = 0; k < PTS; k++) {
                    __builtin_nontemporal_store(x[k], b + THREADS * ((k * 5 + 3) % PTS));
                    y[k] = __builtin_nontemporal_load(b + THREADS * PTS + THREADS * k);
                }
            } else {
#pragma unroll
                for (int k = 0; k < PTS; k++)
                    __builtin_nontemporal_store(x[k], b + THREADS * ((k * 5 + 3) % PTS));
                if (c + 1 < CH) {
#pragma unroll
                    for (int k = 0; k < PTS; k++)
                        y[k] = __builtin_nontemporal_load(b + THREADS * PTS + THREADS * k);
                }
            }
#pragma unroll
            for (int k = 0; k < PTS; k++) x[k] = y[k];
        }
    }
}

// the bare load/store pattern of sdsp_fft_big_kernel exactly as it was measured from Python (uniform base +
// 32-bit per-thread offset addressing, rows written in bit-reversed order)
__device__ __forceinline__ v2f *at(v2f *base, unsigned byte_off) { return (v2f *)((char *)base + byte_off); }
// PERM: where row i's data goes: 0 bit-reversed row (8 of 32 rows keep their place, the rest swap in pairs),
// 1 the same row (every store rewrites what is already there), 2 row i^1 (no row keeps its place),
// 3 row (5 i + 3) mod 32 (no fixed point, not an involution); SCALE: multiply by a constant so that the stored
// bits differ from the loaded ones even where the row keeps its place
template <int L, int PERM = 0, bool SCALE = false>
__global__ __launch_bounds__((1 << L) / 32, 4) void big_shape(v2f *data)
{
    constexpr unsigned N = 1u << L, T = N / 32;
    extern __shared__ float pad[];
    const unsigned t = threadIdx.x;
    v2f *base = data + (size_t)blockIdx.x * N;
    const unsigned toff = t * 8u;
    v2f x[32];
#pragma unroll
    for (int k = 0; k < 32; k++)
        x[k] = __builtin_nontemporal_load(at(base + T * k, toff));
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const int row = PERM == 0 ? (int)(__brev((unsigned)i) >> 27) : PERM == 1 ? i : PERM == 2 ? (i ^ 1) : (5 * i + 3) % 32;
        v2f v = x[i];
        if (SCALE) v *= 1.0000001f;
        __builtin_nontemporal_store(v, at(base + T * row, toff));
    }
}

template <typename F> double time_ms(F launch)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 25; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 25; i++) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 25;
}

int main()
{
    const size_t bytes = 2ull << 30;
    void *d; CK(hipMalloc(&d, bytes));
    // random finite data (all-equal or NaN data moves fewer bits and runs faster)
    {
        unsigned *h = (unsigned *)malloc(bytes);
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < bytes / 4; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = 0x3f000000u | (unsigned)(s & 0x7fffff); }
        CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); free(h);
    }
    unsigned long long t0 = 0;
    printf("%-28s %10s %10s %10s %10s %10s\n", "chunk per workgroup", "delay 0", "1 us", "2 us", "4 us", "8 us");
    auto row = [&](const char *name, auto kernel, int threads, size_t chunk, size_t lds) {
        printf("%-28s", name);
        for (unsigned us : { 0u, 1u, 2u, 4u, 8u }) {
            const unsigned ticks = us * 100; // s_memtime / readcyclecounter: 100 MHz constant clock
            const double ms = time_ms([&] { hipLaunchKernelGGL(kernel, dim3((unsigned)(bytes / chunk)), dim3(threads), lds, 0, (v2f *)d, ticks, 0u); });
            printf(" %9.1f%%", 2.0 * bytes / ms / 1e6 / 80.0);
        }
        printf("\n");
    };
    (void)t0;
    row("32 KiB (256 thr x 16), 4/CU", chunk_copy<256, 16>, 256, 32768, 40 * 1024 - 1024);
    row("32 KiB (128 thr x 32), 8/CU", chunk_copy<128, 32>, 128, 32768, 20 * 1024 - 512);
    row("64 KiB (256 thr x 32), 4/CU", chunk_copy<256, 32>, 256, 65536, 40 * 1024 - 1024);
    row("64 KiB (512 thr x 16), 4/CU", chunk_copy<512, 16>, 512, 65536, 40 * 1024 - 1024);
    row("128 KiB (512 thr x 32), 2/CU", chunk_copy<512, 32>, 512, 131072, 64 * 1024 - 1024);
    row("64 KiB (256 thr x 32), 2/CU", chunk_copy<256, 32>, 256, 65536, 64 * 1024 - 1024);
    printf("\nwalks of CH chunks per workgroup (256 thr x 16 = 32 KiB chunks, <= 4 workgroups per CU)\n");
    auto walk = [&](const char *name, auto kernel, int ch) {
        const double ms = time_ms([&] { hipLaunchKernelGGL(kernel, dim3((unsigned)(bytes / 32768 / ch)), dim3(256), 40 * 1024 - 1024, 0, (v2f *)d, 0u); });
        printf("%-60s %6.1f%%\n", name, 2.0 * bytes / ms / 1e6 / 80.0);
    };
    walk("progressive (store k as load k lands), 1 chunk", chunk_walk<256, 16, 0, 1>, 1);
    walk("progressive, 4 chunks", chunk_walk<256, 16, 0, 4>, 4);
    walk("all loads then all stores, 1 chunk", chunk_walk<256, 16, 1, 1>, 1);
    walk("all loads then all stores, 4 chunks", chunk_walk<256, 16, 1, 4>, 4);
    walk("all loads then all stores, 16 chunks", chunk_walk<256, 16, 1, 16>, 16);
    walk("stores of c interleaved with loads of c+1, 4 chunks", chunk_walk<256, 16, 2, 4>, 4);
    walk("stores of c interleaved with loads of c+1, 16 chunks", chunk_walk<256, 16, 2, 16>, 16);
    walk("stores of c interleaved with loads of c+1, 64 chunks", chunk_walk<256, 16, 2, 64>, 64);
    printf("\nprogressive copies, one chunk per workgroup\n");
    auto prog = [&](const char *name, auto kernel, int threads, size_t chunk, size_t lds) {
        const double ms = time_ms([&] { hipLaunchKernelGGL(kernel, dim3((unsigned)(bytes / chunk)), dim3(threads), lds, 0, (v2f *)d, 0u); });
        printf("%-60s %6.1f%%\n", name, 2.0 * bytes / ms / 1e6 / 80.0);
    };
    prog("32 KiB (128 thr x 32), 16 KiB LDS", chunk_walk<128, 32, 0, 1>, 128, 32768, 16 * 1024);
    prog("64 KiB (256 thr x 32), 32 KiB LDS", chunk_walk<256, 32, 0, 1>, 256, 65536, 32 * 1024);
    prog("64 KiB (256 thr x 32), no LDS", chunk_walk<256, 32, 0, 1>, 256, 65536, 0);
    prog("64 KiB (512 thr x 16), 32 KiB LDS", chunk_walk<512, 16, 0, 1>, 512, 65536, 32 * 1024);
    prog("128 KiB (512 thr x 32), 64 KiB LDS", chunk_walk<512, 32, 0, 1>, 512, 131072, 64 * 1024 - 1024);
    prog("64 KiB all loads then all stores (256 thr x 32), 32 KiB LDS", chunk_walk<256, 32, 1, 1>, 256, 65536, 32 * 1024);
    printf("\nsdsp_fft_big_kernel's bare pattern\n");
    for (int rep = 0; rep < 2; rep++) {
        double ms = time_ms([&] { hipLaunchKernelGGL(big_shape<12>, dim3((unsigned)(bytes / 32768)), dim3(128), 16384, 0, (v2f *)d); });
        printf("N = 4096  (128 thr x 32, 32 KiB chunk)   %6.1f%%\n", 2.0 * bytes / ms / 1e6 / 80.0);
        ms = time_ms([&] { hipLaunchKernelGGL(big_shape<13>, dim3((unsigned)(bytes / 65536)), dim3(256), 32768, 0, (v2f *)d); });
        printf("N = 8192  (256 thr x 32, 64 KiB chunk)   %6.1f%%\n", 2.0 * bytes / ms / 1e6 / 80.0);
        ms = time_ms([&] { hipLaunchKernelGGL(big_shape<14>, dim3((unsigned)(bytes / 131072)), dim3(512), 65536 - 1024, 0, (v2f *)d); });
        printf("N = 16384 (512 thr x 32, 128 KiB chunk)  %6.1f%%\n", 2.0 * bytes / ms / 1e6 / 80.0);
    }
    printf("\nN = 8192 bare pattern (64 KiB chunks): where the rows go, and whether the stored bits change\n");
    auto shape = [&](const char *name, auto kernel) {
        const double ms = time_ms([&] { hipLaunchKernelGGL(kernel, dim3((unsigned)(bytes / 65536)), dim3(256), 32768, 0, (v2f *)d); });
        printf("%-64s %6.1f%%\n", name, 2.0 * bytes / ms / 1e6 / 80.0);
    };
    shape("bit-reversed rows (8 of 32 fixed), same bits", big_shape<13, 0, false>);
    shape("same row (all fixed), same bits", big_shape<13, 1, false>);
    shape("row ^ 1 (none fixed), same bits", big_shape<13, 2, false>);
    shape("row (5i+3) mod 32 (none fixed), same bits", big_shape<13, 3, false>);
    shape("bit-reversed rows, scaled by 1.0000001", big_shape<13, 0, true>);
    shape("same row, scaled by 1.0000001", big_shape<13, 1, true>);
    shape("row ^ 1, scaled by 1.0000001", big_shape<13, 2, true>);
    CK(hipFree(d));
    return 0;
}

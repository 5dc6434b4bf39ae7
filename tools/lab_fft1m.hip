// lab_fft1m.hip -- measurement harness for the N = 2^20 path (not part of the product library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Isimpledsp_amd/csrc tools/lab_fft1m.hip -o build/lab_fft1m
//   build/lab_fft1m [batch = 256]
// Builds the product's tile code (fft1m_kernels.h) in three modes -- the real transform, the same loads/stores without
// butterflies, and the HBM-facing halves alone -- under both schedules (two launches per chunk / one persistent launch)
// and both intermediate layouts, times each in steady state (forward / reverse alternating so the data stays finite),
// and checks the persistent schedule bit for bit against the two-launch one.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft1m_kernels.h"

using namespace sdsp_hip::fft1m;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            std::exit(2);                                                                  \
        }                                                                                  \
    } while (0)

static float2 *g_data, *g_ref, *g_ws, *g_tw;
static unsigned *g_sync;
static uint32_t g_batch = 256;
static int g_cus = 256;

__global__ void fill_kernel(float2 *p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float a = (float)(h & 0xffff) / 65536.0f - 0.5f, b = (float)(h >> 16) / 65536.0f - 0.5f;
        p[i] = float2{ a, b };
    }
}

template <typename K> static void set_lds(K kern)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
}

template <bool REV, int MODE, int LAYOUT> static void two_launch(float2 *data, uint32_t batch, uint32_t chunk)
{
    set_lds(sdsp_fft1m_cols<REV, MODE, LAYOUT>);
    set_lds(sdsp_fft1m_rows<REV, MODE, LAYOUT>);
    for (uint32_t done = 0; done < batch; done += chunk) {
        const uint32_t n = std::min(chunk, batch - done);
        float2 *d = data + (size_t)done * (1u << 20);
        hipLaunchKernelGGL((sdsp_fft1m_cols<REV, MODE, LAYOUT>), dim3(n * kTiles), dim3(kThreads), kLdsBytes, 0, d, g_ws, g_tw);
        hipLaunchKernelGGL((sdsp_fft1m_rows<REV, MODE, LAYOUT>), dim3(n * kTiles), dim3(kThreads), kLdsBytes, 0, g_ws, d, g_tw,
                           1.0f / 1048576.0f);
    }
}

template <bool REV, int MODE, int LAYOUT> static void fused(float2 *data, uint32_t batch, uint32_t ring, uint32_t lag, uint32_t per_cu)
{
    set_lds(sdsp_fft1m_fused<REV, MODE, LAYOUT>);
    CK(hipMemsetAsync(g_sync, 0, ((4 + 2 * (size_t)batch) * 4 + 15) & ~(size_t)15, 0));
    fused_args a;
    a.data = data;
    a.ws = g_ws;
    a.tw_1024 = g_tw;
    a.sync = g_sync;
    a.count = batch;
    a.ring = ring;
    a.lag = lag;
    a.scale = 1.0f / 1048576.0f;
    a.spin_limit = 100000000ull; // 1 s
    hipLaunchKernelGGL((sdsp_fft1m_fused<REV, MODE, LAYOUT>), dim3(per_cu * g_cus), dim3(kThreads), kLdsBytes, 0, a);
}

static unsigned read_abort()
{
    unsigned w[4];
    CK(hipMemcpy(w, g_sync, sizeof(w), hipMemcpyDeviceToHost));
    return w[1];
}

// time `pairs` forward+reverse pairs of `run(rev)`; returns ms per call
template <typename F> static double time_it(F run, int warm = 2, int pairs = 5)
{
    for (int i = 0; i < warm; i++) {
        run(false);
        run(true);
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < pairs; i++) {
        run(false);
        run(true);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / (2.0 * pairs);
}

static void report(const char *what, double ms)
{
    const double gbs = (double)g_batch * 16.0 * 1048576.0 / (ms * 1e-3) / 1e9;
    std::printf("%-72s %8.3f ms  %7.0f GB/s compulsory  %5.1f %%\n", what, ms, gbs, gbs / 80.0);
    std::fflush(stdout);
}

template <int MODE, int LAYOUT> static void sweep(const char *mode_name, const char *layout_name)
{
    char buf[160];
    for (uint32_t chunk : { 32u, 16u, 8u }) {
        std::snprintf(buf, sizeof buf, "%s %s two launches per chunk of %u", mode_name, layout_name, chunk);
        report(buf, time_it([&](bool rev) {
                   if (rev)
                       two_launch<true, MODE, LAYOUT>(g_data, g_batch, chunk);
                   else
                       two_launch<false, MODE, LAYOUT>(g_data, g_batch, chunk);
               }));
    }
    const uint32_t rl[][2] = { { 3, 2 }, { 4, 2 }, { 6, 4 }, { 8, 6 }, { 8, 4 }, { 12, 8 }, { 16, 12 }, { 32, 24 } };
    for (auto &p : rl) {
        for (uint32_t per_cu : { 2u }) {
            std::snprintf(buf, sizeof buf, "%s %s persistent ring %u lag %u, %u wg/CU", mode_name, layout_name, p[0], p[1], per_cu);
            report(buf, time_it([&](bool rev) {
                       if (rev)
                           fused<true, MODE, LAYOUT>(g_data, g_batch, p[0], p[1], per_cu);
                       else
                           fused<false, MODE, LAYOUT>(g_data, g_batch, p[0], p[1], per_cu);
                   }));
            if (read_abort())
                std::printf("   ^^^ ABORTED (a bounded spin gave up)\n");
        }
    }
}

int main(int argc, char **argv)
{
    if (argc > 1)
        g_batch = (uint32_t)std::atoi(argv[1]);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    std::printf("device %s, %d CUs; batch %u x 2^20\n", prop.name, g_cus, g_batch);
    const size_t n = (size_t)g_batch << 20;
    CK(hipMalloc(&g_data, n * 8));
    CK(hipMalloc(&g_ref, n * 8));
    CK(hipMalloc(&g_ws, (size_t)32 << 23));
    CK(hipMalloc(&g_tw, 1024 * 8));
    CK(hipMalloc(&g_sync, (4 + 2 * (size_t)g_batch) * 4 + 16));
    std::vector<float2> tw(1024);
    for (int j = 0; j < 1024; j++) {
        const double a = -2.0 * M_PI * j / 1024.0;
        tw[j] = float2{ (float)std::cos(a), (float)std::sin(a) };
    }
    CK(hipMemcpy(g_tw, tw.data(), 1024 * 8, hipMemcpyHostToDevice));

    // ---- correctness of the persistent schedule: bit-identical to the two-launch schedule, several ring shapes,
    // repeated (a lost or early hand-off shows as a mismatch)
    int bad_total = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, g_ref, n, 1234u + rep);
        two_launch<false, MODE_FFT, WS_ROWS>(g_ref, g_batch, 32);
        CK(hipDeviceSynchronize());
        const uint32_t rl[][2] = { { 1, 0 }, { 2, 1 }, { 4, 2 }, { 8, 6 }, { 32, 24 } };
        for (auto &p : rl)
            for (int layout = 0; layout < 2; layout++) {
                hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, g_data, n, 1234u + rep);
                if (layout)
                    fused<false, MODE_FFT, WS_BLOCKED>(g_data, g_batch, p[0], p[1], 2);
                else
                    fused<false, MODE_FFT, WS_ROWS>(g_data, g_batch, p[0], p[1], 2);
                CK(hipDeviceSynchronize());
                const unsigned ab = read_abort();
                // compare on the host in slices
                size_t bad = 0;
                std::vector<float2> ha(1 << 20), hb(1 << 20);
                for (uint32_t x = 0; x < g_batch; x += (g_batch > 16 ? g_batch / 16 : 1)) {
                    CK(hipMemcpy(ha.data(), g_data + ((size_t)x << 20), 8u << 20, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hb.data(), g_ref + ((size_t)x << 20), 8u << 20, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < ha.size(); i++)
                        bad += (ha[i].x != hb[i].x) || (ha[i].y != hb[i].y);
                }
                std::printf("check rep %d ring %2u lag %2u layout %d: %zu mismatching elements%s\n", rep, p[0], p[1], layout, bad,
                            ab ? "  ABORTED" : "");
                bad_total += bad != 0 || ab;
            }
    }
    std::printf("correctness: %s\n", bad_total ? "FAILED" : "ok");
    std::fflush(stdout);

    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, g_data, n, 99u);
    sweep<MODE_FFT, WS_ROWS>("fft ", "rows   ");
    sweep<MODE_FFT, WS_BLOCKED>("fft ", "blocked");
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, g_data, n, 99u);
    sweep<MODE_MOVE, WS_ROWS>("move", "rows   ");
    sweep<MODE_MOVE, WS_BLOCKED>("move", "blocked");
    sweep<MODE_HBM_ONLY, WS_ROWS>("hbm-only", "");
    return bad_total ? 1 : 0;
}

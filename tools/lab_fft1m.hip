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
#include <string>
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

// per transform: number of mismatching elements and the first mismatching index
__global__ void compare_kernel(const uint2 *a, const uint2 *b, unsigned *bad, unsigned *first)
{
    const size_t base = (size_t)blockIdx.x << 20;
    unsigned n = 0, f = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i < (1u << 20); i += blockDim.x) {
        const uint2 x = a[base + i], y = b[base + i];
        if (x.x != y.x || x.y != y.y) {
            n++;
            f = f < i ? f : i;
        }
    }
    if (n) {
        atomicAdd(bad + blockIdx.x, n);
        atomicMin(first + blockIdx.x, f);
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

static uint32_t g_flags = 0, g_sleep = 0, g_last_queues = 1;
template <bool REV, int MODE, int LAYOUT>
static void fused(float2 *data, uint32_t batch, uint32_t ring, uint32_t lag, uint32_t per_cu, uint32_t queues = 1)
{
    set_lds(sdsp_fft1m_fused<REV, MODE, LAYOUT>);
    CK(hipMemsetAsync(g_sync, 0, fused_sync_words(batch, queues) * 4, 0));
    fused_args a;
    a.data = data;
    a.ws = g_ws;
    a.tw_1024 = g_tw;
    a.sync = g_sync;
    a.count = batch;
    a.ring = ring;
    a.lag = lag;
    a.queues = queues;
    g_last_queues = queues;
    a.flags = g_flags;
    a.sleep = g_sleep;
    a.scale = 1.0f / 1048576.0f;
    a.sticky = nullptr;
    a.spin_limit = 100000000ull; // 1 s
    hipLaunchKernelGGL((sdsp_fft1m_fused<REV, MODE, LAYOUT>), dim3(per_cu * g_cus), dim3(kThreads), kLdsBytes, 0, a);
}

// software-pipelined launches: launch i = pass 1 of chunk i + pass 2 of chunk i - 1 (double-buffered intermediate)
template <bool REV, int MODE, int LAYOUT> static void mixed(float2 *data, uint32_t batch, uint32_t chunk)
{
    set_lds(sdsp_fft1m_mixed<REV, MODE, LAYOUT>);
    const uint32_t n_chunks = (batch + chunk - 1) / chunk;
    for (uint32_t i = 0; i <= n_chunks; i++) {
        const uint32_t n1 = i < n_chunks ? std::min(chunk, batch - i * chunk) : 0;
        const uint32_t n2 = i > 0 ? std::min(chunk, batch - (i - 1) * chunk) : 0;
        float2 *ws1 = g_ws + (size_t)(i & 1) * chunk * (1u << 20);
        float2 *ws2 = g_ws + (size_t)((i + 1) & 1) * chunk * (1u << 20);
        float2 *d1 = data + (size_t)i * chunk * (1u << 20);
        float2 *d2 = i > 0 ? data + (size_t)(i - 1) * chunk * (1u << 20) : data;
        hipLaunchKernelGGL((sdsp_fft1m_mixed<REV, MODE, LAYOUT>), dim3((n1 + n2) * kTiles), dim3(kThreads), kLdsBytes, 0, d1, ws1, n1,
                           ws2, d2, n2, g_tw, 1.0f / 1048576.0f);
    }
}

static unsigned read_abort()
{
    unsigned w[4];
    (void)w;
    unsigned v = 0;
    CK(hipMemcpy(&v, g_sync + 32 * g_last_queues, 4, hipMemcpyDeviceToHost));
    return v;
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

struct shape {
    uint32_t queues, ring, lag, per_cu;
};
template <int MODE, int LAYOUT> static void sweep(const char *mode_name, const char *layout_name, bool quick)
{
    char buf[200];
    for (uint32_t chunk : { 32u }) {
        std::snprintf(buf, sizeof buf, "%s %s two launches per chunk of %u", mode_name, layout_name, chunk);
        report(buf, time_it([&](bool rev) {
                   if (rev)
                       two_launch<true, MODE, LAYOUT>(g_data, g_batch, chunk);
                   else
                       two_launch<false, MODE, LAYOUT>(g_data, g_batch, chunk);
               }));
    }
    for (uint32_t chunk : { 4u, 8u, 12u, 16u }) {
        std::snprintf(buf, sizeof buf, "%s %s pipelined launches, chunk %u", mode_name, layout_name, chunk);
        report(buf, time_it([&](bool rev) {
                   if (rev)
                       mixed<true, MODE, LAYOUT>(g_data, g_batch, chunk);
                   else
                       mixed<false, MODE, LAYOUT>(g_data, g_batch, chunk);
               }));
    }
    const shape shapes[] = { { 1, 8, 4, 2 },  { 1, 16, 12, 2 }, { 1, 32, 24, 2 }, { 8, 2, 1, 2 }, { 8, 3, 1, 2 },
                             { 8, 4, 2, 2 },  { 16, 2, 1, 2 },  { 4, 4, 2, 2 },   { 4, 8, 4, 2 }, { 2, 8, 4, 2 } };
    for (uint32_t flags : { 0u, 3u }) {
        if (flags && MODE == MODE_FFT)
            continue; // the no-fence runs are timing experiments on the data movers only
        for (auto &sh : shapes) {
            if (quick && !(sh.queues == 8 && sh.ring == 3) && !(sh.queues == 1 && sh.ring == 32))
                continue;
            g_flags = flags;
            std::snprintf(buf, sizeof buf, "%s %s persistent queues %u ring %u lag %u%s", mode_name, layout_name, sh.queues, sh.ring,
                          sh.lag, flags ? " NO FENCES" : "");
            report(buf, time_it([&](bool rev) {
                       if (rev)
                           fused<true, MODE, LAYOUT>(g_data, g_batch, sh.ring, sh.lag, sh.per_cu, sh.queues);
                       else
                           fused<false, MODE, LAYOUT>(g_data, g_batch, sh.ring, sh.lag, sh.per_cu, sh.queues);
                   }));
            if (read_abort())
                std::printf("   ^^^ ABORTED (a bounded spin gave up)\n");
        }
    }
    g_flags = 0;
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
    const bool ring_mode = argc > 2 && std::string(argv[2]) == "ring";
    CK(hipMalloc(&g_ws, (size_t)(ring_mode ? 128 : 32) << 23));
    CK(hipMalloc(&g_tw, 1024 * 8));
    CK(hipMalloc(&g_sync, fused_sync_words(g_batch, 16) * 4));
    std::vector<float2> tw(1024);
    for (int j = 0; j < 1024; j++) {
        const double a = -2.0 * M_PI * j / 1024.0;
        tw[j] = float2{ (float)std::cos(a), (float)std::sin(a) };
    }
    CK(hipMemcpy(g_tw, tw.data(), 1024 * 8, hipMemcpyHostToDevice));

    if (ring_mode) {
        // round 2 verdict, next #9: where does the intermediate live?  The same persistent launch (sc1 stores, acquire + plain
        // loads: the product's) with the intermediate ring sized from 8 to 128 transforms = 64 MiB .. 1 GiB, i.e. from well
        // inside the 256 MiB Infinity Cache to four times its size; per-transform time against ring bytes shows the knee.
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, g_data, n, 99u);
        const shape shapes[] = { { 8, 1, 0, 2 }, { 8, 2, 1, 2 }, { 8, 3, 1, 2 }, { 8, 4, 2, 2 }, { 8, 6, 4, 2 }, { 8, 8, 6, 2 },
                                 { 8, 12, 10, 2 }, { 8, 16, 14, 2 }, { 8, 8, 2, 2 }, { 8, 16, 2, 2 }, { 16, 8, 6, 2 } };
        for (int rep = 0; rep < 2; rep++)
            for (auto &sh : shapes) {
                char buf[200];
                g_flags = 1; // sc1 stores carry no release fence
                const double ms = time_it([&](bool rev) {
                    if (rev)
                        fused<true, MODE_FFT, WS_BLOCKED | WS_SC1_STORES>(g_data, g_batch, sh.ring, sh.lag, sh.per_cu, sh.queues);
                    else
                        fused<false, MODE_FFT, WS_BLOCKED | WS_SC1_STORES>(g_data, g_batch, sh.ring, sh.lag, sh.per_cu, sh.queues);
                });
                std::snprintf(buf, sizeof buf, "ring: queues %2u x ring %2u (lag %2u) = %3u intermediates = %4u MiB, %.2f us per transform",
                              sh.queues, sh.ring, sh.lag, sh.queues * sh.ring, sh.queues * sh.ring * 8, ms * 1e3 / g_batch);
                report(buf, ms);
                if (read_abort())
                    std::printf("   ^^^ ABORTED\n");
            }
        g_flags = 0;
        return 0;
    }

    // ---- exhaustive checks of the persistent schedule: every element of every transform against the two-launch
    // schedule, launches back to back (warm caches, the ring re-used at once), batch sizes that leave queues uneven
    int bad_total = 0;
    unsigned *d_bad, *d_first;
    CK(hipMalloc(&d_bad, g_batch * 4));
    CK(hipMalloc(&d_first, g_batch * 4));
    std::vector<unsigned> h_bad(g_batch), h_first(g_batch);
    auto check_all = [&](const char *what, uint32_t batch) {
        CK(hipMemset(d_bad, 0, batch * 4));
        CK(hipMemset(d_first, 0xff, batch * 4));
        hipLaunchKernelGGL(compare_kernel, dim3(batch), dim3(1024), 0, 0, reinterpret_cast<const uint2 *>(g_data),
                           reinterpret_cast<const uint2 *>(g_ref), d_bad, d_first);
        CK(hipMemcpy(h_bad.data(), d_bad, batch * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_first.data(), d_first, batch * 4, hipMemcpyDeviceToHost));
        const unsigned ab = read_abort();
        unsigned n_x = 0;
        size_t total = 0;
        for (uint32_t x = 0; x < batch; x++)
            if (h_bad[x]) {
                if (n_x < 6)
                    std::printf("      transform %u: %u elements differ, first at %u (row %u col %u)\n", x, h_bad[x], h_first[x],
                                h_first[x] >> 10, h_first[x] & 1023);
                n_x++;
                total += h_bad[x];
            }
        std::printf("check %-64s %zu elements in %u of %u transforms%s\n", what, total, n_x, batch, ab ? "  ABORTED" : "");
        std::fflush(stdout);
        bad_total += n_x != 0 || ab;
    };
    struct cfg {
        uint32_t queues, ring, lag, flags;
        int layout;
        const char *name;
    };
    const cfg cfgs[] = {
        { 8, 3, 1, 0, WS_BLOCKED, "q8 r3 l1 plain stores + release, acquire + plain loads" },
        { 8, 3, 1, 1, WS_BLOCKED | WS_SC1_STORES, "q8 r3 l1 sc1 stores (no release), acquire + plain loads" },
        { 8, 2, 1, 1, WS_BLOCKED | WS_SC1_STORES, "q8 r2 l1 sc1 stores (no release), acquire + plain loads" },
        { 8, 4, 2, 1, WS_BLOCKED | WS_SC1_STORES, "q8 r4 l2 sc1 stores (no release), acquire + plain loads" },
        { 16, 2, 1, 1, WS_BLOCKED | WS_SC1_STORES, "q16 r2 l1 sc1 stores (no release), acquire + plain loads" },
        { 4, 4, 2, 1, WS_BLOCKED | WS_SC1_STORES, "q4 r4 l2 sc1 stores (no release), acquire + plain loads" },
        { 3, 3, 1, 1, WS_BLOCKED | WS_SC1_STORES, "q3 r3 l1 (queues across XCDs) sc1 stores, acquire + plain loads" },
        { 1, 1, 0, 1, WS_BLOCKED | WS_SC1_STORES, "q1 r1 l0 (slot re-used at once) sc1 stores, acquire + plain loads" },
        { 1, 2, 1, 1, WS_BLOCKED | WS_SC1_STORES, "q1 r2 l1 sc1 stores, acquire + plain loads" },
        { 1, 8, 4, 1, WS_BLOCKED | WS_SC1_STORES, "q1 r8 l4 sc1 stores, acquire + plain loads" },
        { 1, 32, 24, 0, WS_BLOCKED, "q1 r32 l24 plain stores + release, acquire + plain loads" },
        { 1, 2, 1, 0, WS_BLOCKED, "q1 r2 l1 plain stores + release, acquire + plain loads" },
    };
    auto run_cfg = [&](const cfg &c, bool rev, uint32_t batch) {
        g_flags = c.flags;
#define RUN_L(L)                                                              \
    if (rev)                                                                  \
        fused<true, MODE_FFT, L>(g_data, batch, c.ring, c.lag, 2, c.queues);  \
    else                                                                      \
        fused<false, MODE_FFT, L>(g_data, batch, c.ring, c.lag, 2, c.queues)
        switch (c.layout) {
        case WS_BLOCKED: RUN_L(WS_BLOCKED); break;
        default: RUN_L(WS_BLOCKED | WS_SC1_STORES); break;
        }
        g_flags = 0;
    };
    for (uint32_t batch : { 37u, g_batch }) {
        const size_t nb = (size_t)batch << 20;
        for (auto &c : cfgs) {
            char what[200];
            for (int rep = 0; rep < 2; rep++) {
                // reference: forward, reverse, forward through the two-launch schedule
                hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, g_ref, nb, 500u + rep);
                two_launch<false, MODE_FFT, WS_ROWS>(g_ref, batch, 32);
                two_launch<true, MODE_FFT, WS_ROWS>(g_ref, batch, 32);
                two_launch<false, MODE_FFT, WS_ROWS>(g_ref, batch, 32);
                // the same three transforms back to back through the persistent schedule
                hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, g_data, nb, 500u + rep);
                run_cfg(c, false, batch);
                run_cfg(c, true, batch);
                run_cfg(c, false, batch);
                CK(hipDeviceSynchronize());
                std::snprintf(what, sizeof what, "batch %u rep %d %s:", batch, rep, c.name);
                check_all(what, batch);
            }
        }
    }
    std::printf("correctness: %s\n", bad_total ? "FAILED" : "ok");
    std::fflush(stdout);
    // ---- timings
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, g_data, n, 99u);
    report("fft blocked two launches per chunk of 32", time_it([&](bool rev) {
               if (rev)
                   two_launch<true, MODE_FFT, WS_BLOCKED>(g_data, g_batch, 32);
               else
                   two_launch<false, MODE_FFT, WS_BLOCKED>(g_data, g_batch, 32);
           }));
    for (auto &c : cfgs) {
        char what[200];
        std::snprintf(what, sizeof what, "fft persistent q %u ring %u lag %u: %s", c.queues, c.ring, c.lag, c.name);
        report(what, time_it([&](bool rev) { run_cfg(c, rev, g_batch); }));
        if (read_abort())
            std::printf("   ^^^ ABORTED\n");
    }
    return bad_total ? 1 : 0;
}

// lab_fft4096.hip -- measurement harness for the N = 4096 kernels (not part of the product library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Isimpledsp_amd/csrc tools/lab_fft4096.hip -o build/lab_fft4096
//   build/lab_fft4096 [batch = 65536] [rounds = 2]
// Instantiates the radix-4 kernel of fft4096_kernels.h over its whole grid of scheduling knobs (barrier placement, store
// order, load order, LDS read order: identical arithmetic), checks that every instance produces the same bits, and times
// each in steady state (forward / reverse alternating, all instances interleaved, `rounds` times) beside the radix-2 kernel.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>

#include "fft4096_kernels.h"

using namespace sdsp_hip::fft4096;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            std::exit(2);                                                                  \
        }                                                                                  \
    } while (0)

typedef void (*kern_t)(float2 *, const float2 *, uint64_t, float);
struct inst {
    int bar, sord, lord, ldsb, waves;
    kern_t fwd, rev;
};
static std::vector<inst> g_inst;

template <int BAR, int SORD, int LORD, int LDSB, int WAVES> static void add()
{
    g_inst.push_back({ BAR, SORD, LORD, LDSB, WAVES, sdsp_fft4096_r4_f32<false, BAR, SORD, LORD, LDSB, WAVES>,
                       sdsp_fft4096_r4_f32<true, BAR, SORD, LORD, LDSB, WAVES> });
}
template <int BAR, int SORD, int LORD> static void add_ldsb()
{
    add<BAR, SORD, LORD, 0, 3>();
    add<BAR, SORD, LORD, 1, 3>();
}
template <int BAR, int SORD> static void add_lord()
{
    add_ldsb<BAR, SORD, 0>();
    add_ldsb<BAR, SORD, 1>();
    add_ldsb<BAR, SORD, 2>();
}
template <int BAR> static void add_sord()
{
    add_lord<BAR, 0>();
    add_lord<BAR, 1>();
    add_lord<BAR, 2>();
}

__global__ void fill_kernel(float2 *p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = float2{ (float)(h & 0xffff) / 65536.0f - 0.5f, (float)(h >> 16) / 65536.0f - 0.5f };
    }
}
__global__ void checksum_kernel(const uint2 *p, size_t n, unsigned long long *out)
{
    unsigned long long acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += (unsigned long long)p[i].x * 31u + (unsigned long long)p[i].y * 17u + (i & 0xffff) * (unsigned long long)(p[i].x >> 7);
    atomicAdd(out, acc);
}

// Where does the rate go when the buffer grows?  One buffer of `batch` transforms, the product
// instance, forward / reverse alternating:
//   whole      one launch over the whole buffer
//   pieces     the same buffer in launches of 65536 transforms (launch length held, address range grows)
//   first      launches of 65536 transforms on the first piece only (both held)
//   whole xcd  one launch, every XCD walking its own contiguous eighth
//   2 streams  launches of 32768 transforms (1 GiB), alternating between two streams (piece i of every pass on stream i % 2,
//              so a piece's forward and reverse launches stay ordered): the tail of one launch overlaps the head of the next
//   1 GiB      the same pieces on one stream (what the library does)
static int size_study(uint64_t batch, int rounds)
{
    const size_t n = batch * 4096;
    float2 *data, *tw;
    CK(hipMalloc(&data, n * 8));
    std::vector<float2> t4[2];
    auto W = [](uint32_t idx, bool rev) {
        const double a = (rev ? 2.0 : -2.0) * M_PI * (double)(idx & 4095) / 4096.0;
        return float2{ (float)std::cos(a), (float)std::sin(a) };
    };
    for (int rev = 0; rev < 2; rev++) {
        for (uint32_t mult : { 1u, 4u })
            for (uint32_t r = 1; r < 4; r++)
                for (uint32_t t = 0; t < 256; t++)
                    t4[rev].push_back(W(mult * r * t, rev));
        for (uint32_t mult : { 16u, 64u })
            for (uint32_t r = 1; r < 4; r++)
                for (uint32_t rr = 0; rr < 16; rr++)
                    t4[rev].push_back(W(mult * r * rr, rev));
    }
    CK(hipMalloc(&tw, 2 * t4[0].size() * 8));
    for (int rev = 0; rev < 2; rev++)
        CK(hipMemcpy(tw + rev * t4[0].size(), t4[rev].data(), t4[rev].size() * 8, hipMemcpyHostToDevice));
    const float2 *twf = tw, *twr = tw + t4[0].size();
    const float scale = 1.0f / 4096.0f;
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, data, n, 9u);
    const uint64_t piece = 65536;
    hipStream_t st[2];
    CK(hipStreamCreate(&st[0]));
    CK(hipStreamCreate(&st[1]));
    auto run = [&](int mode, bool rev) {
        const float2 *w = rev ? twr : twf;
        if (mode == 4 || mode == 5) {
            const uint64_t pc = 32768;
            int i = 0;
            for (uint64_t off = 0; off < batch; off += pc, i++) {
                float2 *d = data + off * 4096;
                const uint64_t b = std::min(pc, batch - off);
                hipStream_t q = mode == 4 ? st[i & 1] : (hipStream_t)0;
                if (rev)
                    hipLaunchKernelGGL((sdsp_fft4096_r4_f32<true, 1, 2, 0, 0, 3>), dim3((uint32_t)b), dim3(256), 0, q, d, w, b, scale);
                else
                    hipLaunchKernelGGL((sdsp_fft4096_r4_f32<false, 1, 2, 0, 0, 3>), dim3((uint32_t)b), dim3(256), 0, q, d, w, b, scale);
            }
            return;
        }
        if (mode == 0) {
            if (rev)
                hipLaunchKernelGGL((sdsp_fft4096_r4_f32<true, 1, 2, 0, 0, 3>), dim3((uint32_t)batch), dim3(256), 0, 0, data, w, batch, scale);
            else
                hipLaunchKernelGGL((sdsp_fft4096_r4_f32<false, 1, 2, 0, 0, 3>), dim3((uint32_t)batch), dim3(256), 0, 0, data, w, batch, scale);
        } else if (mode == 3) {
            if (rev)
                hipLaunchKernelGGL((sdsp_fft4096_r4_f32<true, 1, 2, 0, 0, 3, 1>), dim3((uint32_t)batch), dim3(256), 0, 0, data, w, batch, scale);
            else
                hipLaunchKernelGGL((sdsp_fft4096_r4_f32<false, 1, 2, 0, 0, 3, 1>), dim3((uint32_t)batch), dim3(256), 0, 0, data, w, batch, scale);
        } else {
            for (uint64_t off = 0; off < batch; off += piece) {
                float2 *d = data + (mode == 1 ? off : 0) * 4096;
                const uint64_t b = std::min(piece, batch - off);
                if (rev)
                    hipLaunchKernelGGL((sdsp_fft4096_r4_f32<true, 1, 2, 0, 0, 3>), dim3((uint32_t)b), dim3(256), 0, 0, d, w, b, scale);
                else
                    hipLaunchKernelGGL((sdsp_fft4096_r4_f32<false, 1, 2, 0, 0, 3>), dim3((uint32_t)b), dim3(256), 0, 0, d, w, b, scale);
            }
        }
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[6] = { "whole", "pieces", "first", "whole xcd", "2 streams", "1 GiB" };
    for (int i = 0; i < 30; i++) {
        run(2, false);
        run(2, true);
    }
    std::printf("batch %llu = %.1f GiB\n", (unsigned long long)batch, (double)n * 8 / (1 << 30));
    for (int round = 0; round < rounds; round++)
        for (int mode = 0; mode < 6; mode++) {
            for (int i = 0; i < 3; i++) {
                run(mode, false);
                run(mode, true);
            }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 6; i++) {
                run(mode, false);
                run(mode, true);
            }
            CK(hipDeviceSynchronize()); // the two-stream mode runs beside the null stream's events
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double per = ms / 12.0;
            std::printf("%-10s %.4f ms per pass over the buffer, %5.2f %%\n", names[mode], per, (double)batch * 65536.0 / (per * 1e-3) / 8e12 * 100.0);
            std::fflush(stdout);
        }
    return 0;
}

int main(int argc, char **argv)
{
    const uint64_t batch = argc > 1 ? (uint64_t)std::atoll(argv[1]) : 65536;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 2;
    if (argc > 3 && std::string(argv[3]) == "size")
        return size_study(batch, rounds);
    add_sord<0>();
    add_sord<1>();
    add_sord<2>();
    add<1, 2, 0, 0, 4>();
    add<0, 0, 0, 0, 4>();
    add<1, 2, 0, 0, 2>();
    const size_t n = batch * 4096;
    float2 *data, *tw4, *tw2;
    unsigned long long *d_sum;
    CK(hipMalloc(&data, n * 8));
    CK(hipMalloc(&d_sum, 8));
    // thread-twiddle tables, as capi.hip: upload_thread_twiddles_4096 builds them (values rounded from double)
    auto W = [](uint32_t idx, bool rev) {
        const double a = (rev ? 2.0 : -2.0) * M_PI * (double)(idx & 4095) / 4096.0;
        return float2{ (float)std::cos(a), (float)std::sin(a) };
    };
    std::vector<float2> t4[2], t2[2];
    for (int rev = 0; rev < 2; rev++) {
        for (uint32_t mult : { 1u, 4u })
            for (uint32_t r = 1; r < 4; r++)
                for (uint32_t t = 0; t < 256; t++)
                    t4[rev].push_back(W(mult * r * t, rev));
        for (uint32_t mult : { 16u, 64u })
            for (uint32_t r = 1; r < 4; r++)
                for (uint32_t rr = 0; rr < 16; rr++)
                    t4[rev].push_back(W(mult * r * rr, rev));
        for (uint32_t j = 0; j < 4; j++)
            for (uint32_t t = 0; t < 256; t++)
                t2[rev].push_back(W(t << j, rev));
        for (uint32_t j = 0; j < 4; j++)
            for (uint32_t rr = 0; rr < 16; rr++)
                t2[rev].push_back(W((16 * rr) << j, rev));
    }
    CK(hipMalloc(&tw4, 2 * t4[0].size() * 8));
    CK(hipMalloc(&tw2, 2 * t2[0].size() * 8));
    for (int rev = 0; rev < 2; rev++) {
        CK(hipMemcpy(tw4 + rev * t4[0].size(), t4[rev].data(), t4[rev].size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(tw2 + rev * t2[0].size(), t2[rev].data(), t2[rev].size() * 8, hipMemcpyHostToDevice));
    }
    const float2 *tw4f = tw4, *tw4r = tw4 + t4[0].size(), *tw2f = tw2, *tw2r = tw2 + t2[0].size();
    const float scale = 1.0f / 4096.0f;
    const dim3 grid((uint32_t)batch), block(256);

    // ---- every instance computes the same bits
    unsigned long long ref = 0;
    int bad = 0;
    for (size_t i = 0; i < g_inst.size(); i++) {
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, data, n, 7u);
        hipLaunchKernelGGL(g_inst[i].fwd, grid, block, 0, 0, data, tw4f, batch, scale);
        hipLaunchKernelGGL(g_inst[i].rev, grid, block, 0, 0, data, tw4r, batch, scale);
        hipLaunchKernelGGL(g_inst[i].fwd, grid, block, 0, 0, data, tw4f, batch, scale);
        CK(hipMemset(d_sum, 0, 8));
        hipLaunchKernelGGL(checksum_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const uint2 *>(data), n, d_sum);
        unsigned long long s = 0;
        CK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
        if (i == 0)
            ref = s;
        else if (s != ref)
            bad++;
    }
    std::printf("%zu instances, checksum %016llx, %d differ -> %s\n", g_inst.size(), ref, bad, bad ? "FAILED" : "bit-identical");
    std::fflush(stdout);

    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, data, n, 9u);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_pairs = [&](auto launch_fwd, auto launch_rev) {
        for (int i = 0; i < 8; i++) {
            launch_fwd();
            launch_rev();
        }
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 12; i++) {
            launch_fwd();
            launch_rev();
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms / 24.0;
    };
    // wake the device up
    for (int i = 0; i < 100; i++) {
        hipLaunchKernelGGL(g_inst[0].fwd, grid, block, 0, 0, data, tw4f, batch, scale);
        hipLaunchKernelGGL(g_inst[0].rev, grid, block, 0, 0, data, tw4r, batch, scale);
    }
    std::vector<std::vector<double>> res(g_inst.size());
    std::vector<double> r2;
    for (int round = 0; round < rounds; round++) {
        r2.push_back(time_pairs([&] { hipLaunchKernelGGL(sdsp_fft4096_r2_f32<false>, grid, block, 0, 0, data, tw2f, batch, scale); },
                                [&] { hipLaunchKernelGGL(sdsp_fft4096_r2_f32<true>, grid, block, 0, 0, data, tw2r, batch, scale); }));
        for (size_t i = 0; i < g_inst.size(); i++)
            res[i].push_back(time_pairs([&] { hipLaunchKernelGGL(g_inst[i].fwd, grid, block, 0, 0, data, tw4f, batch, scale); },
                                        [&] { hipLaunchKernelGGL(g_inst[i].rev, grid, block, 0, 0, data, tw4r, batch, scale); }));
    }
    const double bytes = (double)batch * 65536.0;
    auto pct = [&](double ms) { return bytes / (ms * 1e-3) / 8e12 * 100.0; };
    std::printf("radix-2 kernel:                              ");
    for (double ms : r2)
        std::printf(" %.4f ms %5.2f %%", ms, pct(ms));
    std::printf("\n");
    for (size_t i = 0; i < g_inst.size(); i++) {
        std::printf("r4 BAR %d SORD %d LORD %d LDSB %d WAVES %d:         ", g_inst[i].bar, g_inst[i].sord, g_inst[i].lord, g_inst[i].ldsb,
                    g_inst[i].waves);
        double best = 1e9;
        for (double ms : res[i]) {
            std::printf(" %.4f ms %5.2f %%", ms, pct(ms));
            best = std::min(best, ms);
        }
        std::printf("   best %5.2f %%\n", pct(best));
    }
    return bad ? 1 : 0;
}

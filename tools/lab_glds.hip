// lab_glds.hip -- what does one wave's global_load_lds_dwordx4 tile fill put where?  (tools only; not part of the library)
//   hipcc --offload-arch=gfx950 -O2 tools/lab_glds.hip -o /tmp/lab_glds && /tmp/lab_glds
// Fills a [64 rows x RB bytes] LDS tile from a row-pitched buffer of uint32 whose value is its own global index, with the
// source-side XOR swizzle of csrc/iir.hip's LDS-DMA kernel, dumps the LDS image and checks every 16-byte chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int RB> __host__ __device__ constexpr int swz(int r) { return RB >= 256 ? (r & 15) : ((r >> 1) & 7); }

__device__ __forceinline__ void glds16(const void *base, uint32_t off, uint32_t dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
}

template <int RB, int SLOTS> __global__ __launch_bounds__(64) void k(const uint32_t *src, uint32_t *out, uint32_t row_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int CPR = RB / 16, RPI = 64 / CPR, NI = CPR, TILE = 64 * RB;
    const int lane = threadIdx.x, q = lane / CPR, c = lane % CPR;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)sm;
    for (int s = 0; s < SLOTS; s++) {
#pragma unroll
        for (int i = 0; i < NI; i++)
            glds16(reinterpret_cast<const char *>(src) + (size_t)s * RB + (size_t)i * RPI * row_bytes,
                   q * row_bytes + 16u * (uint32_t)(c ^ swz<RB>(RPI * i + q)), lds0 + s * TILE + i * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int j = lane; j < SLOTS * TILE / 4; j += 64)
        out[j] = reinterpret_cast<const uint32_t *>(sm)[j];
}

template <int RB, int SLOTS> int run()
{
    const uint32_t row_words = 4096, rows = 64;
    std::vector<uint32_t> h(rows * row_words);
    for (size_t i = 0; i < h.size(); i++)
        h[i] = (uint32_t)i;
    uint32_t *d, *o;
    constexpr int TILE = 64 * RB;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&o, SLOTS * TILE);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<RB, SLOTS>), hipFuncAttributeMaxDynamicSharedMemorySize, SLOTS * TILE);
    hipLaunchKernelGGL((k<RB, SLOTS>), dim3(1), dim3(64), SLOTS * TILE, 0, d, o, row_words * 4);
    std::vector<uint32_t> got(SLOTS * TILE / 4);
    hipError_t e = hipMemcpy(got.data(), o, SLOTS * TILE, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        printf("RB %d slots %d: %s\n", RB, SLOTS, hipGetErrorString(e));
        return 1;
    }
    int bad = 0;
    for (int s = 0; s < SLOTS; s++)
        for (int r = 0; r < 64; r++)
            for (int c = 0; c < RB / 16; c++)
                for (int w = 0; w < 4; w++) {
                    const int srcc = c ^ swz<RB>(r);
                    const uint32_t want = r * row_words + s * RB / 4 + srcc * 4 + w;
                    const uint32_t g = got[(size_t)s * TILE / 4 + r * RB / 4 + c * 4 + w];
                    if (g != want && bad++ < 12)
                        printf("RB %d slot %d row %d chunk %d word %d: got %u (row %u word %u) want %u\n", RB, s, r, c, w, g,
                               g / row_words, g % row_words, want);
                }
    printf("RB %d slots %d: %d wrong words of %d\n", RB, SLOTS, bad, SLOTS * TILE / 4);
    hipFree(d);
    hipFree(o);
    return bad != 0;
}

int main()
{
    int rc = 0;
    rc |= run<128, 2>();
    rc |= run<256, 2>();
    rc |= run<512, 1>();
    rc |= run<512, 2>();
    return rc;
}

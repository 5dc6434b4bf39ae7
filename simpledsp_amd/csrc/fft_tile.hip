// fft_tile.hip -- general LDS-resident FFT for gfx950: any power-of-two length that fits LDS,
// f32 or f64, forward or reverse, with genuine radix-2 DIT stages (sdsp::fft_radix2,
// fft.h:258-299) or radix-4 DIF stages (sdsp::fft_radix4, fft.h:301-360).
//
// One workgroup holds `cols` independent sequences of length n in LDS as tile[row][col].  The
// reference's separate permutation sweeps (fft.h:269-273 bit reversal before the radix-2 stages,
// fft.h:351-355 base-4 digit reversal after the radix-4 stages) are folded into the LDS row
// address on load / store, so HBM is touched exactly once each way and always with the lanes
// running along the contiguous axis.  The same kernel is the column pass and the row pass of the
// four-step decomposition used for transforms larger than LDS (BASELINE config 3), where the
// inter-pass twiddle W_N^(n2*k1) is fused into the first pass's store.
//
// This is the coverage kernel (every size / radix / precision / direction the reference's tests
// touch); the tuned kernel for the headline shape lives in fft4096.hip.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "sdsp_hip_internal.h"

namespace sdsp_hip
{
namespace
{
template <typename R> struct cplx_of;
template <> struct cplx_of<float> { using type = float2; };
template <> struct cplx_of<double> { using type = double2; };

template <typename C> __device__ __forceinline__ C cadd(C a, C b) { return C{ a.x + b.x, a.y + b.y }; }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { return C{ a.x - b.x, a.y - b.y }; }
template <typename C> __device__ __forceinline__ C cmul(C a, C b)
{
    return C{ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x };
}
// multiply by -i (forward) or +i (reverse): the reference's swap/negate at fft.h:339-340
template <typename C> __device__ __forceinline__ C rot90(C a, bool reverse)
{
    return reverse ? C{ -a.y, a.x } : C{ a.y, -a.x };
}

__device__ __forceinline__ uint32_t rev_bits(uint32_t x, uint32_t log2n)
{
    return __brev(x) >> (32u - log2n);
}
// base-4 digit reversal = bit reversal with the two bits of every digit swapped back
__device__ __forceinline__ uint32_t rev_digits4(uint32_t x, uint32_t log2n)
{
    const uint32_t r = __brev(x) >> (32u - log2n);
    return ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
}

struct tile_dev_args {
    fft_tile_args a;
    uint32_t log2cols;
};

template <typename R, int RADIX>
__global__ __launch_bounds__(256) void sdsp_fft_tile_kernel(tile_dev_args p)
{
    using C = typename cplx_of<R>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char sdsp_tile_smem[];
    C *lds = reinterpret_cast<C *>(sdsp_tile_smem);
    const fft_tile_args &a = p.a;

    const uint32_t n = a.n, log2n = a.log2n, cols = a.cols, pitch = a.pitch;
    const uint32_t log2cols = p.log2cols, cmask = cols - 1;
    const bool reverse = a.reverse != 0;
    const uint64_t tile = blockIdx.x;
    const uint64_t group = tile / a.tiles_per_group;
    const uint32_t tig = (uint32_t)(tile % a.tiles_per_group);
    const uint64_t col0 = tile * cols;
    const uint32_t valid = (uint32_t)min((uint64_t)cols, a.total_cols - col0);

    const C *src = reinterpret_cast<const C *>(a.in) + group * a.group_stride + (uint64_t)tig * a.in_tile_step;
    C *dst = reinterpret_cast<C *>(a.out) + group * a.group_stride + (uint64_t)tig * a.out_tile_step;
    const C *tw = reinterpret_cast<const C *>(a.tw);
    const uint32_t total = n << log2cols;

    // ---- HBM -> LDS (radix 2: rows land bit-reversed, fft.h:269-273)
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        uint32_t i, c;
        if (a.in_c_fast) {
            c = e & cmask;
            i = e >> log2cols;
        } else {
            i = e & (n - 1);
            c = e >> log2n;
        }
        if (c < valid) {
            const C v = src[(uint64_t)i * a.in_si + (uint64_t)c * a.in_sc];
            const uint32_t row = RADIX == 2 ? rev_bits(i, log2n) : i;
            lds[row * pitch + c] = v;
        }
    }
    __syncthreads();

    if constexpr (RADIX == 2) {
        // log2(n) DIT stages of n/2 butterflies, fft.h:276-294
        const uint32_t work = total >> 1;
        for (uint32_t s = 0; s < log2n; s++) {
            const uint32_t half = 1u << s;
            const uint32_t tshift = log2n - s - 1;
            for (uint32_t e = threadIdx.x; e < work; e += blockDim.x) {
                const uint32_t c = e & cmask;
                const uint32_t q = e >> log2cols;
                const uint32_t k = q & (half - 1);
                const uint32_t i1 = ((q >> s) << (s + 1)) + k;
                const uint32_t i2 = i1 + half;
                const C w = tw[k << tshift];
                const C x1 = lds[i1 * pitch + c];
                const C t = cmul(lds[i2 * pitch + c], w);
                lds[i1 * pitch + c] = cadd(x1, t);
                lds[i2 * pitch + c] = csub(x1, t);
            }
            __syncthreads();
        }
    } else {
        // log4(n) DIF stages of n/4 butterflies, fft.h:311-349.  The reference multiplies the
        // INPUTS of stage i by the twiddles that stage i-1 owes (fft.h:322-338); here each stage
        // applies its own output twiddles -- the same products, one stage earlier.
        const uint32_t work = total >> 2;
        const uint32_t stages = log2n >> 1;
        for (uint32_t s = 0; s < stages; s++) {
            const uint32_t log2g = log2n - 2 * (s + 1);
            const uint32_t g = 1u << log2g;
            for (uint32_t e = threadIdx.x; e < work; e += blockDim.x) {
                const uint32_t c = e & cmask;
                const uint32_t q = e >> log2cols;
                const uint32_t k = q & (g - 1);
                const uint32_t base = ((q >> log2g) << (log2g + 2)) + k;
                C *p0 = lds + base * pitch + c;
                const uint32_t gp = g * pitch;
                const C x0 = p0[0], x1 = p0[gp], x2 = p0[2 * gp], x3 = p0[3 * gp];
                const C t0 = cadd(x0, x2), t1 = csub(x0, x2);
                const C t2 = cadd(x1, x3), t3 = rot90(csub(x1, x3), reverse);
                C y0 = cadd(t0, t2), y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
                if (g > 1 && k > 0) { // index 0 is exactly 1: skipped like fft.h:327
                    const uint32_t idx = k << (2 * s);
                    y1 = cmul(y1, tw[idx]);
                    y2 = cmul(y2, tw[2 * idx]);
                    y3 = cmul(y3, tw[3 * idx]);
                }
                p0[0] = y0;
                p0[gp] = y1;
                p0[2 * gp] = y2;
                p0[3 * gp] = y3;
            }
            __syncthreads();
        }
    }

    // ---- LDS -> HBM (radix 4: X[k] sits in row digit_reverse4(k), fft.h:351-355)
    const C *twb = reinterpret_cast<const C *>(a.tw_big);
    const R scale = sizeof(R) == 4 ? (R)a.scale : (R)a.scale_d;
    const bool do_scale = a.apply_scale != 0;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        uint32_t k, c;
        if (a.out_c_fast) {
            c = e & cmask;
            k = e >> log2cols;
        } else {
            k = e & (n - 1);
            c = e >> log2n;
        }
        if (c < valid) {
            const uint32_t row = RADIX == 2 ? k : rev_digits4(k, log2n);
            C v = lds[row * pitch + c];
            if (twb) // four-step inter-pass twiddle W_N^(n2 * k1)
                v = cmul(v, twb[(uint64_t)(tig * cols + c) * k]);
            if (do_scale) { // reverse_fft::ScaleValues, fft.h:128-132
                v.x *= scale;
                v.y *= scale;
            }
            dst[(uint64_t)k * a.out_sk + (uint64_t)c * a.out_sc] = v;
        }
    }
}

template <typename R, int RADIX>
int launch_one(const tile_dev_args &p, uint64_t n_tiles, size_t lds, hipStream_t stream)
{
    auto kern = sdsp_fft_tile_kernel<R, RADIX>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return fail(SDSP_HIP_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    }
    // grids above 2^31-1 workgroups are split (x dimension limit)
    const uint64_t max_grid = 0x7fffffffull;
    if (n_tiles > max_grid)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((uint32_t)n_tiles), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("fft tile launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}
} // namespace

namespace
{
template <typename C> __global__ __launch_bounds__(256) void sdsp_pointwise_mul_kernel(C *data, const C *h, uint32_t n, uint64_t total)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256)
        data[i] = cmul(data[i], h[i & (n - 1)]); // n is a power of two
}
} // namespace

int launch_pointwise_mul(int precision, void *data, const void *h, uint32_t n, uint64_t batch, void *stream)
{
    const uint64_t total = (uint64_t)n * batch;
    if (total == 0)
        return SDSP_HIP_OK;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256u * 16u);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (precision == SDSP_HIP_F64)
        hipLaunchKernelGGL(sdsp_pointwise_mul_kernel<double2>, dim3(blocks), dim3(256), 0, s,
                           reinterpret_cast<double2 *>(data), reinterpret_cast<const double2 *>(h), n, total);
    else
        hipLaunchKernelGGL(sdsp_pointwise_mul_kernel<float2>, dim3(blocks), dim3(256), 0, s,
                           reinterpret_cast<float2 *>(data), reinterpret_cast<const float2 *>(h), n, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("pointwise launch: ") + hipGetErrorString(e));
    return SDSP_HIP_OK;
}

size_t fft_tile_lds_bytes(int precision, uint32_t n, uint32_t pitch)
{
    return (size_t)n * pitch * (precision == SDSP_HIP_F64 ? 16 : 8);
}

size_t fft_tile_max_lds_bytes() { return 128 * 1024; }

int launch_fft_tile(int precision, int radix, const fft_tile_args &a, uint64_t n_tiles, void *stream)
{
    if (n_tiles == 0)
        return SDSP_HIP_OK;
    tile_dev_args p;
    p.a = a;
    p.log2cols = sdsp_hip_log2(a.cols);
    const size_t lds = fft_tile_lds_bytes(precision, a.n, a.pitch);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (precision == SDSP_HIP_F32)
        return radix == 2 ? launch_one<float, 2>(p, n_tiles, lds, s) : launch_one<float, 4>(p, n_tiles, lds, s);
    return radix == 2 ? launch_one<double, 2>(p, n_tiles, lds, s) : launch_one<double, 4>(p, n_tiles, lds, s);
}
} // namespace sdsp_hip

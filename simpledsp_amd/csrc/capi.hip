// capi.hip -- the extern "C" boundary declared in include/sdsp_hip.h: plans, launches, host and
// multi-device convenience paths.  Everything that computes goes to the HIP kernels in
// fft_tile.hip / fft4096.hip / iir.hip; there is no CPU implementation behind these entry points.
#include <hip/hip_runtime.h>

#include <cstring>
#include <thread>
#include <vector>

#include "sdsp_hip_internal.h"

using namespace sdsp_hip;

int sdsp_hip::ensure_dynamic_lds(const void *kernel, size_t bytes, std::atomic<uint64_t> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("hipGetDevice: ") + hipGetErrorString(e));
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit)
        return SDSP_HIP_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
        return fail(SDSP_HIP_ERR_HIP, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
    done.fetch_or(bit, std::memory_order_release);
    return SDSP_HIP_OK;
}

namespace
{
int hip_fail(hipError_t e, const char *what)
{
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice)
        return fail(SDSP_HIP_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    return fail(SDSP_HIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                          \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess)                  \
            return hip_fail(_e, #expr);        \
    } while (0)

int use_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(SDSP_HIP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= count)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    return SDSP_HIP_OK;
}

size_t esize(int precision) { return precision == SDSP_HIP_F64 ? 16 : 8; } // one complex element

// round a double table to the plan precision and park it in HBM
// launch granularity, sdsp_hip.h: sdsp_hip_set_launch_piece_bytes
std::atomic<uint64_t> g_piece_bytes{ SDSP_HIP_DEFAULT_PIECE_BYTES };

// units (transforms, channels) per launch for a batch of `units` x `unit_bytes`; a multiple of `multiple`, so that a piece
// boundary never cuts a workgroup's tile
uint64_t piece_units(uint64_t units, uint64_t unit_bytes, uint64_t multiple)
{
    const uint64_t pb = g_piece_bytes.load(std::memory_order_relaxed);
    if (pb == 0 || unit_bytes == 0 || units * unit_bytes <= pb + pb / 2) // a tail under half a piece rides along
        return units;
    const uint64_t u = pb / unit_bytes / multiple * multiple;
    return u ? u : multiple;
}

int upload_twiddles(const std::vector<double> &w, int precision, void **dev)
{
    const size_t n = w.size() / 2;
    if (precision == SDSP_HIP_F64) {
        HIP_TRY(hipMalloc(dev, n * 16));
        HIP_TRY(hipMemcpy(*dev, w.data(), n * 16, hipMemcpyHostToDevice));
    } else {
        std::vector<float> wf(w.size());
        for (size_t i = 0; i < w.size(); i++)
            wf[i] = (float)w[i];
        HIP_TRY(hipMalloc(dev, n * 8));
        HIP_TRY(hipMemcpy(*dev, wf.data(), n * 8, hipMemcpyHostToDevice));
    }
    return SDSP_HIP_OK;
}

// Thread-twiddle table of the tuned N = 4096 f32 kernels (fft4096.hip): the stage twiddles each thread needs,
// taken from the row W_4096^j (same rounded values) and laid out [value][thread] so the kernel reads them
// with coalesced loads.  radix 4: W^(r t), W^(4 r t) (r = 1..3, t < 256), then W^(16 r rr), W^(64 r rr)
// (rr < 16); radix 2: W^(t << j) (j < 4), then W^((16 rr) << j).
int upload_thread_twiddles_4096(const std::vector<double> &w, int radix, void **dev)
{
    std::vector<float> tab;
    auto put = [&](uint32_t idx) {
        tab.push_back((float)w[2 * idx]);
        tab.push_back((float)w[2 * idx + 1]);
    };
    if (radix == 4) {
        for (uint32_t mult : { 1u, 4u })
            for (uint32_t r = 1; r < 4; r++)
                for (uint32_t t = 0; t < 256; t++)
                    put(mult * r * t);
        for (uint32_t mult : { 16u, 64u })
            for (uint32_t r = 1; r < 4; r++)
                for (uint32_t rr = 0; rr < 16; rr++)
                    put(mult * r * rr);
    } else {
        for (uint32_t j = 0; j < 4; j++)
            for (uint32_t t = 0; t < 256; t++)
                put(t << j);
        for (uint32_t j = 0; j < 4; j++)
            for (uint32_t rr = 0; rr < 16; rr++)
                put((16 * rr) << j);
    }
    HIP_TRY(hipMalloc(dev, tab.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(*dev, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return SDSP_HIP_OK;
}

// Thread-twiddle table of the register-pass families (fft_reg.hip, fft_reg64.hip): for every pass I that has thread
// twiddles (point stride S = N >> 4(I+1) > 1) and every thread t < N/16 of a transform (r = t mod S,
// unit = r * 16^I): six slots -- radix 2: W^(unit << v), v < 4; radix 4: W^(unit q), W^(4 unit q), q = 1..3.
int upload_thread_twiddles_reg(const std::vector<double> &w, uint32_t n, int radix, int precision, void **dev)
{
    const uint32_t log2n = sdsp_hip_log2(n), T = n / 16, P = (log2n + 3) / 4;
    std::vector<double> tab((size_t)6 * P * T * 2, 0.0);
    for (uint32_t I = 0; I + 1 < P; I++) {
        const uint32_t S = n >> (4 * (I + 1));
        for (uint32_t t = 0; t < T; t++) {
            const uint32_t unit = (t % S) << (4 * I);
            for (uint32_t v = 0; v < 6; v++) {
                uint32_t idx;
                if (radix == 2)
                    idx = v < 4 ? unit << v : 0;
                else
                    idx = v < 3 ? unit * (v + 1) : 4 * unit * (v - 2);
                const size_t o = ((size_t)(6 * I + v) * T + t) * 2;
                tab[o] = w[2 * (size_t)idx];
                tab[o + 1] = w[2 * (size_t)idx + 1];
            }
        }
    }
    return upload_twiddles(tab, precision, dev); // rounds to the plan precision exactly like the row itself
}

// Thread-twiddle table of fft_wave.hip's N = 256 / 512 / 2048 kernels.  Radix 2: [global stage g][lane t] = W^((t mod s_i) << g) for
// the stages of pass i < last (P = n / 64 points per lane, log2 P stages per pass, s_i = n >> (log2 P (i + 1))); the last
// pass has no thread twiddles.
int upload_thread_twiddles_wave(const std::vector<double> &w, uint32_t n, int radix, void **dev)
{
    if (radix == 4) { // N = 256: [pass i < 3][q = 1 .. 3][lane] = W^(q (t mod s_i) 4^i), s_i = 64 >> 2 i
        std::vector<double> tab((size_t)3 * 3 * 64 * 2, 0.0);
        for (uint32_t i = 0; i < 3; i++)
            for (uint32_t q = 1; q < 4; q++)
                for (uint32_t t = 0; t < 64; t++) {
                    const size_t idx = (size_t)q * (t % (64u >> (2 * i))) << (2 * i);
                    const size_t o = ((size_t)(i * 3 + q - 1) * 64 + t) * 2;
                    tab[o] = w[2 * idx];
                    tab[o + 1] = w[2 * idx + 1];
                }
        return upload_twiddles(tab, SDSP_HIP_F32, dev);
    }
    const uint32_t L = sdsp_hip_log2(n), LP = L - 6, NP = (L + LP - 1) / LP;
    std::vector<double> tab((size_t)(NP - 1) * LP * 64 * 2, 0.0);
    for (uint32_t i = 0; i + 1 < NP; i++) {
        const uint32_t sg = n >> (LP * (i + 1));
        for (uint32_t s = 0; s < LP; s++)
            for (uint32_t t = 0; t < 64; t++) {
                const size_t idx = (size_t)(t % sg) << (i * LP + s);
                const size_t o = ((size_t)(i * LP + s) * 64 + t) * 2;
                tab[o] = w[2 * idx];
                tab[o + 1] = w[2 * idx + 1];
            }
    }
    return upload_twiddles(tab, SDSP_HIP_F32, dev);
}

// Thread-twiddle table of fft_big.hip: [pass (A, B)][stage s < 5][thread t < N/32] = W^(t << s) for pass A,
// W^((32 v) << s), v = t mod (N/1024), for pass B.
int upload_thread_twiddles_big(const std::vector<double> &w, uint32_t n, int precision, void **dev)
{
    const uint32_t T = n / 32, vmask = n / 1024 - 1;
    std::vector<double> tab((size_t)10 * T * 2);
    for (uint32_t pass = 0; pass < 2; pass++)
        for (uint32_t s = 0; s < 5; s++)
            for (uint32_t t = 0; t < T; t++) {
                const uint32_t idx = (pass == 0 ? t : 32 * (t & vmask)) << s;
                const size_t o = ((size_t)(pass * 5 + s) * T + t) * 2;
                tab[o] = w[2 * (size_t)idx];
                tab[o + 1] = w[2 * (size_t)idx + 1];
            }
    return upload_twiddles(tab, precision, dev); // rounds to the plan precision exactly like the row itself
}

// Thread-twiddle table of fft_big.hip's radix-4 form (N = 16384 = 4^7 and N = 4096 = 4^6, fft32_r4.h): [slot < 14][thread t < N/32].
// For N = 16384: slots 0-2: W_N^(q t); 3-5: W_4096^(q t); 6, 7: stage 2's pair for the thread's block parity -- (1, W_1024^(2v)) for an
// even block, (W_1024^v, W_1024^(3v)) for an odd one; 8-10: W_256^(q v); 11-13: W_64^(q v); q = 1, 2, 3, v = t mod 16, block = t / 16.
// `w` is the row W_N^j, direction-folded.  tools/model_fft_big_r4.py is the index arithmetic's model.
int upload_thread_twiddles_big_r4(const std::vector<double> &w, uint32_t n, int precision, void **dev)
{
    const uint32_t T = n / 32, R = sdsp_hip_log2(n) - 10; // R = 4 (N = 16384) or 2 (N = 4096); v = t mod 2^R, block = t >> R
    std::vector<double> tab((size_t)14 * T * 2);
    auto put = [&](uint32_t slot, uint32_t t, uint64_t idx) {
        idx %= n;
        tab[((size_t)slot * T + t) * 2] = w[2 * idx];
        tab[((size_t)slot * T + t) * 2 + 1] = w[2 * idx + 1];
    };
    for (uint32_t t = 0; t < T; t++) {
        const uint32_t v = t & ((1u << R) - 1), odd = (t >> R) & 1;
        for (uint32_t q = 1; q <= 3; q++) {
            put(q - 1, t, (uint64_t)q * t);        // W_N^(q t)
            put(2 + q, t, (uint64_t)4 * q * t);    // W_(N/4)^(q t) = W_N^(4 q t)
            put(7 + q, t, (uint64_t)64 * q * v);   // W_(2^(R+4))^(q v) = W_N^(64 q v)    (N = 16384: W_256^(q v))
            put(10 + q, t, (uint64_t)256 * q * v); // W_(2^(R+2))^(q v) = W_N^(256 q v)   (N = 16384: W_64^(q v))
        }
        put(6, t, odd ? (uint64_t)16 * v : 0);             // stage 2 (G = N/16), j < 16: q = 1 (odd block) / none
        put(7, t, (uint64_t)16 * (odd ? 3 : 2) * v);       // j >= 16: q = 3 (odd) / 2 (even)
    }
    return upload_twiddles(tab, precision, dev); // rounds to the plan precision exactly like the row itself
}

// N = 16384 radix-4 plans: fft_big.hip's radix-4 form is variant 0 and the fft_mix.hip kernel variant 1
inline bool big_r4_form(uint32_t n, int radix) { return radix == 4 && (n == 16384 || n == 4096); } // table of the radix-4 form (4096: real-input plans)
// (N = 8192, AUTO plans: fft_big.hip's radix-2 stages measured 76.9-77.9 % against 74.1-76.2 % for the mixed-radix kernel in one
// run once their thread twiddles were fetched ahead of the passes, so they are variant 0 there too and fft_mix.hip variant 1)
inline bool big_is_default(uint32_t n, int radix) { return big_r4_form(n, radix) || n == 8192; }
// f64 radix-2 plans the double-precision registers-resident kernel serves: the variant number that selects it
inline int big64_variant(uint32_t) { return 0; } // 4096: 73.2-73.6 % against 66.5-68.6 % (fft_reg64.hip), one call, round 3

constexpr uint64_t kFft1mQueues = 8, kFft1mRing = 3; // persistent N = 2^20 kernel: ticket queues x intermediates per queue
constexpr uint64_t kFused2pUnitsPerLaunch = 1024; // units (8 - 32 MiB each) one persistent launch covers (sizes the counter block)
constexpr uint64_t kFft1mPerLaunch = 4096; // transforms one persistent launch covers (sizes the counter block)
// Tables of fft_mix.hip (N = R x 4096, R = 2 / 4): the N = 4096 radix-4 thread-twiddle table built from W_4096^j = W_N^(R j),
// and the leading stage's thread twiddles [q - 1][t] = W_N^(q t), q < R, t < 256 (R = 2) / 512 (R = 4).
int upload_thread_twiddles_mix(const std::vector<double> &w, uint32_t n, void **sub, void **lead)
{
    const uint32_t R = n / 4096;
    std::vector<double> w4096(2 * 4096);
    for (uint32_t j = 0; j < 4096; j++) {
        w4096[2 * j] = w[2 * (size_t)(R * j)];
        w4096[2 * j + 1] = w[2 * (size_t)(R * j) + 1];
    }
    if (int rc = upload_thread_twiddles_4096(w4096, 4, sub))
        return rc;
    std::vector<float> tab;
    const uint32_t threads = R == 4 ? 512 : 256; // the N = 16384 kernel runs 512 threads per transform
    for (uint32_t q = 1; q < R; q++)
        for (uint32_t t = 0; t < threads; t++) {
            tab.push_back((float)w[2 * (size_t)(q * t)]);
            tab.push_back((float)w[2 * (size_t)(q * t) + 1]);
        }
    HIP_TRY(hipMalloc(lead, tab.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(*lead, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return SDSP_HIP_OK;
}

enum fft_path { PATH_NOOP = 0, PATH_TILE = 1, PATH_FFT4096 = 2, PATH_FOUR_STEP = 3, PATH_FFT1M = 4, PATH_REG = 5 };
} // namespace

struct sdsp_hip_fft_plan {
    uint32_t n = 0;
    int radix = 0, direction = 0, precision = 0, device = 0;
    uint64_t max_batch = 0;
    int path = PATH_NOOP;
    int variant = 0;
    void *tw = nullptr;            // W_n (single pass) or W_N (four-step inter-pass twiddle)
    void *tw1 = nullptr;           // four-step: W_n1
    void *tw2 = nullptr;           // four-step: W_n2
    void *twt = nullptr;           // tuned N = 4096 f32 kernels: thread-twiddle table
    void *twt_reg = nullptr;       // register-pass family (f32): thread-twiddle table
    void *twt_big = nullptr;       // fft_big.hip: thread-twiddle table
    void *twt_wave = nullptr;      // fft_wave.hip, N = 256 / 512 / 2048 radix 2: [stage][lane] table
    void *twt_mix = nullptr;       // fft_mix.hip (N = 8192 / 16384): the sub-transforms' thread-twiddle table ...
    void *tw_lead = nullptr;       // ... and the leading stage's thread twiddles W_N^(q t)
    uint32_t n1 = 0, n2 = 0;       // four-step split
    uint32_t cols = 1, pitch = 1;  // tile shape (single pass)
    uint32_t cols1 = 1, pitch1 = 1, cols2 = 1, pitch2 = 1;
    void *workspace = nullptr;
    uint64_t workspace_bytes = 0;
    uint64_t ws_batch = 0;         // transforms the workspace holds (multi-pass paths run in slices)
    uint64_t twiddle_bytes = 0;
    void *host_stage = nullptr;    // device staging buffer of the *_host path
    uint64_t host_stage_bytes = 0;
    void *sync = nullptr;          // persistent two-pass kernels: ticket / arrival counters
    uint64_t sync_count = 0;       // ... units (N = 2^20 f32: transforms) one launch covers
    uint64_t sticky_off = 0;       // ... byte offset of the sticky abort word behind the per-launch block
    uint32_t f2_unit = 0, f2_queues = 0, f2_ring = 0, f2_lag = 0; // fft_2pass.hip's persistent schedule (0 = workspace too small)
    uint64_t wait_limit = 200000000ull; // ... and what a hand-off poll may take (100 MHz ticks: 2 s) before the launch gives up
    sdsp_hip_fft_plan *partner = nullptr; // reverse plan of the generic convolution path (lazy)
    sdsp_hip_fft_plan *mid_rows = nullptr; // N = 2^16 .. 2^19 f32: plan of the 16 row transforms (fft_mid.hip)
    void *tw1024 = nullptr;                // ... and W_1024^j, the coarse factor of its inter-pass twiddle
    int real_mode = 0;                    // 0 complex; 1 real forward; 2 real inverse (n = n_real / 2)
    bool allow_mix = false;               // radix-4 stages behind one leading radix-2 / radix-4 stage may serve this plan
};

struct sdsp_hip_iir_plan {
    uint32_t sections = 0;
    int kind = 0, precision = 0, device = 0, variant = 0;
    double gain = 1.0;
    double a[3 * SDSP_HIP_MAX_SECTIONS] = {};
    double b[3 * SDSP_HIP_MAX_SECTIONS] = {};
};

struct sdsp_hip_fir_plan {
    uint32_t taps = 0;
    int precision = 0, device = 0, variant = 0;
    void *h_dev = nullptr;
};

namespace
{
// largest power-of-two column count whose padded tile fits the LDS budget
void pick_tile(int precision, uint32_t n, uint32_t want_cols, uint32_t *cols, uint32_t *pitch)
{
    uint32_t c = want_cols;
    while (c > 1 && fft_tile_lds_bytes(precision, n, c + 1) > fft_tile_max_lds_bytes())
        c >>= 1;
    *cols = c;
    *pitch = c > 1 ? c + 1 : 1; // odd pitch: both access orders spread over the banks
}

// the multi-pass paths' plan-owned workspace, allocated on the first exec that needs it (single-pass default kernels
// such as fft_big at N = 32768 never do)
int ensure_workspace(sdsp_hip_fft_plan *p)
{
    if (p->workspace || p->workspace_bytes == 0)
        return SDSP_HIP_OK;
    hipError_t e = hipMalloc(&p->workspace, p->workspace_bytes);
    if (e != hipSuccess) {
        p->workspace = nullptr;
        return fail(SDSP_HIP_ERR_NOMEM, std::string("workspace hipMalloc: ") + hipGetErrorString(e));
    }
    return SDSP_HIP_OK;
}

// the persistent kernels' sticky abort word: set by any launch whose bounded hand-off wait gave up, cleared by
// sdsp_hip_fft_exec at the start of a call (the per-launch abort flag beside the tickets is re-zeroed for every launch)
void *fft1m_sticky(const sdsp_hip_fft_plan *p) { return reinterpret_cast<char *>(p->sync) + p->sticky_off; }

// chunk sizes of the multi-pass schedules (shared by exec and the launch count)
uint64_t fft1m_chunk(const sdsp_hip_fft_plan *p) { return std::max<uint64_t>(1, std::min<uint64_t>(32, p->ws_batch)); }
uint64_t fft2p_chunk(const sdsp_hip_fft_plan *p) // an intermediate of at most 256 MiB per chunk
{
    const uint64_t cap = 1ull << 28; // the Infinity Cache: 37 % at 256 MiB, 35 % at 128 / 192, 33 - 34 % at 288 MiB and beyond (profiles/r03_fft2p_chunk_lab.txt)
    return std::max<uint64_t>(1, std::min<uint64_t>(p->ws_batch, cap / ((uint64_t)p->n * esize(p->precision))));
}

// ------------------------------------------------------------------------------------------------------------------
// The ONE dispatch table: which kernel serves (plan, variant).  sdsp_hip_fft_exec, sdsp_hip_fft_plan_get_info and
// sdsp_hip_fft_plan_launches all go through select_kernel(), so what the plan reports is what runs, by construction.
enum fft_kernel_id {
    K_NOOP = 0,
    K_FFT4096_R4,    // fft4096.hip: the headline kernel (cfg 2 / 5)
    K_FFT4096_R2,    // fft4096.hip: its radix-2 sibling
    K_MIX,           // fft_mix.hip: one leading radix-2 / radix-4 stage + the N = 4096 radix-4 machinery
    K_BIG,           // fft_big.hip: transform in registers, N = 8192 .. 32768
    K_BIG_REAL,      // fft_big.hip, REAL form (real-input plans, n = n_real / 2 = 2048 .. 32768)
    K_BIG64,         // fft_big64.hip: the same design in double, N = 4096 .. 16384
    K_BIG64_REAL,    // fft_big64.hip, REAL form (real-input plans in double, n = n_real / 2 = 4096 .. 16384)
    K_REG64,         // fft_reg64.hip: register-pass family in double
    K_WAVE64,        // fft_wave.hip: N = 1024 in double, one transform per two waves
    K_WAVE1024,      // fft_wave.hip: N = 1024 f32, one transform per wave
    K_WAVE2,         // fft_wave.hip: N = 256 / 512 / 2048 f32
    K_REG32,         // fft_reg.hip: register-pass family, f32
    K_TILE,          // fft_tile.hip: coverage kernel, one pass
    K_FFT1M_CHUNKED, // fft1m.hip: two launches per chunk of <= 32 transforms
    K_FFT1M_FUSED,   // fft1m.hip: one persistent launch (cfg 3)
    K_2PASS,         // fft_2pass.hip: N = N1 x N2, two passes
    K_2PASS_FUSED,   // fft_2pass.hip: the same tiles in one persistent, ticketed launch
    K_MID,           // fft_mid.hip: 16-point column step + row plan + untwist
    K_FOUR_STEP,     // fft_tile.hip twice: the general four-step
    K_UNSUPPORTED,   // no kernel serves this (plan, variant)
};

struct fft_kernel_sel {
    fft_kernel_id id;
    const char *name;  // as rocprofv3 prints the dominant kernel(s)
    int hbm_passes;    // passes over HBM of one transform
    int stage_radix;   // butterflies that run: 2, 4, or SDSP_HIP_STAGES_2_THEN_4
    bool workspace;    // needs the plan-owned workspace
    bool pieces;       // one launch per batch, issued in launch pieces (sdsp_hip_set_launch_piece_bytes)
};

fft_kernel_sel select_kernel(const sdsp_hip_fft_plan *p, int variant)
{
    const bool f32 = p->precision == SDSP_HIP_F32;
    // launch pieces: kernels with many short workgroups (N <= 8192); see fft_exec_pieces
    const bool pc = (p->path == PATH_FFT4096 || p->path == PATH_REG || p->path == PATH_TILE) && p->n <= 8192;
    if (p->path == PATH_NOOP)
        return { K_NOOP, "none", 0, p->radix, false, false };
    if (p->path == PATH_FFT4096 && variant < fft4096_num_variants())
        return { K_FFT4096_R4, "sdsp_fft4096_r4_f32", 1, 4, false, pc };
    if (p->path == PATH_REG && f32 && variant == 0 && p->n == 4096 && p->radix == 2 && !p->real_mode)
        return { K_FFT4096_R2, "sdsp_fft4096_r2_f32", 1, 2, false, pc };
    // N = 8192 / 16384 f32, either stage type: one leading radix-2 / radix-4 stage + the tuned N = 4096 radix-4 machinery
    const bool mix_size = p->path == PATH_REG && f32 && !p->real_mode && p->tw_lead;
    const int mix_variant = big_is_default(p->n, p->radix) ? 1 : 0;
    if (mix_size && variant == mix_variant)
        return { K_MIX, "sdsp_fft_mix_f32", 1, p->n == 8192 ? SDSP_HIP_STAGES_2_THEN_4 : 4, false, pc };
    // N = 32768 (variant 0) and N = 8192 / 16384 (variant 0 or 1, see big_is_default), f32: registers-resident kernel
    if ((p->path == PATH_REG || p->path == PATH_FOUR_STEP) && f32 && variant == (mix_size ? 1 - mix_variant : 0) && !p->real_mode &&
        fft_big_supports(p->n, p->radix))
        return { K_BIG, "sdsp_fft_big_kernel", 1, big_r4_form(p->n, p->radix) ? 4 : 2, false, pc };
    // real-input plans of n_real = 4096 .. 65536: split / merge inside the registers-resident kernel; variants 1 / 2 keep
    // the register-pass family's MODE 1 / 2
    if (p->path == PATH_REG && f32 && variant == 0 && p->real_mode && p->twt_big && fft_big_real_supports(p->n, p->radix))
        return { K_BIG_REAL, "sdsp_fft_big_kernel", 1, big_r4_form(p->n, p->radix) ? 4 : 2, false, pc };
    // double precision, N = 4096 / 8192 / 16384: the registers-resident kernel in double (fft_big64.hip) -- radix-2 stages, or, for
    // radix-4 plans of N = 4096 / 16384, genuine radix-4 stages (its R4 form).
    // The default of all three sizes: 73.4 / 74.5 / 59.5 % of HBM peak against 67.5 % (N = 4096, fft_reg64.hip), 51.4 % (N = 8192:
    // the whole tile in LDS) and 24.2 % (N = 16384: three streaming passes) in one call (tools/sweep_sizes64.py, round 3);
    // what served a size before is its variant 1
    const bool big64 = !f32 && !p->real_mode && p->twt_big && fft_big64_supports(p->n, p->radix);
    if (big64 && variant == big64_variant(p->n))
        return { K_BIG64, "sdsp_fft_big_f64_kernel", 1, p->radix == 4 ? 4 : 2, false, pc && p->n <= 4096 };
    // real-input plans in double of n_real = 8192 / 16384 / 32768 (radix 2): split / merge around the same transform (its REAL form);
    // variant 1 keeps the register-pass family's MODE 1 / 2 (n_real <= 16384)
    const bool big64_real = !f32 && p->real_mode && p->twt_big && fft_big64_real_supports(p->n, p->radix);
    if (big64_real && variant == 0)
        return { K_BIG64_REAL, "sdsp_fft_big_f64_real_kernel", 1, 2, false, pc && p->n <= 4096 };
    if (p->path == PATH_REG && !f32 && fft_reg64_supports(p->n, p->radix)) {
        if ((big64 && big64_variant(p->n) == 0) || big64_real) // the kernel that was the default becomes variant 1
            variant = variant == 1 ? 0 : variant;
        const bool wave64 = !p->real_mode && fft_wave_supports(p->n, p->radix);
        if (variant == 1 && wave64) // N = 1024 alternate: same bits; measured 71.4-72.1 % against 71.7-72.6 %: no gain in double
            return { K_WAVE64, "sdsp_fft1024_wave", 1, p->radix, false, pc };
        if (variant == 0)
            return { K_REG64, "sdsp_fft_reg_f64_kernel", 1, p->radix, false, pc };
    }
    if (p->path == PATH_REG && f32 && variant < 3) {
        // Complex plans: the register-pass family (fft_reg.hip) at every N <= 2048 since its tiles are 2048 points (eight 128-thread
        // workgroups per CU instead of four of 256; round 3): 75.8-78.9 % of HBM peak at N = 16 .. 2048, either radix, in one call,
        // where it measured 68-74 % with 4096-point tiles and the one-wave kernels (fft_wave.hip) 72.5 / 74.7 / 74.2 % at N = 256 /
        // 1024 / 2048 -- those are variant 2 of their sizes now (variant 1: the family with the default cache policy).
        // Real-input plans (the one-wave kernels split / merge by ds_bpermute, the family in LDS): with the 2048-point tiles the family
        // leads at n_real = 512 (74.7 against 71.8 %; radix 4: 75.5 / 71.0) and 2048 (70.5 / 67.2), the one-wave kernel keeps n_real = 1024
        // (70.8 / 69.8) -- tools/rfft_probe.py, one call each; n_real = 4096 .. 65536 radix 2 were taken by K_BIG_REAL above
        const int wave_variant = (p->real_mode && p->n == 512) ? 0 : 2;
        if (variant == wave_variant && fft_wave_supports(p->n, p->radix) && (!p->real_mode || p->radix == 2))
            return { K_WAVE1024, "sdsp_fft1024_wave", 1, p->radix, false, pc };
        if (variant == wave_variant && p->twt_wave && (p->real_mode ? p->n <= 512 : p->n != 512))
            return { K_WAVE2, "sdsp_fft_wave_f32", 1, p->radix, false, pc };
        return { K_REG32, "sdsp_fft_reg_kernel", 1, p->radix, false, pc };
    }
    if (p->real_mode)
        return { K_UNSUPPORTED, "none", 0, p->radix, false, false };
    if (p->path == PATH_TILE || p->path == PATH_FFT4096 || p->path == PATH_REG)
        return { K_TILE, "sdsp_fft_tile_kernel", 1, p->radix, false, pc };
    // ---- everything below is multi-pass and uses the plan's workspace
    if (p->path == PATH_FFT1M && variant < 2) {
        if (variant == 1 || p->ws_batch < kFft1mQueues * kFft1mRing)
            return { K_FFT1M_CHUNKED, "sdsp_fft1m_cols+sdsp_fft1m_rows", 2, 2, true, false };
        return { K_FFT1M_FUSED, "sdsp_fft1m_fused", 2, 2, true, false };
    }
    // variant 2: the same schedule through fft_2pass.hip's generic persistent kernel (its tile functions at 1024 x 1024) -- the A/B that
    // says what the dedicated kernel of fft1m_kernels.h is worth
    if (p->path == PATH_FFT1M && variant == 2 && p->ws_batch >= 4 * kFft1mQueues)
        return { K_2PASS_FUSED, "sdsp_fft2p_fused", 2, 2, true, false };
    const bool two_pass_size = fft_2pass_supports(p->n, p->precision);
    // two schedules over the same tiles (bit-identical results): ONE persistent, ticketed launch (the workspace is a ring of
    // intermediates inside the Infinity Cache; needs a plan whose workspace holds that ring) -- variant 0, level or ahead at every size in
    // one-call A/Bs (fft_2pass.hip, profiles/r03_fft2p_fused_lab.txt) -- and two launches per chunk of 256 MiB, variant 3
    if (p->path == PATH_FOUR_STEP && two_pass_size && (variant == 0 || variant == 3)) {
        if (p->f2_unit && variant == 0)
            return { K_2PASS_FUSED, "sdsp_fft2p_fused", 2, 2, true, false };
        return { K_2PASS, "sdsp_fft2p_cols+sdsp_fft2p_rows", 2, 2, true, false };
    }
    // three streaming passes, N = 16 x N2 with the rows on a tuned single-pass kernel (or, nested, on another plan)
    if (p->path == PATH_FOUR_STEP && p->mid_rows && variant == ((two_pass_size || (big64 && big64_variant(p->n) == 0)) ? 1 : 0)) {
        const fft_kernel_sel rows = select_kernel(p->mid_rows, p->mid_rows->variant);
        // the column step is four radix-2 stages (fft_mid.hip), whatever runs in the rows
        return { K_MID, "sdsp_fft_col16_kernel+rows+sdsp_fft_untwist16", 2 + rows.hbm_passes,
                 rows.stage_radix == 2 ? 2 : SDSP_HIP_STAGES_2_THEN_4, true, false };
    }
    return { K_FOUR_STEP, "sdsp_fft_tile_kernel", 2, p->radix, true, false };
}

// `variant`: the kernel variant to run (normally the plan's; the convolution path overrides it without touching the plan)
// hmul (the two-pass kernels, forward plans): every output leaves multiplied by hmul[k] -- the fused convolution's forward half
int fft_exec_device(sdsp_hip_fft_plan *p, void *data, uint64_t batch, hipStream_t stream, int variant, const void *hmul = nullptr)
{
    if (batch == 0 || p->path == PATH_NOOP)
        return SDSP_HIP_OK;
    const bool rev = p->direction == SDSP_HIP_REVERSE;
    const fft_kernel_sel sel = select_kernel(p, variant);
    if (sel.workspace)
        if (int rc = ensure_workspace(p))
            return rc;

    switch (sel.id) {
    case K_NOOP:
        return SDSP_HIP_OK;
    case K_UNSUPPORTED:
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "real-input plans have no alternative kernel variant");
    case K_FFT4096_R4:
    case K_FFT4096_R2: {
        fft4096_args a;
        a.data = data;
        a.tw = p->twt;
        a.batch = batch;
        a.scale = 1.0f / 4096.0f;
        a.reverse = rev;
        return sel.id == K_FFT4096_R4 ? launch_fft4096_r4_f32(a, variant, stream) : launch_fft4096_r2_f32(a, stream);
    }
    case K_MIX: {
        fft_mix_args a;
        a.data = data;
        a.tw = p->twt_mix;
        a.tw_lead = p->tw_lead;
        a.n = p->n;
        a.batch = batch;
        a.scale = (float)(1.0 / p->n);
        a.reverse = rev;
        return launch_fft_mix_f32(a, stream);
    }
    case K_BIG:
    case K_BIG_REAL: {
        fft_reg_args a;
        a.data = data;
        a.tw = p->twt_big;
        a.n = p->n;
        a.radix = p->radix;
        a.batch = batch;
        a.scale = (float)(1.0 / p->n);
        a.reverse = rev;
        a.nontemporal = 1;
        if (sel.id == K_BIG_REAL) {
            a.real_mode = p->real_mode;
            a.tw2 = p->tw2;
        }
        return launch_fft_big_f32(a, stream);
    }
    case K_BIG64:
    case K_BIG64_REAL: {
        fft_reg_args a;
        a.data = data;
        a.tw = p->twt_big;
        a.n = p->n;
        a.radix = p->radix;
        a.batch = batch;
        a.scale = 1.0f;
        a.scale_d = 1.0 / p->n;
        a.reverse = rev;
        if (sel.id == K_BIG64_REAL) {
            a.real_mode = p->real_mode;
            a.tw2 = p->tw2;
        }
        a.nontemporal = 1;
        return launch_fft_big_f64(a, stream);
    }
    case K_REG64:
    case K_WAVE64: {
        fft_reg_args a;
        a.data = data;
        a.tw = p->twt_reg;
        a.n = p->n;
        a.radix = p->radix;
        a.batch = batch;
        a.scale = 1.0f;
        a.scale_d = 1.0 / p->n;
        a.reverse = rev;
        a.nontemporal = 1;
        a.real_mode = p->real_mode;
        a.tw2 = p->tw2;
        return sel.id == K_WAVE64 ? launch_fft_wave_f64(a, stream) : launch_fft_reg_f64(a, stream);
    }
    case K_WAVE1024:
    case K_WAVE2:
    case K_REG32: {
        fft_reg_args a;
        a.data = data;
        a.tw = sel.id == K_WAVE2 ? p->twt_wave : p->twt_reg;
        a.n = p->n;
        a.radix = p->radix;
        a.batch = batch;
        a.scale = (float)(1.0 / p->n);
        a.reverse = rev;
        a.nontemporal = variant != 1 || (p->tw_lead && !p->real_mode); // variant 1: default cache policy (N < 8192)
        a.real_mode = p->real_mode;
        a.tw2 = p->tw2;
        if (sel.id == K_WAVE1024)
            return launch_fft_wave_f32(a, stream);
        return sel.id == K_WAVE2 ? launch_fft_wave2_f32(a, stream) : launch_fft_reg_f32(a, stream);
    }
    case K_TILE: {
        fft_tile_args a{};
        a.in = data;
        a.out = data;
        a.tw = p->tw;
        a.tw_big = nullptr;
        a.n = p->n;
        a.log2n = sdsp_hip_log2(p->n);
        a.cols = p->cols;
        a.pitch = p->pitch;
        a.total_cols = batch;
        a.tiles_per_group = 1;
        a.group_stride = (uint64_t)p->cols * p->n;
        a.in_tile_step = a.out_tile_step = 0;
        a.in_si = a.out_sk = 1;
        a.in_sc = a.out_sc = p->n;
        a.in_c_fast = a.out_c_fast = 0;
        a.reverse = rev;
        a.apply_scale = rev;
        a.scale = (float)(1.0 / p->n);
        a.scale_d = 1.0 / p->n;
        const uint64_t tiles = (batch + p->cols - 1) / p->cols;
        return launch_fft_tile(p->precision, p->radix, a, tiles, stream);
    }
    default:
        break; // the multi-pass kernels follow
    }

    if (sel.id == K_FFT1M_CHUNKED || sel.id == K_FFT1M_FUSED) {
        const uint64_t N = 1ull << 20;
        const float scale = (float)(1.0 / (double)N);
        if (sel.id == K_FFT1M_CHUNKED) {
            // two launches per chunk of <= 32 transforms (round 1's schedule; also what plans with a small workspace run)
            const uint64_t chunk = fft1m_chunk(p);
            for (uint64_t done = 0; done < batch; done += chunk) {
                fft1m_args a;
                a.data = reinterpret_cast<char *>(data) + done * N * 8;
                a.workspace = p->workspace;
                a.tw_n = p->tw;
                a.tw_1024 = p->tw1;
                a.count = std::min<uint64_t>(chunk, batch - done);
                a.scale = scale;
                a.reverse = rev;
                if (int rc = launch_fft1m_pass(a, 1, stream))
                    return rc;
                if (int rc = launch_fft1m_pass(a, 2, stream))
                    return rc;
            }
            return SDSP_HIP_OK;
        }
        // default: ONE persistent launch (fft1m_kernels.h): eight ticket queues, per queue a ring of three intermediates,
        // pass 2 of a queue's transform one ticket step behind its pass 1
        for (uint64_t done = 0; done < batch; done += p->sync_count) {
            fft1m_fused_args a;
            a.data = reinterpret_cast<char *>(data) + done * N * 8;
            a.workspace = p->workspace;
            a.tw_1024 = p->tw1;
            a.sync = p->sync;
            a.sticky = fft1m_sticky(p);
            a.spin_limit = p->wait_limit;
            a.count = std::min<uint64_t>(p->sync_count, batch - done);
            // a ring of 4 with pass 2 two steps behind measured 42.0-42.2 %, 3 / one step 41.4-41.6 % (profiles/r02_fft1m_lab.md)
            a.ring = p->ws_batch >= 4 * kFft1mQueues ? 4 : (uint32_t)kFft1mRing;
            a.lag = a.ring - 2;
            a.queues = (uint32_t)kFft1mQueues;
            a.scale = scale;
            a.reverse = rev;
            if (int rc = launch_fft1m_fused(a, stream))
                return rc;
        }
        return SDSP_HIP_OK;
    }

    // the two-pass sizes in one persistent launch per kFused2pUnitsPerLaunch units
    if (sel.id == K_2PASS_FUSED) {
        const bool m1 = p->path == PATH_FFT1M; // unit = one transform, the counters of the dedicated kernel
        const uint64_t per_launch = m1 ? p->sync_count : p->sync_count * p->f2_unit;
        for (uint64_t done = 0; done < batch; done += per_launch) {
            fft_2pass_fused_args a;
            a.data = reinterpret_cast<char *>(data) + done * p->n * esize(p->precision);
            a.workspace = p->workspace;
            a.tw_1024 = p->path == PATH_FFT1M ? p->tw1 : p->tw1024; // N = 2^20: n1 = 1024, so W_n1 is that table
            a.sync = p->sync;
            a.sticky = reinterpret_cast<char *>(p->sync) + p->sticky_off;
            a.spin_limit = p->wait_limit;
            a.count = std::min<uint64_t>(per_launch, batch - done);
            a.n = p->n;
            a.unit = m1 ? 1 : p->f2_unit;
            a.ring = m1 ? 4 : p->f2_ring;
            a.lag = m1 ? 2 : p->f2_lag;
            a.queues = m1 ? (uint32_t)kFft1mQueues : p->f2_queues;
            a.scale = (float)(1.0 / p->n);
            a.scale_d = 1.0 / p->n;
            a.reverse = rev;
            a.hmul = hmul;
            if (int rc = launch_fft_2pass_fused(p->precision, a, stream))
                return rc;
        }
        return SDSP_HIP_OK;
    }
    // N = 2^16 .. 2^19, f32: two passes over HBM (fft_2pass.hip), in chunks whose intermediate is at most 256 MiB
    if (sel.id == K_2PASS) {
        const uint64_t chunk = fft2p_chunk(p);
        for (uint64_t done = 0; done < batch; done += chunk) {
            fft_2pass_args a;
            a.data = reinterpret_cast<char *>(data) + done * p->n * esize(p->precision);
            a.workspace = p->workspace;
            a.tw_1024 = p->tw1024;
            a.n = p->n;
            a.count = std::min<uint64_t>(chunk, batch - done);
            a.scale = (float)(1.0 / p->n);
            a.scale_d = 1.0 / p->n;
            a.reverse = rev;
            a.hmul = hmul;
            if (int rc = launch_fft_2pass(p->precision, a, stream))
                return rc;
        }
        return SDSP_HIP_OK;
    }

    // three streaming passes, N = 16 x N2 with the rows on a tuned single-pass kernel: f32 N = 2^21 .. 2^23, f64
    // N = 2^14 .. 2^21, and variant 1 of the f32 sizes above
    if (sel.id == K_MID) {
        const uint32_t n2 = p->n / 16;
        uint64_t done = 0;
        while (done < batch) {
            const uint64_t nb = std::min<uint64_t>(p->ws_batch, batch - done);
            char *d = reinterpret_cast<char *>(data) + done * p->n * esize(p->precision);
            if (int rc = launch_fft_mid_cols(p->precision, d, p->workspace, p->tw1024, n2, nb, rev, stream))
                return rc;
            if (int rc = fft_exec_device(p->mid_rows, p->workspace, nb * 16, stream, p->mid_rows->variant))
                return rc;
            if (int rc = launch_fft_mid_untwist(p->precision, p->workspace, d, n2, nb, stream))
                return rc;
            done += nb;
        }
        return SDSP_HIP_OK;
    }

    // four-step: N = n1 x n2 viewed as a row-major [n1][n2] matrix (index n = n2_count*i1 + i2).
    //   pass 1: length-n1 transforms down the columns, times W_N^(i2*k1), data -> workspace
    //   pass 2: length-n2 transforms along the rows, written transposed, workspace -> data
    const uint64_t N = (uint64_t)p->n1 * p->n2;
    uint64_t done = 0;
    while (done < batch) {
        const uint64_t nb = std::min<uint64_t>(p->ws_batch, batch - done);
        char *d = reinterpret_cast<char *>(data) + done * N * esize(p->precision);
        fft_tile_args a{};
        a.in = d;
        a.out = p->workspace;
        a.tw = p->tw1;
        a.tw_big = p->tw;
        a.n = p->n1;
        a.log2n = sdsp_hip_log2(p->n1);
        a.cols = p->cols1;
        a.pitch = p->pitch1;
        a.tiles_per_group = p->n2 / p->cols1;
        a.total_cols = nb * p->n2;
        a.group_stride = N;
        a.in_tile_step = a.out_tile_step = p->cols1;
        a.in_si = a.out_sk = p->n2;
        a.in_sc = a.out_sc = 1;
        a.in_c_fast = a.out_c_fast = 1;
        a.reverse = rev;
        a.apply_scale = 0;
        a.scale = 1.0f;
        a.scale_d = 1.0;
        int rc = launch_fft_tile(p->precision, p->radix, a, nb * a.tiles_per_group, stream);
        if (rc)
            return rc;

        fft_tile_args c{};
        c.in = p->workspace;
        c.out = d;
        c.tw = p->tw2;
        c.tw_big = nullptr;
        c.n = p->n2;
        c.log2n = sdsp_hip_log2(p->n2);
        c.cols = p->cols2;
        c.pitch = p->pitch2;
        c.tiles_per_group = p->n1 / p->cols2;
        c.total_cols = nb * p->n1;
        c.group_stride = N;
        c.in_tile_step = (uint64_t)p->cols2 * p->n2;
        c.out_tile_step = p->cols2;
        c.in_si = 1;
        c.in_sc = p->n2;
        c.in_c_fast = 0;
        c.out_sk = p->n1;
        c.out_sc = 1;
        c.out_c_fast = 1;
        c.reverse = rev;
        c.apply_scale = rev;
        c.scale = (float)(1.0 / (double)N);
        c.scale_d = 1.0 / (double)N;
        rc = launch_fft_tile(p->precision, p->radix, c, nb * c.tiles_per_group, stream);
        if (rc)
            return rc;
        done += nb;
    }
    return SDSP_HIP_OK;
}

// single-launch paths in pieces (sdsp_hip_set_launch_piece_bytes); the multi-pass paths chunk by their workspace already
// N <= 8192: kernels with many short workgroups, where pieces measured +1 .. +3 points (N = 64: 66.6 -> 69.8 %, 1024:
// 68.7 -> 71.4 %, 4096: 72.1 -> 76.3 %, 8192: 75.4 -> 76.4 %, 4 GiB buffers).  The N = 16384 / 32768 kernels keep one
// or two transforms per CU for tens of microseconds: 62.9 -> 61.3 % and 44.9 -> 43.3 % in pieces, so they stay whole.
uint64_t fft_piece(const sdsp_hip_fft_plan *p, const fft_kernel_sel &sel, uint64_t batch)
{
    if (!sel.pieces)
        return batch;
    const uint64_t row_bytes = (uint64_t)p->n * esize(p->precision); // real-input plans: n = n_real / 2 complex elements
    return piece_units(batch, row_bytes, p->n < 16 ? 4096 : 256);    // a multiple of what one workgroup owns
}

int fft_exec_pieces(sdsp_hip_fft_plan *p, void *data, uint64_t batch, hipStream_t stream, int variant)
{
    const fft_kernel_sel sel = select_kernel(p, variant);
    const uint64_t row_bytes = (uint64_t)p->n * esize(p->precision);
    const uint64_t piece = fft_piece(p, sel, batch);
    for (uint64_t done = 0; done < batch; done += piece) {
        const uint64_t nb = std::min(piece, batch - done);
        if (int rc = fft_exec_device(p, static_cast<char *>(data) + done * row_bytes, nb, stream, variant))
            return rc;
    }
    return SDSP_HIP_OK;
}

// kernel launches one sdsp_hip_fft_exec(plan, data, batch) issues with kernel variant `variant` (memsets not counted)
uint64_t fft_launch_count(const sdsp_hip_fft_plan *p, uint64_t batch, int variant, bool in_pieces = true)
{
    if (batch == 0)
        return 0;
    const fft_kernel_sel sel = select_kernel(p, variant);
    auto ceil_div = [](uint64_t a, uint64_t b) { return (a + b - 1) / b; };
    switch (sel.id) {
    case K_NOOP:
    case K_UNSUPPORTED: return 0;
    case K_FFT1M_CHUNKED: return 2 * ceil_div(batch, fft1m_chunk(p));
    case K_FFT1M_FUSED: return ceil_div(batch, p->sync_count);
    case K_2PASS: return 2 * ceil_div(batch, fft2p_chunk(p));
    case K_2PASS_FUSED: return ceil_div(batch, p->path == PATH_FFT1M ? p->sync_count : p->sync_count * p->f2_unit);
    case K_FOUR_STEP: return 2 * ceil_div(batch, p->ws_batch);
    case K_MID: {
        uint64_t n = 0;
        for (uint64_t done = 0; done < batch; done += p->ws_batch)
            n += 2 + fft_launch_count(p->mid_rows, std::min<uint64_t>(p->ws_batch, batch - done) * 16, p->mid_rows->variant, false);
        return n;
    }
    default: return in_pieces ? ceil_div(batch, fft_piece(p, sel, batch)) : 1;
    }
}

int fft1m_check_sticky(sdsp_hip_fft_plan *p)
{
    if (!p->sync)
        return SDSP_HIP_OK;
    unsigned flag = 0;
    HIP_TRY(hipMemcpy(&flag, fft1m_sticky(p), sizeof(flag), hipMemcpyDeviceToHost));
    if (flag)
        return fail(SDSP_HIP_ERR_HIP, "a persistent two-pass launch gave up on a bounded wait between its passes: the output of the last call is invalid");
    return SDSP_HIP_OK;
}

} // namespace

extern "C" {

// ------------------------------------------------------------------ runtime

int sdsp_hip_device_count(int *count)
{
    if (!count)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "count is null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *count = e == hipSuccess ? c : 0;
    return SDSP_HIP_OK;
}

int sdsp_hip_malloc(void **dev_ptr, size_t bytes, int device)
{
    if (!dev_ptr)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "dev_ptr is null");
    if (int rc = use_device(device))
        return rc;
    HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return SDSP_HIP_OK;
}

int sdsp_hip_free(void *dev_ptr, int device)
{
    if (!dev_ptr)
        return SDSP_HIP_OK;
    if (int rc = use_device(device))
        return rc;
    HIP_TRY(hipFree(dev_ptr));
    return SDSP_HIP_OK;
}

int sdsp_hip_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes, int device)
{
    if (int rc = use_device(device))
        return rc;
    HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return SDSP_HIP_OK;
}

int sdsp_hip_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes, int device)
{
    if (int rc = use_device(device))
        return rc;
    HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return SDSP_HIP_OK;
}

int sdsp_hip_device_synchronize(int device)
{
    if (int rc = use_device(device))
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    return SDSP_HIP_OK;
}

// ------------------------------------------------------------------ FFT plans

int sdsp_hip_fft_plan_create(sdsp_hip_fft_plan **out, uint32_t n, int radix, int direction, int precision,
                             uint64_t max_batch, int device)
{
    if (!out)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan out-pointer is null");
    *out = nullptr;
    // radix 0 (SDSP_HIP_RADIX_AUTO): radix-4 stages where n is a power of 4, radix-2 stages otherwise -- the
    // "mixed" entry of SURVEY 8(f)-4: any power of two without the caller choosing the function
    const bool radix_auto = radix == SDSP_HIP_RADIX_AUTO;
    if (radix_auto) {
        if (!sdsp_hip_is_power_of_2(n))
            return fail(SDSP_HIP_ERR_INVALID_SIZE, "FFT size must be a power of 2!");
        // the stage type of the fastest kernel of the size: radix 4 where n is a power of 4 -- except n = 16384, where
        // fft_big.hip's radix-2 stages measure 69-70 % of HBM peak and its seven radix-4 stages 67-69 %
        radix = (sdsp_hip_is_power_of_4(n) && n != 16384) ? 4 : 2;
    }
    // the reference's static_asserts (fft.h:261, :304) as run-time checks
    if (radix == 2) {
        if (!sdsp_hip_is_power_of_2(n))
            return fail(SDSP_HIP_ERR_INVALID_SIZE, "FFT size must be a power of 2!");
    } else if (radix == 4) {
        if (!sdsp_hip_is_power_of_4(n))
            return fail(SDSP_HIP_ERR_INVALID_SIZE, "FFT radix 4 size must be a power of 4!");
    } else {
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "radix must be 2 or 4");
    }
    if (direction != SDSP_HIP_FORWARD && direction != SDSP_HIP_REVERSE)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "direction must be SDSP_HIP_FORWARD or SDSP_HIP_REVERSE");
    if (precision != SDSP_HIP_F32 && precision != SDSP_HIP_F64)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "precision must be SDSP_HIP_F32 or SDSP_HIP_F64");
    if (n > (1u << 24))
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "FFT sizes above 2^24 are not supported");
    if (int rc = use_device(device))
        return rc;

    auto *p = new sdsp_hip_fft_plan();
    p->n = n;
    p->radix = radix;
    p->direction = direction;
    p->precision = precision;
    p->device = device;
    p->max_batch = max_batch ? max_batch : 1;
    // An explicit radix is the stage type that runs (radix 2: radix-2 butterflies only; radix 4: radix-4 only).  AUTO asks
    // for the fastest kernel: at N = 8192 = 2 * 4^6 that is the radix-4 machinery behind ONE radix-2 stage.
    p->allow_mix = (radix_auto && n == 8192) || radix == 4;

    int rc = SDSP_HIP_OK;
    std::vector<double> w;
    const uint32_t lds_cap_n = (uint32_t)(fft_tile_max_lds_bytes() / esize(precision));
    if (n == 1) {
        p->path = PATH_NOOP;
    } else if (n <= lds_cap_n) {
        if (n == 4096 && radix == 4 && precision == SDSP_HIP_F32)
            p->path = PATH_FFT4096;
        else if (precision == SDSP_HIP_F32 && fft_reg_supports(n, radix))
            p->path = PATH_REG;
        else if (precision == SDSP_HIP_F64 && fft_reg64_supports(n, radix))
            p->path = PATH_REG;
        else
            p->path = PATH_TILE;
        make_twiddles(n, direction, w);
        rc = upload_twiddles(w, precision, &p->tw);
        p->twiddle_bytes = (uint64_t)n * esize(precision);
        if (!rc && n == 4096 && precision == SDSP_HIP_F32)
            rc = upload_thread_twiddles_4096(w, radix, &p->twt);
        if (!rc && ((precision == SDSP_HIP_F32 && fft_reg_supports(n, radix)) ||
                    (precision == SDSP_HIP_F64 && fft_reg64_supports(n, radix))))
            rc = upload_thread_twiddles_reg(w, n, radix, precision, &p->twt_reg);
        if (!rc && precision == SDSP_HIP_F32 && (fft_big_supports(n, radix) || fft_big_conv_supports(n, radix)))
            rc = big_r4_form(n, radix) ? upload_thread_twiddles_big_r4(w, n, SDSP_HIP_F32, &p->twt_big) : upload_thread_twiddles_big(w, n, SDSP_HIP_F32, &p->twt_big);
        if (!rc && precision == SDSP_HIP_F64 && fft_big64_supports(n, radix))
            rc = radix == 4 ? upload_thread_twiddles_big_r4(w, n, SDSP_HIP_F64, &p->twt_big) : upload_thread_twiddles_big(w, n, SDSP_HIP_F64, &p->twt_big);
        if (!rc && precision == SDSP_HIP_F32 && fft_wave2_supports(n, radix))
            rc = upload_thread_twiddles_wave(w, n, radix, &p->twt_wave);
        if (!rc && precision == SDSP_HIP_F32 && fft_mix_supports(n) && p->allow_mix)
            rc = upload_thread_twiddles_mix(w, n, &p->twt_mix, &p->tw_lead);
        pick_tile(precision, n, std::max<uint32_t>(1, 1024 / n), &p->cols, &p->pitch);
        if (p->cols > 16)
            pick_tile(precision, n, 16, &p->cols, &p->pitch);
    } else {
        // N = 2^20 f32 runs the tuned two-pass kernels for either radix: a radix-4 DIF stage (fft.h:311-349) is two fused
        // radix-2 stages, so the 10 + 10 radix-2 stages of fft1m.hip are the same dataflow as ten radix-4 stages (as for
        // N = 16384 in fft_big.hip); variants >= 8 of such a plan still run genuine radix-4 stages (coverage kernel)
        p->path = (n == (1u << 20) && precision == SDSP_HIP_F32) ? PATH_FFT1M : PATH_FOUR_STEP;
        const uint32_t k = sdsp_hip_log2(n);
        if (radix == 2) {
            p->n1 = 1u << ((k + 1) / 2);
        } else {
            const uint32_t d = k / 2;
            p->n1 = 1u << (2 * ((d + 1) / 2));
        }
        p->n2 = n / p->n1;
        make_twiddles(n, direction, w);
        rc = upload_twiddles(w, precision, &p->tw);
        if (!rc && precision == SDSP_HIP_F32 && fft_big_supports(n, radix))
            rc = upload_thread_twiddles_big(w, n, SDSP_HIP_F32, &p->twt_big);
        if (!rc && precision == SDSP_HIP_F64 && fft_big64_supports(n, radix))
            rc = radix == 4 ? upload_thread_twiddles_big_r4(w, n, SDSP_HIP_F64, &p->twt_big) : upload_thread_twiddles_big(w, n, SDSP_HIP_F64, &p->twt_big);
        if (!rc) {
            make_twiddles(p->n1, direction, w);
            rc = upload_twiddles(w, precision, &p->tw1);
        }
        if (!rc) {
            make_twiddles(p->n2, direction, w);
            rc = upload_twiddles(w, precision, &p->tw2);
        }
        p->twiddle_bytes = ((uint64_t)n + p->n1 + p->n2) * esize(precision);
        pick_tile(precision, p->n1, 16, &p->cols1, &p->pitch1);
        pick_tile(precision, p->n2, 16, &p->cols2, &p->pitch2);
        // the tuned 2^20 path keeps 24 intermediates for the persistent kernel (8 queues x 3) / 32 for variant 1's chunks
        p->ws_batch = p->path == PATH_FFT1M ? std::min<uint64_t>(p->max_batch, 32) : p->max_batch;
        // the two-pass sizes never touch more than 256 MiB of intermediate at a time (a chunk of the two launches, the persistent
        // launch's ring): their workspace stops there instead of growing with max_batch (the alternates run in slices of it)
        if (p->path == PATH_FOUR_STEP && fft_2pass_supports(n, precision))
            p->ws_batch = std::min<uint64_t>(p->max_batch, std::max<uint64_t>(1, (1ull << 28) / ((uint64_t)n * esize(precision))));
        p->workspace_bytes = p->ws_batch * n * esize(precision); // allocated by the first exec that needs it
        if (!rc && p->path == PATH_FFT1M) {
            p->sync_count = std::min<uint64_t>(p->max_batch, kFft1mPerLaunch);
            // + one line behind the per-launch block for the sticky abort word (never touched by the per-launch memset)
            p->sticky_off = fft1m_sync_bytes(p->sync_count, (uint32_t)kFft1mQueues);
            const size_t sync_bytes = p->sticky_off + 64;
            hipError_t e = hipMalloc(&p->sync, sync_bytes);
            if (e == hipSuccess) // the abort words must read 0 before the first persistent launch (sdsp_hip_fft_plan_status)
                e = hipMemset(p->sync, 0, sync_bytes);
            if (e != hipSuccess)
                rc = fail(SDSP_HIP_ERR_NOMEM, std::string("fft1m counters hipMalloc: ") + hipGetErrorString(e));
        }
    }
    // three-pass schedule (fft_mid.hip): f32 N = 2^16 .. 2^23 (2^20 is PATH_FFT1M); f64 N = 2^14 .. 2^21 -- the rows
    // then land on the f64 register-pass family (N <= 8192) or, nested, on another three-pass plan
    if (!rc && p->path == PATH_FOUR_STEP &&
        ((precision == SDSP_HIP_F32 && n >= (1u << 16) && n <= (1u << 23)) ||
         (precision == SDSP_HIP_F64 && n >= (1u << 14) && n <= (1u << 21)))) {
        const uint32_t n2 = n / 16;
        const int sub_radix = (radix == 4 && sdsp_hip_is_power_of_4(n2)) ? 4 : 2;
        rc = sdsp_hip_fft_plan_create(&p->mid_rows, n2, sub_radix, direction, precision, p->ws_batch * 16, device);
        if (!rc) {
            make_twiddles(1024, direction, w);
            rc = upload_twiddles(w, precision, &p->tw1024);
        }
    }
    // the two-pass sizes of fft_2pass.hip: counters of the persistent schedule, where the workspace holds its ring of intermediates
    if (!rc && p->path == PATH_FOUR_STEP && fft_2pass_supports(n, precision)) {
        uint32_t unit, queues, ring, lag;
        fft_2pass_fused_shape(n, precision, &unit, &queues, &ring, &lag);
        if (unit && p->ws_batch >= (uint64_t)unit * queues * ring) {
            p->sync_count = std::min<uint64_t>((p->max_batch + unit - 1) / unit, kFused2pUnitsPerLaunch);
            p->sticky_off = fft_2pass_sync_bytes(p->sync_count, queues);
            hipError_t e = hipMalloc(&p->sync, p->sticky_off + 64);
            if (e == hipSuccess)
                e = hipMemset(p->sync, 0, p->sticky_off + 64);
            if (e != hipSuccess)
                rc = fail(SDSP_HIP_ERR_NOMEM, std::string("fft_2pass counters hipMalloc: ") + hipGetErrorString(e));
            else {
                p->f2_unit = unit;
                p->f2_queues = queues;
                p->f2_ring = ring;
                p->f2_lag = lag;
            }
        }
    }
    // plans whose DEFAULT kernel is multi-pass own their workspace from here on (no allocation on the launch path: stream
    // capture works, out-of-memory is a create-time error); single-pass defaults with multi-pass alternates stay lazy
    if (!rc && select_kernel(p, 0).workspace)
        rc = ensure_workspace(p);
    if (rc) {
        sdsp_hip_fft_plan_destroy(p);
        return rc;
    }
    *out = p;
    return SDSP_HIP_OK;
}

int sdsp_hip_rfft_plan_create_p(sdsp_hip_fft_plan **out, uint32_t n_real, int radix, int direction, int precision,
                                uint64_t max_batch, int device)
{
    if (!out)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan out-pointer is null");
    *out = nullptr;
    if (!sdsp_hip_is_power_of_2(n_real) || n_real < 32)
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "FFT size must be a power of 2! (real-input plans: >= 32)");
    if (precision != SDSP_HIP_F32 && precision != SDSP_HIP_F64)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "precision must be SDSP_HIP_F32 or SDSP_HIP_F64");
    const uint32_t n = n_real / 2;
    if (radix == 4 && !sdsp_hip_is_power_of_4(n))
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "FFT radix 4 size must be a power of 4! (n_real / 2)");
    if (radix != 2 && radix != 4)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "radix must be 2 or 4");
    const bool big_real = (precision == SDSP_HIP_F32 && fft_big_real_supports(n, radix)) ||   // fft_big.hip, REAL
                          (precision == SDSP_HIP_F64 && fft_big64_real_supports(n, radix));    // fft_big64.hip, REAL
    if (!big_real && (precision == SDSP_HIP_F32 ? !fft_reg_supports(n, radix) : !fft_reg64_supports(n, radix)))
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "real-input plans cover n_real = 32 .. 32768 (f32; radix 2: .. 65536) / 32 .. 16384 (f64; radix 2: .. 32768)");
    sdsp_hip_fft_plan *p = nullptr;
    if (int rc = sdsp_hip_fft_plan_create(&p, n, radix, direction, precision, max_batch, device))
        return rc;
    p->path = PATH_REG; // also at n = 4096 f32 (the tuned complex kernels have no split stage)
    p->real_mode = direction == SDSP_HIP_FORWARD ? 1 : 2;
    std::vector<double> w;
    if (big_real && !p->twt_big) { // (every complex plan these kernels serve has the table already)
        make_twiddles(n, direction, w);
        if (int rc = (precision == SDSP_HIP_F32 && big_r4_form(n, radix)) ? upload_thread_twiddles_big_r4(w, n, SDSP_HIP_F32, &p->twt_big)
                                                                          : upload_thread_twiddles_big(w, n, precision, &p->twt_big)) {
            sdsp_hip_fft_plan_destroy(p);
            return rc;
        }
    }
    make_twiddles(n_real, direction, w);
    if (int rc = upload_twiddles(w, precision, &p->tw2)) {
        sdsp_hip_fft_plan_destroy(p);
        return rc;
    }
    p->twiddle_bytes += (uint64_t)n_real * esize(precision);
    *out = p;
    return SDSP_HIP_OK;
}

int sdsp_hip_rfft_plan_create(sdsp_hip_fft_plan **out, uint32_t n_real, int radix, int direction, uint64_t max_batch,
                              int device)
{
    return sdsp_hip_rfft_plan_create_p(out, n_real, radix, direction, SDSP_HIP_F32, max_batch, device);
}

int sdsp_hip_fft_plan_destroy(sdsp_hip_fft_plan *p)
{
    if (!p)
        return SDSP_HIP_OK;
    if (hipSetDevice(p->device) == hipSuccess) {
        (void)hipFree(p->tw);
        (void)hipFree(p->tw1);
        (void)hipFree(p->tw2);
        (void)hipFree(p->twt);
        (void)hipFree(p->twt_reg);
        (void)hipFree(p->twt_big);
        (void)hipFree(p->twt_wave);
        (void)hipFree(p->twt_mix);
        (void)hipFree(p->tw_lead);
        (void)hipFree(p->tw1024);
        (void)hipFree(p->workspace);
        (void)hipFree(p->host_stage);
        if (p->partner)
            sdsp_hip_fft_plan_destroy(p->partner);
        p->partner = nullptr;
        if (p->mid_rows)
            sdsp_hip_fft_plan_destroy(p->mid_rows);
        p->mid_rows = nullptr;
        (void)hipSetDevice(p->device);
        (void)hipFree(p->sync);
    }
    delete p;
    return SDSP_HIP_OK;
}

int sdsp_hip_fft_exec(sdsp_hip_fft_plan *p, void *data, uint64_t batch, void *stream)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    if (batch == 0)
        return SDSP_HIP_OK;
    if (!data)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data is null");
    if ((uintptr_t)data % esize(p->precision) != 0)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data must be aligned to one complex element");
    if (int rc = use_device(p->device))
        return rc;
    if (p->sync) // this call's launches report into a clean sticky abort word
        HIP_TRY(hipMemsetAsync(fft1m_sticky(p), 0, sizeof(unsigned), reinterpret_cast<hipStream_t>(stream)));
    return fft_exec_pieces(p, data, batch, reinterpret_cast<hipStream_t>(stream), p->variant);
}

int sdsp_hip_fft_exec_host(sdsp_hip_fft_plan *p, void *host_data, uint64_t batch)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    if (batch == 0)
        return SDSP_HIP_OK;
    if (!host_data)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data is null");
    if (int rc = use_device(p->device))
        return rc;
    const uint64_t bytes = batch * p->n * esize(p->precision);
    if (bytes > p->host_stage_bytes) {
        (void)hipFree(p->host_stage);
        p->host_stage = nullptr;
        p->host_stage_bytes = 0;
        hipError_t e = hipMalloc(&p->host_stage, bytes);
        if (e != hipSuccess)
            return fail(SDSP_HIP_ERR_NOMEM, std::string("staging hipMalloc: ") + hipGetErrorString(e));
        p->host_stage_bytes = bytes;
    }
    HIP_TRY(hipMemcpy(p->host_stage, host_data, bytes, hipMemcpyHostToDevice));
    if (p->sync)
        HIP_TRY(hipMemsetAsync(fft1m_sticky(p), 0, sizeof(unsigned), nullptr));
    if (int rc = fft_exec_device(p, p->host_stage, batch, nullptr, p->variant))
        return rc;
    HIP_TRY(hipMemcpy(host_data, p->host_stage, bytes, hipMemcpyDeviceToHost));
    return fft1m_check_sticky(p); // synchronous path: a launch that gave up is reported here, not only by _plan_status
}

int sdsp_hip_fft_exec_sharded(sdsp_hip_fft_plan *const *plans, int n_plans, void *host_data, uint64_t batch)
{
    if (!plans || n_plans <= 0)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "no plans");
    for (int i = 0; i < n_plans; i++) {
        if (!plans[i])
            return fail(SDSP_HIP_ERR_INVALID_ARG, "null plan in shard list");
        if (plans[i]->n != plans[0]->n || plans[i]->radix != plans[0]->radix ||
            plans[i]->direction != plans[0]->direction || plans[i]->precision != plans[0]->precision)
            return fail(SDSP_HIP_ERR_INVALID_ARG, "shard plans must describe the same transform");
    }
    if (batch == 0)
        return SDSP_HIP_OK;
    if (!host_data)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data is null");
    // contiguous ranges [g*B/G, (g+1)*B/G): independent transforms, no exchange step
    const size_t per = (size_t)plans[0]->n * esize(plans[0]->precision);
    std::vector<int> rcs(n_plans, 0);
    std::vector<std::string> errs(n_plans);
    std::vector<std::thread> th;
    for (int g = 0; g < n_plans; g++) {
        const uint64_t lo = batch * g / n_plans, hi = batch * (g + 1) / n_plans;
        th.emplace_back([=, &rcs, &errs] {
            rcs[g] = sdsp_hip_fft_exec_host(plans[g], reinterpret_cast<char *>(host_data) + lo * per, hi - lo);
            if (rcs[g])
                errs[g] = g_last_error;
        });
    }
    for (auto &t : th)
        t.join();
    for (int g = 0; g < n_plans; g++)
        if (rcs[g])
            return fail(rcs[g], errs[g]);
    return SDSP_HIP_OK;
}

static int convolve_device(sdsp_hip_fft_plan *p, void *data, const void *h, uint64_t batch, void *stream);

int sdsp_hip_fft_convolve(sdsp_hip_fft_plan *p, void *data, const void *h, uint64_t batch, void *stream)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    if (p->direction != SDSP_HIP_FORWARD)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "convolve needs a forward plan");
    if (batch == 0)
        return SDSP_HIP_OK;
    if (!data || !h)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null pointer");
    if (p->real_mode) // a real-input plan's transform is not the complex DFT the product is defined on
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "convolve needs a complex plan (real-input plans are not supported)");
    if (int rc = use_device(p->device))
        return rc;
    // this call's launches report into clean sticky abort words (the persistent two-pass kernels; forward and reverse half)
    if (p->sync)
        HIP_TRY(hipMemsetAsync(fft1m_sticky(p), 0, sizeof(unsigned), reinterpret_cast<hipStream_t>(stream)));
    if (p->partner && p->partner->sync)
        HIP_TRY(hipMemsetAsync(fft1m_sticky(p->partner), 0, sizeof(unsigned), reinterpret_cast<hipStream_t>(stream)));
    // launch pieces as in sdsp_hip_fft_exec (N <= 8192, single-launch kernels); the three-launch composition runs piece by piece
    const bool single = (p->path == PATH_FFT4096 || p->path == PATH_REG || p->path == PATH_TILE) && p->n <= 8192;
    const uint64_t row_bytes = (uint64_t)p->n * esize(p->precision);
    const uint64_t piece = single ? piece_units(batch, row_bytes, p->n < 16 ? 4096 : 256) : batch;
    for (uint64_t done = 0; done < batch; done += piece)
        if (int rc = convolve_device(p, static_cast<char *>(data) + done * row_bytes, h, std::min(piece, batch - done), stream))
            return rc;
    return SDSP_HIP_OK;
}

static int convolve_device(sdsp_hip_fft_plan *p, void *data, const void *h, uint64_t batch, void *stream)
{
    if (p->path == PATH_FFT4096 && p->variant == 0)
        return launch_fft4096_conv_f32(data, p->twt, h, batch, stream);
    // N = 8192 / 16384 / 32768 radix-2 stages, N = 16384 radix-4 stages: both transforms and the multiply in the
    // registers-resident kernel (fft_big.hip)
    if ((p->path == PATH_REG || p->path == PATH_FOUR_STEP) && p->precision == SDSP_HIP_F32 && p->variant == 0 && !p->real_mode &&
        p->twt_big && fft_big_conv_supports(p->n, p->radix)) {
        fft_reg_args a;
        a.data = data;
        a.tw = p->twt_big;
        a.n = p->n;
        a.radix = p->radix;
        a.batch = batch;
        a.scale = (float)(1.0 / p->n);
        a.reverse = 0;
        a.nontemporal = 1;
        a.real_mode = 3;
        a.tw2 = h;
        return launch_fft_big_f32(a, stream);
    }
    // variant 0: the fused kernel of the size; variant 2: the register-pass family's fused MODE 3 where a one-wave kernel
    // is the default (A/B and cross-check); any other variant: three launches
    if (p->path == PATH_REG && p->precision == SDSP_HIP_F32 && (p->variant == 0 || p->variant == 2) && !p->real_mode) {
        fft_reg_args a;
        a.data = data;
        a.tw = p->twt_reg;
        a.n = p->n;
        a.radix = p->radix;
        a.batch = batch;
        a.scale = (float)(1.0 / p->n);
        a.reverse = 0;
        a.nontemporal = 1;
        a.real_mode = 3; // fused convolution: one kernel, h travels in tw2
        a.tw2 = h;
        if (p->variant == 0 && fft_wave_supports(p->n, p->radix)) // N = 1024: both transforms in one wave's registers (fft_wave.hip)
            return launch_fft_wave_f32(a, stream);
        if (p->variant == 0 && p->twt_wave) { // N = 256 / 512 radix 2 (N = 2048 radix 2 was taken by the fft_big.hip form above: 65-69 % against 52 %)
            a.tw = p->twt_wave;
            return launch_fft_wave2_f32(a, stream);
        }
        return launch_fft_reg_f32(a, stream);
    }
    // double, N = 4096 / 8192 / 16384 radix-2 plans: both transforms and the multiply in the registers-resident kernel (fft_big64.hip);
    // variant 2: what served N <= 8192 before (the register-pass family's fused MODE 3), any other variant: three launches
    if ((p->path == PATH_REG || p->path == PATH_FOUR_STEP) && p->precision == SDSP_HIP_F64 && p->variant == 0 && !p->real_mode && p->twt_big &&
        fft_big64_conv_supports(p->n, p->radix)) {
        fft_reg_args a;
        a.data = data;
        a.tw = p->twt_big;
        a.n = p->n;
        a.radix = p->radix;
        a.batch = batch;
        a.scale = 1.0f;
        a.scale_d = 1.0 / p->n;
        a.reverse = 0;
        a.nontemporal = 1;
        a.real_mode = 3;
        a.tw2 = h;
        return launch_fft_big_f64(a, stream);
    }
    if (p->path == PATH_REG && p->precision == SDSP_HIP_F64 && (p->variant == 0 || p->variant == 2) && !p->real_mode) { // f64, N = 16 .. 8192
        fft_reg_args a;
        a.data = data;
        a.tw = p->twt_reg;
        a.n = p->n;
        a.radix = p->radix;
        a.batch = batch;
        a.scale = 1.0f;
        a.scale_d = 1.0 / p->n;
        a.reverse = 0;
        a.nontemporal = 1;
        a.real_mode = 3;
        a.tw2 = h;
        return launch_fft_reg_f64(a, stream);
    }
    if (!p->partner) {
        if (int rc = sdsp_hip_fft_plan_create(&p->partner, p->n, p->radix, SDSP_HIP_REVERSE, p->precision,
                                              p->max_batch, p->device))
            return rc;
        p->partner->wait_limit = p->wait_limit;
    }
    // PATH_FFT4096 with variant != 0 selects this three-launch path for cross-checking: its transforms run variant 0
    int v = p->path == PATH_FFT4096 ? 0 : p->variant;
    // the two-pass sizes: the forward transform's pass 2 multiplies by h on its way out (fft_2pass.hip, HM) -- four passes over HBM
    // instead of five.  N = 2^20 f32 takes the generic persistent kernel (its variant 2) for that half
    if (p->path == PATH_FFT1M && v == 0 && select_kernel(p, 2).id == K_2PASS_FUSED)
        v = 2;
    const fft_kernel_id fwd_id = select_kernel(p, v).id;
    const bool fused_mul = fwd_id == K_2PASS || fwd_id == K_2PASS_FUSED;
    int rc = fft_exec_device(p, data, batch, reinterpret_cast<hipStream_t>(stream), v, fused_mul ? h : nullptr);
    if (!rc && !fused_mul)
        rc = launch_pointwise_mul(p->precision, data, h, p->n, batch, stream);
    if (!rc)
        rc = fft_exec_device(p->partner, data, batch, reinterpret_cast<hipStream_t>(stream), p->partner->variant);
    return rc;
}

int sdsp_hip_fft_plan_get_info(const sdsp_hip_fft_plan *p, sdsp_hip_fft_plan_info *info)
{
    if (!p || !info)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null argument");
    std::memset(info, 0, sizeof(*info));
    info->n = p->n;
    info->radix = p->radix;
    info->direction = p->direction;
    info->precision = p->precision;
    info->device = p->device;
    const fft_kernel_sel sel = select_kernel(p, p->variant); // the same table sdsp_hip_fft_exec dispatches on
    info->hbm_passes = sel.hbm_passes;
    info->stage_radix = sel.stage_radix;
    info->algorithmic_bytes = 2ull * p->n * esize(p->precision); // real plans: n complex = n_real floats, same bytes
    info->workspace_bytes = p->workspace_bytes;
    info->twiddle_bytes = p->twiddle_bytes;
    const char *name = sel.name;
    std::strncpy(info->kernel, name, sizeof(info->kernel) - 1);
    return SDSP_HIP_OK;
}

int sdsp_hip_fft_plan_status(sdsp_hip_fft_plan *p)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    if (int rc = use_device(p->device))
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (int rc = fft1m_check_sticky(p))
        return rc;
    // sdsp_hip_fft_convolve runs its reverse half on the plan's partner: a hand-off lost there belongs to this plan's last call too
    return p->partner ? fft1m_check_sticky(p->partner) : SDSP_HIP_OK;
}

int sdsp_hip_fft_plan_set_wait_limit(sdsp_hip_fft_plan *p, uint64_t ticks)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    p->wait_limit = ticks;
    if (p->partner)
        p->partner->wait_limit = ticks;
    return SDSP_HIP_OK;
}

int sdsp_hip_fft_plan_launches(const sdsp_hip_fft_plan *p, uint64_t batch, uint64_t *launches)
{
    if (!p || !launches)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null argument");
    *launches = fft_launch_count(p, batch, p->variant);
    return SDSP_HIP_OK;
}

int sdsp_hip_set_launch_piece_bytes(uint64_t bytes)
{
    g_piece_bytes.store(bytes, std::memory_order_relaxed);
    return SDSP_HIP_OK;
}

int sdsp_hip_get_launch_piece_bytes(uint64_t *bytes)
{
    if (!bytes)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "bytes is null");
    *bytes = g_piece_bytes.load(std::memory_order_relaxed);
    return SDSP_HIP_OK;
}

int sdsp_hip_fft_plan_get_twiddles(const sdsp_hip_fft_plan *p, void *host_out)
{
    if (!p || !host_out)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null argument");
    if (!p->tw)
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "plan has no twiddle table");
    if (int rc = use_device(p->device))
        return rc;
    HIP_TRY(hipMemcpy(host_out, p->tw, (size_t)p->n * esize(p->precision), hipMemcpyDeviceToHost));
    return SDSP_HIP_OK;
}

int sdsp_hip_fft_plan_set_variant(sdsp_hip_fft_plan *p, int variant)
{
    if (!p || variant < 0)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "bad argument");
    p->variant = variant;
    return SDSP_HIP_OK;
}

// ------------------------------------------------------------------ IIR banks

int sdsp_hip_iir_plan_create(sdsp_hip_iir_plan **out, uint32_t sections, int kind, const double *a,
                             const double *b, double gain, int precision, int device)
{
    if (!out)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan out-pointer is null");
    *out = nullptr;
    if (sections == 0 || sections % 2 != 0) // static_assert casc_2o_iir.h:25
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "M must be even!");
    if (sections > SDSP_HIP_MAX_SECTIONS) // 2 .. 8: the tuned kernels; 10 .. 16: the direct kernel
        return fail(SDSP_HIP_ERR_UNSUPPORTED, "at most SDSP_HIP_MAX_SECTIONS (16) sections are compiled in");
    if (kind < SDSP_HIP_IIR_GENERIC || kind > SDSP_HIP_IIR_BP)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "unknown IIR kind");
    if (!a || (kind == SDSP_HIP_IIR_GENERIC && !b))
        return fail(SDSP_HIP_ERR_INVALID_ARG, "coefficient pointer is null");
    if (precision != SDSP_HIP_F32 && precision != SDSP_HIP_F64 && precision != SDSP_HIP_F32_F64STATE)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "precision must be SDSP_HIP_F32, SDSP_HIP_F64 or SDSP_HIP_F32_F64STATE");
    if (int rc = use_device(device))
        return rc;
    auto *p = new sdsp_hip_iir_plan();
    p->sections = sections;
    p->kind = kind;
    p->precision = precision;
    p->device = device;
    p->gain = gain;
    std::memcpy(p->a, a, sizeof(double) * 3 * sections);
    if (b)
        std::memcpy(p->b, b, sizeof(double) * 3 * sections);
    *out = p;
    return SDSP_HIP_OK;
}

int sdsp_hip_iir_plan_destroy(sdsp_hip_iir_plan *p)
{
    delete p;
    return SDSP_HIP_OK;
}

int sdsp_hip_iir_state_bytes(const sdsp_hip_iir_plan *p, uint64_t channels, uint64_t *bytes)
{
    if (!p || !bytes)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null argument");
    *bytes = 3ull * (p->sections + 1) * channels * (p->precision == SDSP_HIP_F32 ? 4 : 8);
    return SDSP_HIP_OK;
}

int sdsp_hip_iir_plan_set_variant(sdsp_hip_iir_plan *p, int variant)
{
    if (!p || variant < 0)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "bad argument");
    p->variant = variant;
    return SDSP_HIP_OK;
}

int sdsp_hip_iir_process(sdsp_hip_iir_plan *p, void *data, uint64_t channels, uint64_t samples, uint64_t stride,
                         void *state, void *stream)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    if (channels == 0 || samples == 0)
        return SDSP_HIP_OK;
    if (!data)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data is null");
    if (stride < samples && channels > 1)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "stride must be >= samples");
    if (int rc = use_device(p->device))
        return rc;
    iir_args a{};
    a.data = data;
    a.state = state;
    a.channels = channels;
    a.samples = samples;
    a.stride = stride;
    a.sections = p->sections;
    a.kind = p->kind;
    a.gain = p->gain;
    for (uint32_t j = 0; j < p->sections; j++) {
        a.a1[j] = p->a[3 * j + 1];
        a.a2[j] = p->a[3 * j + 2];
        a.b1[j] = p->b[3 * j + 1];
        a.b2[j] = p->b[3 * j + 2];
    }
    return launch_iir(p->precision, a, p->variant, stream);
}

int sdsp_hip_iir_plan_kernel(const sdsp_hip_iir_plan *p, const void *data, uint64_t channels, uint64_t samples, uint64_t stride,
                             char *name, size_t name_bytes)
{
    if (!p || !name || name_bytes == 0)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null argument");
    iir_args a{};
    a.data = const_cast<void *>(data);
    a.channels = channels;
    a.samples = samples;
    a.stride = stride;
    a.sections = p->sections;
    a.kind = p->kind;
    std::strncpy(name, iir_kernel_for(p->precision, a, p->variant), name_bytes - 1);
    name[name_bytes - 1] = 0;
    return SDSP_HIP_OK;
}

int sdsp_hip_iir_process_interleaved(sdsp_hip_iir_plan *p, void *data, uint64_t channels, uint64_t samples,
                                     uint64_t stride, void *state, void *stream)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    if (channels == 0 || samples == 0)
        return SDSP_HIP_OK;
    if (!data)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data is null");
    if (stride < channels && samples > 1)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "stride must be >= channels");
    if (int rc = use_device(p->device))
        return rc;
    iir_args a{};
    a.data = data;
    a.state = state;
    a.channels = channels;
    a.samples = samples;
    a.stride = stride;
    a.sections = p->sections;
    a.kind = p->kind;
    a.gain = p->gain;
    for (uint32_t j = 0; j < p->sections; j++) {
        a.a1[j] = p->a[3 * j + 1];
        a.a2[j] = p->a[3 * j + 2];
        a.b1[j] = p->b[3 * j + 1];
        a.b2[j] = p->b[3 * j + 2];
    }
    return launch_iir_interleaved(p->precision, a, p->variant, stream);
}

int sdsp_hip_iir_process_host(sdsp_hip_iir_plan *p, void *host_data, uint64_t channels, uint64_t samples,
                              uint64_t stride, void *host_state)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    if (channels == 0 || samples == 0)
        return SDSP_HIP_OK;
    if (!host_data)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data is null");
    if (int rc = use_device(p->device))
        return rc;
    const size_t rs = p->precision == SDSP_HIP_F64 ? 8 : 4; // sample size (the mixed mode stores floats)
    const size_t data_bytes = ((channels - 1) * stride + samples) * rs;
    uint64_t state_bytes = 0;
    sdsp_hip_iir_state_bytes(p, channels, &state_bytes);
    void *d = nullptr, *s = nullptr;
    HIP_TRY(hipMalloc(&d, data_bytes));
    int rc = SDSP_HIP_OK;
    hipError_t e = hipMemcpy(d, host_data, data_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && host_state) {
        e = hipMalloc(&s, state_bytes);
        if (e == hipSuccess)
            e = hipMemcpy(s, host_state, state_bytes, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess)
        rc = hip_fail(e, "iir host staging");
    if (!rc)
        rc = sdsp_hip_iir_process(p, d, channels, samples, stride, s, nullptr);
    if (!rc) {
        e = hipMemcpy(host_data, d, data_bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess && host_state)
            e = hipMemcpy(host_state, s, state_bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            rc = hip_fail(e, "iir host read-back");
    }
    (void)hipFree(d);
    (void)hipFree(s);
    return rc;
}

int sdsp_hip_iir_process_sharded(sdsp_hip_iir_plan *const *plans, int n_plans, void *host_data, uint64_t channels,
                                 uint64_t samples)
{
    if (!plans || n_plans <= 0)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "no plans");
    for (int i = 0; i < n_plans; i++)
        if (!plans[i] || plans[i]->precision != plans[0]->precision || plans[i]->sections != plans[0]->sections)
            return fail(SDSP_HIP_ERR_INVALID_ARG, "shard plans must describe the same filter bank");
    if (channels == 0 || samples == 0)
        return SDSP_HIP_OK;
    if (!host_data)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data is null");
    const size_t row = samples * (plans[0]->precision == SDSP_HIP_F64 ? 8 : 4);
    std::vector<int> rcs(n_plans, 0);
    std::vector<std::string> errs(n_plans);
    std::vector<std::thread> th;
    for (int g = 0; g < n_plans; g++) {
        const uint64_t lo = channels * g / n_plans, hi = channels * (g + 1) / n_plans;
        th.emplace_back([=, &rcs, &errs] {
            rcs[g] = sdsp_hip_iir_process_host(plans[g], reinterpret_cast<char *>(host_data) + lo * row, hi - lo,
                                               samples, samples, nullptr);
            if (rcs[g])
                errs[g] = g_last_error;
        });
    }
    for (auto &t : th)
        t.join();
    for (int g = 0; g < n_plans; g++)
        if (rcs[g])
            return fail(rcs[g], errs[g]);
    return SDSP_HIP_OK;
}
// ------------------------------------------------------------------ FIR banks

int sdsp_hip_fir_plan_create(sdsp_hip_fir_plan **out, uint32_t taps, const double *h, int precision, int device)
{
    if (!out)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan out-pointer is null");
    *out = nullptr;
    if (taps == 0 || taps > SDSP_HIP_FIR_MAX_TAPS)
        return fail(SDSP_HIP_ERR_INVALID_SIZE, "taps must be in [1, SDSP_HIP_FIR_MAX_TAPS]");
    if (!h)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "coefficient pointer is null");
    if (precision != SDSP_HIP_F32 && precision != SDSP_HIP_F64)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "precision must be SDSP_HIP_F32 or SDSP_HIP_F64");
    if (int rc = use_device(device))
        return rc;
    auto *p = new sdsp_hip_fir_plan();
    p->taps = taps;
    p->precision = precision;
    p->device = device;
    hipError_t e;
    if (precision == SDSP_HIP_F64) {
        e = hipMalloc(&p->h_dev, taps * sizeof(double));
        if (e == hipSuccess)
            e = hipMemcpy(p->h_dev, h, taps * sizeof(double), hipMemcpyHostToDevice);
    } else {
        std::vector<float> hf(h, h + taps);
        e = hipMalloc(&p->h_dev, taps * sizeof(float));
        if (e == hipSuccess)
            e = hipMemcpy(p->h_dev, hf.data(), taps * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        (void)hipFree(p->h_dev);
        delete p;
        return hip_fail(e, "fir coefficients");
    }
    *out = p;
    return SDSP_HIP_OK;
}

int sdsp_hip_fir_plan_destroy(sdsp_hip_fir_plan *p)
{
    if (!p)
        return SDSP_HIP_OK;
    if (p->h_dev && use_device(p->device) == SDSP_HIP_OK)
        (void)hipFree(p->h_dev);
    delete p;
    return SDSP_HIP_OK;
}

int sdsp_hip_fir_state_bytes(const sdsp_hip_fir_plan *p, uint64_t channels, uint64_t *bytes)
{
    if (!p || !bytes)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "null argument");
    *bytes = static_cast<uint64_t>(p->taps - 1) * channels * (p->precision == SDSP_HIP_F64 ? 8 : 4);
    return SDSP_HIP_OK;
}

int sdsp_hip_fir_plan_set_variant(sdsp_hip_fir_plan *p, int variant)
{
    if (!p || variant < 0)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "bad argument");
    p->variant = variant;
    return SDSP_HIP_OK;
}

int sdsp_hip_fir_process(sdsp_hip_fir_plan *p, void *data, uint64_t channels, uint64_t samples, uint64_t stride,
                         void *state, void *stream)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    if (channels == 0 || samples == 0)
        return SDSP_HIP_OK;
    if (!data)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data is null");
    if (stride < samples && channels > 1)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "stride must be >= samples");
    if (int rc = use_device(p->device))
        return rc;
    fir_args a{};
    a.data = data;
    a.state = p->taps > 1 ? state : nullptr;
    a.h = p->h_dev;
    a.channels = channels;
    a.samples = samples;
    a.stride = stride;
    a.taps = p->taps;
    return launch_fir(p->precision, a, p->variant, stream);
}

int sdsp_hip_fir_process_host(sdsp_hip_fir_plan *p, void *host_data, uint64_t channels, uint64_t samples,
                              uint64_t stride, void *host_state)
{
    if (!p)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "plan is null");
    if (channels == 0 || samples == 0)
        return SDSP_HIP_OK;
    if (!host_data)
        return fail(SDSP_HIP_ERR_INVALID_ARG, "data is null");
    if (int rc = use_device(p->device))
        return rc;
    const size_t rs = p->precision == SDSP_HIP_F64 ? 8 : 4;
    const size_t data_bytes = ((channels - 1) * stride + samples) * rs;
    uint64_t state_bytes = 0;
    sdsp_hip_fir_state_bytes(p, channels, &state_bytes);
    const bool with_state = host_state && state_bytes;
    void *d = nullptr, *s = nullptr;
    HIP_TRY(hipMalloc(&d, data_bytes));
    int rc = SDSP_HIP_OK;
    hipError_t e = hipMemcpy(d, host_data, data_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && with_state) {
        e = hipMalloc(&s, state_bytes);
        if (e == hipSuccess)
            e = hipMemcpy(s, host_state, state_bytes, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess)
        rc = hip_fail(e, "fir host staging");
    if (!rc)
        rc = sdsp_hip_fir_process(p, d, channels, samples, stride, s, nullptr);
    if (!rc) {
        e = hipMemcpy(host_data, d, data_bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess && with_state)
            e = hipMemcpy(host_state, s, state_bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            rc = hip_fail(e, "fir host read-back");
    }
    (void)hipFree(d);
    (void)hipFree(s);
    return rc;
}
}

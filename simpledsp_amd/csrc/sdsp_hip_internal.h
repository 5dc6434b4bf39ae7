// sdsp_hip_internal.h -- shared declarations of the libsdsp_hip implementation (not installed).
#pragma once

#include "sdsp_hip.h"

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

namespace sdsp_hip
{
extern thread_local std::string g_last_error;
int fail(int code, const std::string &msg);

// Raise a kernel's dynamic-LDS limit (needed above 64 KiB).  The attribute belongs to the function object of the
// CURRENT device, so it is set once per (kernel, device): `done` is the caller's per-kernel mask, bit = device index.
// Safe from concurrent host threads (sharded entry points run one thread per device).  Defined in capi.hip.
int ensure_dynamic_lds(const void *kernel, size_t bytes, std::atomic<uint64_t> &done);

// host_math.cpp
void make_twiddles(uint32_t n, int direction, std::vector<double> &out);
int design_lp(uint32_t m, double f0, double fs, double gain_in, double *a, double *b, double *gain);
int design_hp(uint32_t m, double f0, double fs, double gain_in, double *a, double *b, double *gain);
int design_bp(uint32_t m, double f0, double fs, double q, double gain_in, double *a, double *b, double *gain);
int design_bs(uint32_t m, double f0, double fs, double q, double gain_in, double *a, double *b, double *gain);
int design_fir(uint32_t taps, int filter_type, double f0, double fs, double q, double gain_in, double *h);
int preload(uint32_t m, int filter_type, const double *a, const double *b, double gain, double value,
            double *mem);

// ------------------------------------------------------------------------------------------
// FFT: one "tile" launch = every workgroup transforms `cols` independent length-n sequences held
// in LDS.  The same kernel serves contiguous batches (cols transforms per workgroup) and the two
// HBM passes of the four-step decomposition for transforms larger than LDS.
struct fft_tile_args {
    const void *in;
    void *out;
    const void *tw;     // W_n^j, j in [0,n): direction already folded in (conjugated for reverse)
    const void *tw_big; // four-step pass 1: W_N^j, j in [0,N) of the whole transform; else null
    uint32_t n;         // sub-transform length held in LDS
    uint32_t log2n;
    uint32_t cols;      // sequences per workgroup
    uint32_t pitch;     // LDS row pitch in complex elements (cols + padding)
    uint64_t total_cols;       // sequences in the whole launch (ragged last tile is masked)
    uint32_t tiles_per_group;  // tiles that make up one big transform (1 for contiguous batches)
    uint64_t group_stride;     // elements between big transforms
    uint64_t in_tile_step, out_tile_step;   // element offset between consecutive tiles of a group
    uint64_t in_si, in_sc;     // input strides: sequence index i, column c
    uint64_t out_sk, out_sc;   // output strides: frequency index k, column c
    uint32_t in_c_fast, out_c_fast; // which index runs fastest over the lanes (coalescing)
    uint32_t reverse;          // direction of the +-i rotation (twiddles are pre-conjugated)
    uint32_t apply_scale;      // multiply by `scale` on store (reverse_fft::ScaleValues)
    float scale;               // 1/N
    double scale_d;
};

int launch_fft_tile(int precision, int radix, const fft_tile_args &a, uint64_t n_tiles, void *stream);
size_t fft_tile_lds_bytes(int precision, uint32_t n, uint32_t pitch);
size_t fft_tile_max_lds_bytes();

// fast path: batched n = 4096, radix 4, f32 (BASELINE config 2 / 5)
struct fft4096_args {
    void *data;
    const void *tw; // the plan's thread-twiddle table (make_thread_twiddles_4096), f32 complex
    uint64_t batch;
    float scale;
    int reverse;
};
int launch_fft4096_r4_f32(const fft4096_args &a, int variant, void *stream);
int fft4096_num_variants();
int launch_fft4096_r2_f32(const fft4096_args &a, void *stream); // tuned radix-2 sibling
// fused y = IFFT(FFT(x) .* h), n = 4096, f32; tw = FORWARD thread-twiddle table
int launch_fft4096_conv_f32(void *data, const void *tw, const void *h, uint64_t batch, void *stream);
// data[b][i] *= h[i] (generic three-launch convolution path)
int launch_pointwise_mul(int precision, void *data, const void *h, uint32_t n, uint64_t batch, void *stream);

// fast path for every other batched f32 size 16 .. 4096, radix 2 or 4 (register-pass family)
struct fft_reg_args {
    void *data;
    const void *tw; // fft_reg.hip (f32): the plan's thread-twiddle table; fft_reg64 / fft_big: the row W_n^j
    uint32_t n;
    int radix;
    uint64_t batch;
    float scale;
    double scale_d = 1.0;      // f64 kernels
    int reverse;
    int nontemporal;
    int real_mode = 0;         // 0 complex; 1 real forward (split); 2 real inverse (merge): SURVEY 8(f)-3
    const void *tw2 = nullptr; // real modes: W_{2n}^k
};
bool fft_reg_supports(uint32_t n, int radix);
int launch_fft_reg_f32(const fft_reg_args &a, void *stream);
bool fft_reg64_supports(uint32_t n, int radix); // f64 family: 16 .. 8192
int launch_fft_reg_f64(const fft_reg_args &a, void *stream);
// N = 8192 / 16384 / 32768, radix 2, f32: transform held in registers, LDS only for the exchanges (fft_big.hip)
// N = 1024 f32, one transform per wave (fft_wave.hip); a.tw = the register-pass thread-twiddle table
bool fft_wave_supports(uint32_t n, int radix);
int launch_fft_wave_f32(const fft_reg_args &a, void *stream);
int launch_fft_wave_f64(const fft_reg_args &a, void *stream); // scale_d
// N = 256 / 512 / 2048 f32, radix-2 stages: 1024 points (or one transform of 2048) per wave; a.tw = twt_wave
bool fft_wave2_supports(uint32_t n, int radix);
int launch_fft_wave2_f32(const fft_reg_args &a, void *stream);
bool fft_big_supports(uint32_t n, int radix);
bool fft_big_real_supports(uint32_t n, int radix); // real-input plans, n = n_real / 2
bool fft_big_conv_supports(uint32_t n, int radix); // fused convolution
int launch_fft_big_f32(const fft_reg_args &a, void *stream);
// the same design in double: N = 4096 / 8192 / 16384, radix-2 stages (fft_big64.hip); a.tw = the [pass][stage][thread] table in double
bool fft_big64_supports(uint32_t n, int radix);
bool fft_big64_real_supports(uint32_t n, int radix); // the real-input form (real_mode = 1 / 2, W_2N in tw2)
bool fft_big64_conv_supports(uint32_t n, int radix); // the fused convolution form (launch_fft_big_f64 with real_mode = 3, h in tw2)
int launch_fft_big_f64(const fft_reg_args &a, void *stream);

// N = 8192 / 16384, f32: one leading radix-2 / radix-4 stage + the tuned N = 4096 radix-4 machinery (fft_mix.hip)
struct fft_mix_args {
    void *data;
    const void *tw;      // sub-transform thread-twiddle table (layout of the N = 4096 radix-4 kernel's)
    const void *tw_lead; // [q - 1][t] = W_N^(q t), q < R, t < 256
    uint32_t n;
    uint64_t batch;
    float scale;
    int reverse;
};
bool fft_mix_supports(uint32_t n);
int launch_fft_mix_f32(const fft_mix_args &a, void *stream);

// N = 2^16 .. 2^19, f32: two passes over HBM, N = N1 x N2 with N1, N2 in {256, 512, 1024} (fft_2pass.hip)
struct fft_2pass_args {
    void *data;          // count x n complex, in place
    void *workspace;     // count x n complex
    const void *tw_1024; // W_1024^j, direction-folded
    uint32_t n;
    uint64_t count;
    float scale;
    int reverse;
    double scale_d = 1.0; // f64
    const void *hmul = nullptr; // forward only: every output X[k] leaves multiplied by hmul[k] (the fused convolution's forward half)
};
bool fft_2pass_supports(uint32_t n, int precision);
int launch_fft_2pass(int precision, const fft_2pass_args &a, void *stream);

// N = 2^16 .. 2^19, f32 (variant 1) and larger / f64: the two streaming passes around 16 x batch row transforms (fft_mid.hip)
int launch_fft_mid_cols(int precision, const void *in, void *out, const void *tw, uint32_t n2, uint64_t batch, int reverse,
                        void *stream);
int launch_fft_mid_untwist(int precision, const void *in, void *out, uint32_t n2, uint64_t batch, void *stream);

// fast path: batched n = 2^20, radix 2, f32 (BASELINE config 3), one chunk of transforms
struct fft1m_args {
    void *data;          // count x 2^20 complex, in place
    void *workspace;     // count x 2^20 complex
    const void *tw_n;    // W_N^j, j < 1024 used
    const void *tw_1024; // W_1024^j
    uint64_t count;
    float scale;
    int reverse;
};
int launch_fft1m_pass(const fft1m_args &a, int which, void *stream); // which = 1 columns pass, 2 rows pass (variant 1)
// the default schedule: ONE persistent launch over `count` transforms (fft1m_kernels.h)
struct fft1m_fused_args {
    void *data;          // count x 2^20 complex, in place
    void *workspace;     // ring x 2^20 complex
    const void *tw_1024; // W_1024^j
    void *sync;          // fft1m_sync_bytes(count) bytes of device memory (zeroed by the launcher)
    void *sticky = nullptr;  // one word outside that block, or null: set to 1 by a launch that gave up, never cleared by the launcher
    uint64_t spin_limit = 200000000ull; // wall_clock64 ticks (100 MHz) a hand-off poll may take: 2 s
    uint64_t count;
    uint32_t ring, lag;  // intermediate ring slots per queue; steps pass 2 trails pass 1 (lag < ring)
    uint32_t queues;     // independent ticket queues (workspace holds queues x ring transforms)
    float scale;
    int reverse;
};
size_t fft1m_sync_bytes(uint64_t count, uint32_t queues);
int launch_fft1m_fused(const fft1m_fused_args &a, void *stream);

// the two-pass sizes of fft_2pass.hip in ONE persistent, ticketed launch (the schedule of launch_fft1m_fused; handoff.h)
struct fft_2pass_fused_args {
    void *data;          // count x n complex, in place
    void *workspace;     // queues x ring x unit transforms
    const void *tw_1024; // W_1024^j
    void *sync;          // fft_2pass_sync_bytes(units, queues) bytes of device memory (zeroed by the launcher)
    void *sticky = nullptr;  // one word outside that block, or null: set to 1 by a launch that gave up
    uint64_t spin_limit = 200000000ull; // wall_clock64 ticks (100 MHz) a hand-off poll may take: 2 s; 0 = fault injection
    uint64_t count;
    uint32_t n;
    uint32_t unit;       // transforms per ticket step
    uint32_t ring, lag;  // ring slots (units) per queue; steps pass 2 trails pass 1 (lag < ring)
    uint32_t queues;
    float scale;
    double scale_d;
    int reverse;
    const void *hmul = nullptr; // forward only: every output X[k] leaves multiplied by hmul[k]
};
size_t fft_2pass_sync_bytes(uint64_t units, uint32_t queues);
void fft_2pass_fused_shape(uint32_t n, int precision, uint32_t *unit, uint32_t *queues, uint32_t *ring, uint32_t *lag);
int launch_fft_2pass_fused(int precision, const fft_2pass_fused_args &a, void *stream);

// ------------------------------------------------------------------------------------------
// IIR bank
struct iir_args {
    void *data;
    void *state; // nullable
    uint64_t channels, samples, stride;
    uint32_t sections;
    int kind;
    // coefficients in double; kernels round to their precision
    double gain;
    double a1[SDSP_HIP_MAX_SECTIONS], a2[SDSP_HIP_MAX_SECTIONS];
    double b1[SDSP_HIP_MAX_SECTIONS], b2[SDSP_HIP_MAX_SECTIONS];
};
int launch_iir(int precision, const iir_args &a, int variant, void *stream);
// the kernel launch_iir would run for this shape and variant (iir.hip: iir_select -- the same function the launcher uses)
const char *iir_kernel_for(int precision, const iir_args &a, int variant);
int launch_iir_interleaved(int precision, const iir_args &a, int variant, void *stream);

// FIR bank
struct fir_args {
    void *data;
    void *state;   // nullable; channels x (taps-1), newest first
    const void *h; // device, plan precision, `taps` values
    uint64_t channels, samples, stride;
    uint32_t taps;
};
int launch_fir(int precision, const fir_args &a, int variant, void *stream);
} // namespace sdsp_hip

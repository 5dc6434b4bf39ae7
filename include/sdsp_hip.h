/*
 * sdsp_hip.h -- C ABI of the MI355X (gfx950) batched-FFT + cascaded-biquad engine.
 *
 * This is the drop-in boundary for simpledsp's FFT/IIR hot path.  The reference has no FFI
 * layer of its own: its boundary is the header-only C++ surface (include/sdsp/fft.h,
 * include/sdsp/casc_2o_iir.h).  The sdsp:: headers shipped in include/sdsp/ keep that surface
 * and call the entry points below; each entry point cites the reference interface it replaces
 * (paths relative to the reference checkout).  Plain pointers and sizes only -- no torch, no
 * C++ types.  `stream` arguments are hipStream_t passed as void* (NULL = the default stream).
 *
 * Conventions
 *   - every function returns an sdsp_hip_status (0 = ok, negative = error); a human-readable
 *     description of the calling thread's last error: sdsp_hip_last_error_string().
 *   - the caller owns all data buffers; transforms and filters run IN PLACE (fft.h:259,302;
 *     casc_2o_iir.h:37,71).  Plans own twiddles/coefficients/workspaces on their device.
 *   - complex data is interleaved (re,im), the layout of std::complex (fft.h:51-52).
 *   - there is no CPU fallback: without a usable HIP device the compute calls fail with
 *     SDSP_HIP_ERR_NO_DEVICE / SDSP_HIP_ERR_HIP.
 *   - distinct plans may be used from distinct host threads concurrently.
 */
#ifndef SDSP_HIP_H
#define SDSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SDSP_HIP_OK = 0,
    SDSP_HIP_ERR_INVALID_SIZE = -1, /* replaces static_assert fft.h:261,304 / casc_2o_iir.h:25 */
    SDSP_HIP_ERR_UNSUPPORTED = -2,
    SDSP_HIP_ERR_HIP = -3,
    SDSP_HIP_ERR_NO_DEVICE = -4,
    SDSP_HIP_ERR_INVALID_ARG = -5,
    SDSP_HIP_ERR_NOMEM = -6
} sdsp_hip_status;

typedef enum {
    SDSP_HIP_F32 = 0,
    SDSP_HIP_F64 = 1,
    /* IIR banks only: samples stored as float (8 bytes of HBM traffic per sample, like F32), per-channel state,
     * coefficients and the recurrence in double -- the reference computes in double (casc_2o_iir.h:11-18), and an f32
     * recurrence loses up to 1e-4 at low normalised cutoffs (f0/fs = 0.005); this mode is within float rounding of the
     * double result everywhere.  State buffers then hold doubles. */
    SDSP_HIP_F32_F64STATE = 2
} sdsp_hip_precision;
/* forward_fft / reverse_fft policy, fft.h:121-146 (reverse: conjugate twiddles and 1/N scale) */
typedef enum { SDSP_HIP_FORWARD = 1, SDSP_HIP_REVERSE = -1 } sdsp_hip_direction;
/* filter_type.h:6 -- the same integer values */
typedef enum {
    SDSP_HIP_FILTER_NONE = 0,
    SDSP_HIP_FILTER_LOW_PASS = 1,
    SDSP_HIP_FILTER_HIGH_PASS = 2,
    SDSP_HIP_FILTER_BAND_PASS = 3,
    SDSP_HIP_FILTER_BAND_STOP = 4 /* not in the reference (README.md:15 TODO); SURVEY 8(f)-4 */
} sdsp_hip_filter_type;
/* which process() body runs: casc_2o_iir::process (casc_2o_iir.h:36-80) or the
 * numerator-folded casc_2o_iir_{lp,hp,bp}::process_spec (:286-295, :344-353, :402-411) */
typedef enum {
    SDSP_HIP_IIR_GENERIC = 0,
    SDSP_HIP_IIR_LP = 1,
    SDSP_HIP_IIR_HP = 2,
    SDSP_HIP_IIR_BP = 3
} sdsp_hip_iir_kind;

#define SDSP_HIP_MAX_SECTIONS 16
#define SDSP_HIP_RADIX_AUTO 0
#define SDSP_HIP_STAGES_2_THEN_4 24 /* plan info: radix-2 stage(s) in front of radix-4 stages (mixed radix) */
#define SDSP_HIP_FIR_MAX_TAPS 4096

typedef struct sdsp_hip_fft_plan sdsp_hip_fft_plan;
typedef struct sdsp_hip_iir_plan sdsp_hip_iir_plan;

/* ------------------------------------------------------------------ runtime */

const char *sdsp_hip_last_error_string(void);
const char *sdsp_hip_version(void);
int sdsp_hip_device_count(int *count);
/* device memory helpers so that a C/C++ host can stay free of <hip/hip_runtime.h> */
int sdsp_hip_malloc(void **dev_ptr, size_t bytes, int device);
int sdsp_hip_free(void *dev_ptr, int device);
int sdsp_hip_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes, int device);
int sdsp_hip_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes, int device);
int sdsp_hip_device_synchronize(int device);

/* ------------------------------------------------------------------ size helpers, fft.h:12-43 */

unsigned sdsp_hip_log2(unsigned num);
unsigned sdsp_hip_log4(unsigned num);
int sdsp_hip_is_power_of_2(unsigned num);
int sdsp_hip_is_power_of_4(unsigned num);
/* digit_reverse<N,base>, fft.h:217-236 (the GPU folds this into load/store addressing) */
unsigned sdsp_hip_digit_reverse(unsigned n, unsigned base, unsigned x);
/* one row of the run-time twiddle precompute that replaces the compile-time calc_wCoeffs
 * (fft.h:197-214): out[j] = exp(-/+ 2*pi*i*j/n), j in [0,n), n interleaved complex doubles,
 * first quadrant from libm, the rest by exact mirror symmetry (fft.h:148-194). */
int sdsp_hip_calc_twiddles(unsigned n, int direction, double *out);

/* ------------------------------------------------------------------ FFT */

/*
 * Replaces sdsp::fft_radix2<T,N> (fft.h:258-299, radix = 2, n a power of 2) and
 * sdsp::fft_radix4<T,N> (fft.h:301-360, radix = 4, n a power of 4) for a BATCH of transforms.
 * radix: 2 or 4 checks the size the way the reference's function does (power of 2 / power of 4) and selects the
 * butterflies where a kernel of that stage type exists -- every n <= 16384, both precisions: a radix-4 plan runs radix-4
 * butterflies, a radix-2 plan radix-2 butterflies.  Above that the DEFAULT kernels of both radices are the multi-pass
 * kernels, whose register passes are radix-2 butterflies (the same DFT, held to the same tolerance against the radix-4
 * oracle); sdsp_hip_fft_plan_get_info reports the butterflies that actually run in `stage_radix`, and a variant >= 8 of a
 * radix-4 plan runs genuine radix-4 stages at any size (coverage kernel, fft_tile.hip).
 * radix = SDSP_HIP_RADIX_AUTO (0) asks for the fastest kernel of the size: any power of two through one entry (radix-4
 * stages where n is a power of 4 -- except n = 16384 --, radix-2 stages otherwise; `radix` in the plan info says which).
 * The mixed-radix case of SURVEY 8(f)-4, n = 8192 = 2 * 4^6 through the radix-4 machinery behind ONE radix-2 stage, is
 * variant 1 of AUTO plans of that size (kernel "sdsp_fft_mix_f32", stage_radix SDSP_HIP_STAGES_2_THEN_4); their default is
 * the registers-resident radix-2 kernel "sdsp_fft_big_kernel", which measured 1-2 points faster.
 * n must satisfy the radix (else SDSP_HIP_ERR_INVALID_SIZE -- the run-time form of the
 * reference's static_asserts).  `max_batch` sizes the plan-owned workspace that transforms too
 * large for on-chip memory need (n > 32768 in f32, n > 16384 in f64): allocated here when the plan's default kernel is
 * multi-pass (so sdsp_hip_fft_exec never allocates and can be stream-captured), on first use by an alternate variant
 * otherwise; larger batches are processed in slices of max_batch (the two-pass sizes never hold more than 256 MiB of intermediate:
 * their workspace stops growing there).  Twiddles are precomputed in double, rounded once to
 * the plan precision and kept resident in HBM.
 * One exec per plan in flight: the multi-pass kernels share the plan's workspace (and the persistent kernels their ticket
 * counters), so two sdsp_hip_fft_exec calls on the SAME plan must not overlap (different streams / host threads): use one
 * plan per stream.  Distinct plans are independent.
 */
int sdsp_hip_fft_plan_create(sdsp_hip_fft_plan **plan, uint32_t n, int radix, int direction,
                             int precision, uint64_t max_batch, int device);
int sdsp_hip_fft_plan_destroy(sdsp_hip_fft_plan *plan);

/* data: DEVICE pointer, batch x n interleaved complex of the plan precision, transformed in
 * place.  Asynchronous on `stream`. */
int sdsp_hip_fft_exec(sdsp_hip_fft_plan *plan, void *data, uint64_t batch, void *stream);
/* same with a HOST pointer: H2D, transform, D2H, synchronous (the single-call drop-in path) */
int sdsp_hip_fft_exec_host(sdsp_hip_fft_plan *plan, void *host_data, uint64_t batch);
/* contiguous batch split over `n_plans` devices (one plan per device, same n/radix/direction/
 * precision), one host thread + stream per device, no collective: SURVEY 8(e) */
int sdsp_hip_fft_exec_sharded(sdsp_hip_fft_plan *const *plans, int n_plans, void *host_data,
                              uint64_t batch);

/*
 * Fast convolution, SURVEY 8(f)-1: per transform, in place, data <- IFFT( FFT(data) .* h ), i.e. what a
 * reference user writes as fft_radix4(x); x[k] *= H[k]; fft_radix4<reverse_fft>(x); (the reverse_fft
 * policy with its 1/N scale exists for exactly this, fft.h:121-133).  `plan` must be a FORWARD plan;
 * h: DEVICE pointer to n complex values of the plan precision (frequency response, natural order).
 * f32 with n = 16 .. 16384 (radix-2 plans: .. 32768) and f64 with n = 16 .. 8192 (radix-2 plans: .. 16384) run as ONE kernel (one HBM read + one
 * write per element instead of three of each); larger n runs forward and reverse as two transforms with the multiply riding on the
 * forward transform's last pass (two-pass sizes) or as a third launch.
 */
int sdsp_hip_fft_convolve(sdsp_hip_fft_plan *plan, void *data, const void *h, uint64_t batch,
                          void *stream);

/*
 * Real-input packing, SURVEY 8(f)-3.  Every reference test feeds REAL signals through the complex
 * transform (testFFT.cpp:23-25,84-90); a plan made here moves half the bytes: n_real real samples
 * are transformed as n_real/2 complex points and split / merged on chip.
 *   direction FORWARD: data = batch x n_real floats in, batch x n_real/2 complex out, in place:
 *     out[k] = X[k] for 0 < k < n_real/2 (the spectrum of the real signal, same values the complex
 *     transform would give), out[0] = (X[0], X[n_real/2]) -- both are real; the upper half of the
 *     spectrum is the conjugate mirror.
 *   direction REVERSE: the inverse of that (packed half spectrum in, real samples out, 1/N scaled).
 * f32; radix 2: n_real = 32 .. 65536 a power of 2; radix 4: n_real/2 a power of 4, n_real <= 32768.  Use the plan
 * with sdsp_hip_fft_exec / _exec_host (batch counts transforms).
 */
int sdsp_hip_rfft_plan_create(sdsp_hip_fft_plan **plan, uint32_t n_real, int radix, int direction,
                              uint64_t max_batch, int device);
/* the same with a precision: SDSP_HIP_F64 packs n_real doubles <-> n_real/2 complex doubles (n_real = 32 .. 16384; radix 2: .. 32768); the
 * reference computes in double (fft.h:51-52). */
int sdsp_hip_rfft_plan_create_p(sdsp_hip_fft_plan **plan, uint32_t n_real, int radix, int direction, int precision,
                                uint64_t max_batch, int device);

/* Synchronises the plan's device and reports the health of its last sdsp_hip_fft_exec call.  The persistent kernels (n = 2^20
 * f32: "sdsp_fft1m_fused"; the other two-pass sizes where the plan info names "sdsp_fft2p_fused")
 * hand an intermediate from one workgroup to another inside a launch; every wait of that hand-off is bounded (2 s), and a
 * wait that gives up marks the call (a sticky word that every launch of the call can set and only the next call clears)
 * instead of hanging the GPU: this returns SDSP_HIP_ERR_HIP then, SDSP_HIP_OK otherwise (always OK for plans whose kernels
 * have no in-kernel hand-off).  The synchronous sdsp_hip_fft_exec_host checks the same word itself and returns the error;
 * ASYNCHRONOUS callers of such plans (sdsp_hip_fft_exec, sdsp_hip_fft_convolve -- whose reverse half is covered too) must call this
 * before trusting the output.  Has no reference
 * counterpart. */
int sdsp_hip_fft_plan_status(sdsp_hip_fft_plan *plan);
/* Testing hook for the error path above: the bound of the hand-off waits in 100 MHz ticks (default 200 000 000 = 2 s);
 * 0 = fault injection: every hand-off wait of the next launches gives up at once (their output is invalid by construction). */
int sdsp_hip_fft_plan_set_wait_limit(sdsp_hip_fft_plan *plan, uint64_t ticks);
/* Kernel launches that one sdsp_hip_fft_exec(plan, data, batch) issues with the plan's current variant (launch pieces and
 * workspace slices included; memsets not counted).  For profilers and bench.py: per-launch bytes = batch x
 * algorithmic_bytes / launches when hbm_passes == 1. */
int sdsp_hip_fft_plan_launches(const sdsp_hip_fft_plan *plan, uint64_t batch, uint64_t *launches);

/* Launch granularity (process-wide; has no reference counterpart).  A batch of transforms whose buffer is larger than
 * 1.5 x `bytes` is issued as consecutive launches over pieces of at most `bytes` of the buffer, in stream order (same bits,
 * same single call).  Why: workgroups are dealt to the eight XCDs round-robin and the XCDs drift apart over a long launch, so
 * the window of DRAM pages the chip works on widens.  Measured on the N = 4096 kernel: one launch over 8 GiB 72.1 % of HBM
 * peak, the same buffer in 2 GiB pieces 75.4 %, in 1 GiB pieces 76.3 %, 512 MiB 75.8 %, 256 MiB 74.4 % (DESIGN.md section
 * 5.1c).  Applies to the FFT kernels that cover a batch with one launch, N <= 8192 (many short workgroups); not to
 * N = 16384 / 32768 (one or two transforms fill a CU: pieces cost 1.6 points there), not to the multi-pass sizes (they chunk
 * by their workspace) and not to the IIR / FIR kernels (a workgroup there walks whole rows for milliseconds: pieces only add launch
 * tails -- 70.0 % in one launch, 68.7 / 66.4 / 45.2 % in 2 GiB / 1 GiB / 512 MiB pieces).  bytes = 0: never split. */
#define SDSP_HIP_DEFAULT_PIECE_BYTES (1ull << 30)
int sdsp_hip_set_launch_piece_bytes(uint64_t bytes);
int sdsp_hip_get_launch_piece_bytes(uint64_t *bytes);

typedef struct {
    uint32_t n;
    int radix;
    int direction;
    int precision;
    int device;
    int hbm_passes;              /* passes over HBM of the kernel(s) that run: 1 = one read + one write per element */
    uint64_t algorithmic_bytes;  /* per transform: n * sizeof(complex) * 2 (read + write once) */
    uint64_t workspace_bytes;
    uint64_t twiddle_bytes;
    char kernel[64];             /* name of the dominant kernel (for rocprofv3 matching) */
    int stage_radix;             /* the butterflies that kernel executes: 2, 4, or SDSP_HIP_STAGES_2_THEN_4 */
} sdsp_hip_fft_plan_info;
int sdsp_hip_fft_plan_get_info(const sdsp_hip_fft_plan *plan, sdsp_hip_fft_plan_info *info);
/* copy the plan's resident twiddle row W_n^j (plan precision, n complex) back to the host */
int sdsp_hip_fft_plan_get_twiddles(const sdsp_hip_fft_plan *plan, void *host_out);
/* choose among kernel variants of a plan (tuning/testing).  Variant 0 is the default; a plan has at most two documented
 * alternates (same transform, same tolerance; DESIGN.md section 5 lists them per size: e.g. n = 4096 radix 4: 1, 2 = other
 * store / barrier schedules of the same kernel; n = 8192 AUTO plans and n = 16384 radix-4 plans: 1 = the fft_mix.hip kernel
 * (mixed radix / leading radix-4 stage); the two-pass sizes (f32 n = 2^16 .. 2^19, 2^21, 2^22; f64 n = 2^15 .. 2^20): 1 = three
 * streaming passes, 3 = the other SCHEDULE of the same two passes -- one persistent, ticketed launch ("sdsp_fft2p_fused") against two
 * launches per chunk ("sdsp_fft2p_cols+sdsp_fft2p_rows"), bit-identical results; the default is the persistent launch (level or faster at
 * every size, 1 - 5 points of HBM peak), and plans whose workspace (max_batch) is smaller than the
 * persistent launch's 256 MiB ring of intermediates run the two launches under either number; n = 2^20 f32: 1 = two launches per chunk, 2 = the persistent launch through
 * fft_2pass.hip's generic kernel instead of the dedicated one (42.3 against 42.7 % of HBM peak); n = 16 .. 2048 f32 (the register-pass family, fft_reg.hip): 1 = the
 * same kernel with the default cache policy, 2 = the one-wave kernel (fft_wave.hip) at n = 256 / 1024 / 2048; real-input plans of
 * n_real = 1024, whose default IS the one-wave kernel: 1, 2 = the register-pass family with the default / streaming cache
 * policy (n_real = 512 / 2048: as the complex plans); the same as variant 2 of sdsp_hip_fft_convolve for the fused convolution of n = 256 .. 2048 (default: the one-wave kernel); f64 n = 1024: 1 = the one-wave kernel); any larger number selects the untuned coverage kernel
 * (fft_tile.hip), which the tests use as an independent implementation. */
int sdsp_hip_fft_plan_set_variant(sdsp_hip_fft_plan *plan, int variant);

/* ------------------------------------------------------------------ cascaded biquads */

/*
 * Coefficient design (host, double) -- set_lp_coeff / set_hp_coeff / set_bp_coeff,
 * casc_2o_iir.h:168-194, :140-166, :82-138 (identical in the specialised classes :297-467).
 * sections = m_t (even).  Outputs: a[sections*3], b[sections*3], *gain (= m_gain).
 */
int sdsp_hip_iir_design_lp(uint32_t sections, double f0, double fs, double gain_in, double *a,
                           double *b, double *gain);
int sdsp_hip_iir_design_hp(uint32_t sections, double f0, double fs, double gain_in, double *a,
                           double *b, double *gain);
int sdsp_hip_iir_design_bp(uint32_t sections, double f0, double fs, double q, double gain_in,
                           double *a, double *b, double *gain);
/*
 * Band-stop design -- the reference's README.md:15 TODO, SURVEY 8(f)-4 (no reference code: parity is
 * pinned to scipy.signal.butter(btype='bandstop') instead).  Same parameters as design_bp: centre f0,
 * -3 dB width f0/q, Butterworth prototype of order `sections`.  Every section's numerator is
 * [1, -2cos(2 pi f0/fs), 1]; run it on a GENERIC plan.
 */
int sdsp_hip_iir_design_bs(uint32_t sections, double f0, double fs, double q, double gain_in,
                           double *a, double *b, double *gain);
/* preload_filter, casc_2o_iir.h:197-214: mem[(sections+1)*3] for a steady input `value` */
int sdsp_hip_iir_preload(uint32_t sections, int filter_type, const double *a, const double *b,
                         double gain, double value, double *mem);

/*
 * A bank of identical cascades (shared coefficients, per-channel state) -- the batched form of
 * sdsp::casc_2o_iir<m_t> (kind GENERIC) and sdsp::casc_2o_iir_{lp,hp,bp}<m_t>.
 * a: sections*3, b: sections*3 (may be NULL for the specialised kinds), gain: m_gain.
 */
int sdsp_hip_iir_plan_create(sdsp_hip_iir_plan **plan, uint32_t sections, int kind,
                             const double *a, const double *b, double gain, int precision,
                             int device);
int sdsp_hip_iir_plan_destroy(sdsp_hip_iir_plan *plan);

/*
 * process(): casc_2o_iir.h:36-80 / :228-263 for `channels` independent streams.
 * data: DEVICE pointer; channel c's samples are data[c*stride + 0 .. samples) (channel-major,
 * each channel is what one reference process() call sees), filtered in place.
 * state: DEVICE pointer or NULL.  NULL = every channel starts from a zero-initialised filter
 *   and the final state is dropped.  Otherwise state[(3*(sections+1)) * channels] of the plan
 *   precision, laid out state[(3*j + age) * channels + c] = y_j[n-1-age] of channel c
 *   (j = 0 is the gain-scaled input history, j = sections the output history, age 0..2):
 *   the reference's m_mem ring (casc_2o_iir.h:15) rotated so that m_pos is implicit.  Read at
 *   entry, written at exit, so consecutive calls continue the stream bit-identically
 *   (testIIR.cpp:61-75).
 */
int sdsp_hip_iir_process(sdsp_hip_iir_plan *plan, void *data, uint64_t channels,
                         uint64_t samples, uint64_t stride, void *state, void *stream);
/*
 * The same filter bank on the interleaved ("wire") layout: sample s of channel c is
 * data[s*stride + c] (stride >= channels elements between consecutive sample rows), i.e. what an
 * ADC / network frame delivers.  No transpose is needed, so this is the fastest entry; results are
 * bit-identical to sdsp_hip_iir_process on the transposed data.  Fast path: rows 8-byte aligned and
 * an even channel count for f32 (any 4-byte aligned f32 shape still works).  SURVEY 8(f)-2.
 */
int sdsp_hip_iir_process_interleaved(sdsp_hip_iir_plan *plan, void *data, uint64_t channels,
                                     uint64_t samples, uint64_t stride, void *state, void *stream);
/* same with HOST pointers (synchronous) */
int sdsp_hip_iir_process_host(sdsp_hip_iir_plan *plan, void *host_data, uint64_t channels,
                              uint64_t samples, uint64_t stride, void *host_state);
/* contiguous channel range split over devices, no collective */
int sdsp_hip_iir_process_sharded(sdsp_hip_iir_plan *const *plans, int n_plans, void *host_data,
                                 uint64_t channels, uint64_t samples);
/* bytes of a state buffer for `channels` channels */
int sdsp_hip_iir_state_bytes(const sdsp_hip_iir_plan *plan, uint64_t channels, uint64_t *bytes);
/* kernel variants of sdsp_hip_iir_process (tuning / testing; identical arithmetic, bit-identical results): 0 = default -- the
 * landing-slot kernel (LDS-DMA fill one tile ahead of the recurrence) for f32 banks of up to 4 sections on whole [64
 * channels x 512 bytes] tiles, the super-tile kernel for everything else; 1 = wide super-tile; 2 = direct (any alignment);
 * 3 = super-tile.  DESIGN.md section 5.4. */
int sdsp_hip_iir_plan_set_variant(sdsp_hip_iir_plan *plan, int variant);
/* name of the kernel sdsp_hip_iir_process would launch for this buffer shape with the plan's variant (for matching
 * rocprofv3 rows); the same selection function as the launcher's.  Has no reference counterpart. */
int sdsp_hip_iir_plan_kernel(const sdsp_hip_iir_plan *plan, const void *data, uint64_t channels, uint64_t samples,
                             uint64_t stride, char *name, size_t name_bytes);

/* ------------------------------------------------------------------ FIR filter bank */

/*
 * The reference lists "FIR filter" as a TODO (README.md:16) and has no code for it -- SURVEY 8(f)-4.
 * These entries follow the conventions of the IIR bank above (what sdsp::casc_2o_iir does for one
 * stream, casc_2o_iir.h:36-80: in place, stateful across calls); parity is pinned to
 * scipy.signal.firwin / lfilter, not to the reference.
 *
 * Design: Hamming-windowed sinc, `taps` coefficients h[0..taps), unit gain (times gain_in) in the
 * pass band.  filter_type: LOW_PASS / HIGH_PASS (cutoff f0; q ignored), BAND_PASS / BAND_STOP (centre
 * f0, edges f0 -+ f0/(2q)).  Filters that pass fs/2 (HIGH_PASS, BAND_STOP) need an odd tap count.
 */
int sdsp_hip_fir_design(uint32_t taps, int filter_type, double f0, double fs, double q,
                        double gain_in, double *h);

typedef struct sdsp_hip_fir_plan sdsp_hip_fir_plan;
/* h: taps doubles (host), rounded to the plan precision and kept resident on the device */
int sdsp_hip_fir_plan_create(sdsp_hip_fir_plan **plan, uint32_t taps, const double *h,
                             int precision, int device);
int sdsp_hip_fir_plan_destroy(sdsp_hip_fir_plan *plan);
/*
 * y[n] = sum_{k<taps} h[k] x[n-k] for `channels` independent streams, in place.  data: DEVICE pointer,
 * channel-major like sdsp_hip_iir_process (channel c = data[c*stride .. +samples)).
 * state: DEVICE pointer or NULL (zero history, final history dropped); otherwise
 * state[c*(taps-1) + j] = x_c[n-1-j] (j = 0 is the newest input), plan precision, read at entry and
 * written at exit so block-by-block calls equal one long call bit for bit.  Accumulation order:
 * ascending k, one multiply and one add per tap in f64 (bit-identical to the CPU oracle), FMA in f32.
 */
int sdsp_hip_fir_process(sdsp_hip_fir_plan *plan, void *data, uint64_t channels, uint64_t samples,
                         uint64_t stride, void *state, void *stream);
/* same with HOST pointers (synchronous) */
int sdsp_hip_fir_process_host(sdsp_hip_fir_plan *plan, void *host_data, uint64_t channels,
                              uint64_t samples, uint64_t stride, void *host_state);
int sdsp_hip_fir_state_bytes(const sdsp_hip_fir_plan *plan, uint64_t channels, uint64_t *bytes);
int sdsp_hip_fir_plan_set_variant(sdsp_hip_fir_plan *plan, int variant);

#ifdef __cplusplus
}
#endif
#endif /* SDSP_HIP_H */

/*
 * fmrx.h -- C ABI of libfmrx.so: the MI355X (gfx950) implementation of the
 * FM-receiver DSP hot path of mnigm2001/Software-Defined-Radio.
 *
 * This header is the drop-in boundary.  The reference has no FFI layer; its
 * operator API for this path is the set of C++ free functions declared in
 * include/filter.h:18-43 and include/iofunc.h:36 (std::vector<float>&
 * arguments), plus the stdin/stdout process contract of src/project.cpp and
 * src/threadMonoOnly.cpp.  Each entry point below names the reference
 * interface it replaces (file:line under /root/reference).  A header-only C++
 * shim (include/fmrx_filter.hpp) re-exposes the exact filter.h signatures on
 * top of this ABI; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *  - plain pointers + explicit lengths, no C++ or torch types;
 *  - every function returns an fmrx_status (0 = ok); nothing calls exit();
 *    fmrx_last_error() gives the message for the calling thread;
 *  - caller owns every buffer; outputs are caller-allocated to the sizes the
 *    reference's functions resize() to (stated per function);
 *  - "state" buffers are in/out and have exactly the reference's layout;
 *  - the reference's unchecked preconditions (n >= taps-1, n % decim == 0 for
 *    cross-block continuity, taps <= 65535) are validated: FMRX_EINVAL;
 *  - all compute runs on the GPU (HIP kernels).  There is NO CPU fallback:
 *    with no usable device the compute entry points return FMRX_ENODEV.
 *    Only the filter-coefficient design (a3/a4), which is host code in the
 *    reference too and runs once per run, executes on the host;
 *  - functions with a `_dev` suffix take DEVICE pointers and a HIP stream
 *    (passed as void* so this header needs no HIP include) and are
 *    asynchronous on that stream; all others take HOST pointers and are
 *    synchronous (internally H2D -> kernels -> D2H on the current device);
 *  - re-entrant; a pipeline handle is single-owner (not thread-safe), handles
 *    on different devices (= channels) are independent.
 */
#ifndef FMRX_H
#define FMRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define FMRX_API __attribute__((visibility("default")))
#else
#define FMRX_API
#endif

typedef enum fmrx_status {
    FMRX_OK = 0,
    FMRX_EINVAL = 1, /* bad argument / violated precondition */
    FMRX_ENODEV = 2, /* no usable HIP device (never falls back to the CPU) */
    FMRX_EHIP = 3,   /* HIP runtime error */
    FMRX_ENOMEM = 4
} fmrx_status;

/* ------------------------------------------------------------------ */
/* library                                                              */
/* ------------------------------------------------------------------ */
FMRX_API const char *fmrx_version(void);
FMRX_API const char *fmrx_last_error(void);
/* number of HIP devices visible to the process (0 when there is none) */
FMRX_API int fmrx_device_count(void);
/* select the device used by the host-pointer stage functions of this thread */
FMRX_API int fmrx_set_device(int device);

/* Run-time options (tuning and A/B knobs; none is needed in normal use).  The process-wide defaults are
 * the built-in values, overridden ONCE, when the library first needs them, by the environment variable
 * FMRX_<NAME IN CAPITALS>; fmrx_set_option changes the defaults afterwards.  A pipeline handle copies the
 * defaults when it is created and keeps its own set (fmrx_pipeline_set_option); the stage functions use the
 * defaults.  Nothing on a per-block path reads the environment.
 *   "fe_variant"       0 = matrix-core front-end kernels (default; FMRX_FE_VARIANT=mfma), 1 = vector-ALU kernels (=valu)
 *   "fused_min_audio"  audio samples per call from which the fused mono kernel is used (default 65536; 0 = always)
 *   "resample_l2"      1 = the L2-table resampler kernel also for large calls
 *   "resample_exact"   1 = the pipeline's resampler (modes 2/3) keeps the reference's rounding sequence (bit-exact kernels) instead
 *                      of the matrix-core kernel (float32-rounding-equal); the primitive fmrx_resample_fir is always bit-exact
 *   "resample_chains"  workgroups per XCD and tile group of the matrix-core resampler (0 = as many as are resident; A/B)
 *   "overlap_calls"    stereo, modes 0/1: 1 (or 2) = the caller vouches that the INPUT of a fmrx_pipeline_process_dev call is
 *                      complete when the call is made; the stages of consecutive calls then run on internal streams, a call
 *                      apart (1: the next call's front end under this call's PLL and output stage; 2: three lanes).  Outputs
 *                      stay complete in the order of the stream passed to the call; results are bit-identical.  Default 0
 *   "fe_wgs_per_cu"    cap on resident workgroups per CU of the front-end kernels (0 = auto)
 *   "pll_warmup", "pll_segment", "pll_head"   lane shape of the parallel-in-time PLL (-1 = built-in)
 *   "pll_start"        where the parallel PLL's lanes start: 1 (default) = the locked loop solved as a linear system of the
 *                      input's signs + 64 true steps, 0 = the block's initial state plus drift + 512 true steps
 *   "pll_align"        1 = lanes of the parallel PLL (pll_start 0) start on a multiple of the loop's period, 0 (default) = exactly pll_warmup early
 *   "pll_mode"         stereo PLL of the specialised pipeline: 0 = parallel in time, fast math (default),
 *                      1 = serial, fast math, 2 = serial, glibc's functions (cause-by-cause variants)
 *   "demod"            0 (default) = the C++ reference's discriminator fmDemod (src/filter.cpp:248-266); 1 = the Python model's
 *                      arctangent demodulator fmDemodArctan (model/fmSupportLib.py:502-531, float64 atan2 + unwrap): the pipeline
 *                      then runs its unfused kernels (front end -> IF stream -> arctan -> audio / stereo stages)
 *   "bank_streams"     fast stereo banks (fmrx_channels_create_ex, exact = 0): internal streams of a call: 3 (default) = front end |
 *                      band-pass pair + output stage | PLL lanes; 2 = the front end with the other wide kernels; 4 = the output stage apart too
 *   "bank_fused"       fast stereo banks: 1 = front end + band-pass pair in ONE kernel (f32 matrix cores; measured slower: default 0)
 *   "bank_fe_wgs", "bank_fe_wgs_fused"        workgroups per CU of the bank's matrix-core front end (1) / of the fused kernel (2)
 *   "fused_tune", "fe_mfma_tune"              ablation kernels (timing only, WRONG results): FMRX_EINVAL unless the
 *                                            library was built with -DFMRX_TUNING (make TUNING=1; never shipped) */
FMRX_API int fmrx_set_option(const char *name, long value);
FMRX_API int fmrx_get_option(const char *name, long *value);

/* page-locked host buffers for the block-streaming callers (faster, truly
 * asynchronous H2D/D2H); free with fmrx_host_free.  Plain malloc memory works
 * everywhere too. */
FMRX_API int fmrx_host_alloc(void **out, size_t bytes);
FMRX_API int fmrx_host_free(void *p);

/* ------------------------------------------------------------------ */
/* filter-coefficient API (host, float32 bit-compatible)                */
/* ------------------------------------------------------------------ */
/* replaces impulseResponseLPF  include/filter.h:24, src/filter.cpp:103-114.
 * h[num_taps]. */
FMRX_API int fmrx_impulse_response_lpf(float Fs, float Fc, unsigned short num_taps, float *h);
/* replaces bandPass  include/filter.h:20, src/filter.cpp:83-99.  h[num_taps]. */
FMRX_API int fmrx_band_pass(float Fs, float Fb, float Fe, unsigned short num_taps, float *h);

/* ------------------------------------------------------------------ */
/* stage API on host buffers: one call per reference primitive          */
/* ------------------------------------------------------------------ */
/* replaces readStdinBlockData's conversion  include/iofunc.h:36,
 * src/iofunc.cpp:128-135: out[k] = (raw[k]-128)/128.  out[n]. */
FMRX_API int fmrx_u8_to_f32(const uint8_t *raw, size_t n, float *out);
/* replaces the I/Q split  src/project.cpp:98-105.  I[n_pairs], Q[n_pairs]. */
FMRX_API int fmrx_deinterleave(const float *iq, size_t n_pairs, float *I, float *Q);
/* replaces convolveFIR  include/filter.h:26, src/filter.cpp:118-130.
 * y[n + taps - 1]. */
FMRX_API int fmrx_convolve_fir(float *y, const float *x, size_t n, const float *h, size_t taps);
/* replaces convolveBlockFIR  include/filter.h:28-29, src/filter.cpp:133-154.
 * y[n]; state[taps-1] in/out.  Requires n >= taps-1. */
FMRX_API int fmrx_convolve_block_fir(float *y, const float *x, size_t n, const float *h, size_t taps, float *state);
/* replaces convolveBlockFastFIR  include/filter.h:31-32,
 * src/filter.cpp:158-188.  y[n / decim]; state[taps-1] in/out.
 * Requires n >= taps-1, decim >= 1. */
FMRX_API int fmrx_convolve_block_fast_fir(float *y, const float *x, size_t n, const float *h, size_t taps,
                                          float *state, unsigned decim);
/* replaces convolveBlockResampleFIR  include/filter.h:34-35,
 * src/filter.cpp:191-223.  y[(n*upsamp)/decim]; state[taps-1] in/out in the
 * reference's upsampled index space (slots == upsamp-1 mod upsamp are live).
 * Output gain is (1+upsamp), as in the reference (:213).
 * Requires n*upsamp >= taps-1. */
FMRX_API int fmrx_convolve_block_resample_fir(float *y, const float *x, size_t n, const float *h, size_t taps,
                                              float *state, unsigned decim, unsigned upsamp);
/* replaces upsample / downsample  include/filter.h:37-39,
 * src/filter.cpp:227-245.  xu[n*up];  out[ceil(n/ds)] (count via *n_out). */
FMRX_API int fmrx_upsample(const float *x, size_t n, float *xu, int up);
FMRX_API int fmrx_downsample(float *out, size_t *n_out, const float *in, size_t n, unsigned short ds);
/* replaces fmDemod  include/filter.h:41, src/filter.cpp:248-266.  out[n];
 * *prev_i / *prev_q in/out. */
FMRX_API int fmrx_fm_demod(float *out, const float *I, const float *Q, size_t n, float *prev_i, float *prev_q);
/* the arctangent demodulator of the reference's Python MODEL, fmDemodArctan  model/fmSupportLib.py:502-531 (the C++ receiver
 * uses fmDemod above; BASELINE.json names "arctan/PLL demod"): out[k] = the phase step atan2(Q[k], I[k]) - previous phase,
 * unwrapped into (-pi, pi] by np.unwrap's rule; float64 like the model.  *prev_phase in/out: the model's running (unwrapped)
 * phase.  Within 1e-9 of the model (the model's own rounding of its growing phase is 1e-12; tests/golden/arctan.npz).
 * In a pipeline: option "demod" = 1 (FMRX_DEMOD=arctan) replaces the discriminator by this one. */
FMRX_API int fmrx_fm_demod_arctan(double *out, const double *I, const double *Q, size_t n, double *prev_phase);
/* replaces allPass  include/filter.h:18, src/filter.cpp:14-29 (note the
 * reference's argument order: in, state, out).  out[n]; state[nstate] in/out;
 * requires n >= nstate. */
FMRX_API int fmrx_all_pass(const float *in, size_t n, float *state, size_t nstate, float *out);
/* replaces fmPLL  include/filter.h:22, src/filter.cpp:32-80.  nco_out[n+1];
 * state[6] = {integrator, phaseEst, feedbackI, feedbackQ, lastOut, trigOffset}.
 * Serial recurrence with the reference's float32 operations in its order and the
 * sinf / cosf / atan2f of its C library (glibc 2.35, restated in
 * csrc/glibc_libm.hpp and pinned against it): bit-identical to the reference. */
FMRX_API int fmrx_fm_pll(const float *in, size_t n, float *nco_out, float *state, float freq, float Fs,
                         float ncoScale, float phaseAdjust, float normBandwidth);
/* replaces the mixer and L/R combine loops  src/project.cpp:246-248, 277-280 */
FMRX_API int fmrx_stereo_mix(const float *stereo_filt, const float *pll, size_t n, float *mixer);
FMRX_API int fmrx_stereo_combine(const float *stereo_final, const float *mono, size_t n, float *left, float *right);
/* replaces the PCM writer's conversion  src/threadMonoOnly.cpp:185-191:
 * NaN -> 0 else (short)(a*16384).  wrap != 0 reproduces the compiled
 * reference on overflow (int32 truncation, low 16 bits); wrap == 0 saturates. */
FMRX_API int fmrx_pcm16(const float *audio, size_t n, int16_t *out, int wrap);

/* ------------------------------------------------------------------ */
/* diagnostics                                                          */
/* ------------------------------------------------------------------ */
/* out[i] = sinf(a[i]) (fn 0), cosf(a[i]) (fn 1) or atan2f(a[i], b[i]) (fn 2) as the DEVICE evaluates the
 * restatement of glibc 2.35's functions that fmPLL uses (csrc/glibc_libm.hpp): lets a test compare the
 * device build with the C library of the host, bit for bit.  b may be NULL for fn 0, 1.  fn 3, 4, 5: the same three
 * through the branch-free forms the receiver banks' PLL lanes run (general function where those are not defined). */
FMRX_API int fmrx_diag_libm(int fn, const float *a, const float *b, size_t n, float *out);
/* Measurement aid for bench.py: ONE pure streaming read of a device buffer (>= 3 MiB, 16-byte aligned) by
 * one of the access methods the front-end kernels use -- method 0: non-temporal global loads into
 * registers, 1: LDS-DMA ring, one contiguous run per wave (the fused kernel's pattern), m >= 2: LDS-DMA ring, chunks of
 * m - 1 steps of 3 KiB dealt round-robin over the waves -- asynchronous on `stream`; the caller times it (what the
 * memory system gives a read-only kernel, next to the nominal 8 TB/s). */
FMRX_API int fmrx_diag_stream_read_dev(const void *d_buf, size_t bytes, int method, void *stream);
/* replaces estimatePSD  include/fourier.h, src/fourier.cpp:44-128 (with its DFT,
 * :15-23): Bartlett average, in dB, of Hann-windowed nfft-point spectra of
 * `samples`; the reference fixes nfft = NFFT = 512 (include/dy4.h:27).
 * freq[nfft/2] (Hz), psd[nfft/2] (dB).  Requires n >= nfft, nfft even. */
FMRX_API int fmrx_estimate_psd(float *freq, float *psd, const float *samples, size_t n, float Fs, int nfft);

/* ------------------------------------------------------------------ */
/* mode table and pipeline handle                                       */
/* ------------------------------------------------------------------ */
/* replaces struct PARAMS + the mode table  src/project.cpp:17-27, 424-427
 * and the block-size rule :55-57 */
typedef struct fmrx_params {
    int mode;         /* 0..3 */
    int rf_Fs, if_Fs; /* Hz */
    float audio_Fs;
    int rf_decim, audio_decim, audio_upsamp; /* upsamp 0 = integer decimation */
    int rf_taps;
    int audio_taps;   /* already multiplied by audio_upsamp for modes 2, 3 */
    int stereo_taps;
    int block_bytes;  /* the reference's per-mode stdin block size */
} fmrx_params;

/* base_audio_taps: 101 (src/threadMonoOnly.cpp:229-232) or 13
 * (src/project.cpp:424-427); rf_taps: 101 / 151 / 13. */
FMRX_API int fmrx_mode_params(int mode, int rf_taps, int base_audio_taps, int stereo_taps, fmrx_params *p);

typedef struct fmrx_pipeline fmrx_pipeline;

/* PCM overflow policy for s16 outputs */
#define FMRX_PCM_WRAP 1     /* what the compiled reference does */
#define FMRX_PCM_SATURATE 0

/* replaces main()'s setup + the RF_FrontEnd / RF_MONO / RF_STEREO thread
 * bodies  src/project.cpp:40-152, 154-309, 311-382, 385-500.
 * channels: 1 mono, 2 stereo.  max_block_bytes: the largest block that will
 * be passed to process (device buffers are sized once, here).  device: HIP
 * device ordinal.  The handle owns all device memory and the carried state
 * (I/Q FIR history, prev I/Q, audio FIR histories, all-pass delay, PLL). */
FMRX_API int fmrx_pipeline_create(fmrx_pipeline **out, const fmrx_params *p, int channels, size_t max_block_bytes,
                                  int device);
FMRX_API int fmrx_pipeline_destroy(fmrx_pipeline *pl);
/* restore the all-zero initial state of src/project.cpp:61-65, 446-458 */
FMRX_API int fmrx_pipeline_reset(fmrx_pipeline *pl);
FMRX_API size_t fmrx_pipeline_n_if(const fmrx_pipeline *pl, size_t n_bytes);
FMRX_API size_t fmrx_pipeline_n_audio(const fmrx_pipeline *pl, size_t n_bytes);

/* One block, host buffers: iq[n_bytes] interleaved u8 I,Q (the stdin format,
 * src/iofunc.cpp:128-135) -> audio.  Any of the outputs may be NULL.
 *   audio_f32 : mono [n_audio], or stereo left then right planar [2*n_audio]
 *   pcm16     : mono [n_audio], or stereo interleaved L,R [2*n_audio]
 *               (the layout of the writer at src/project.cpp:292-302)
 * n_bytes must satisfy the reference's divisibility rules for the mode
 * (n_bytes/2 % rf_decim == 0, n_if % audio_decim == 0 or
 * n_if*upsamp % decim == 0) and n_bytes/2 >= rf_taps-1. */
FMRX_API int fmrx_pipeline_process(fmrx_pipeline *pl, const uint8_t *iq, size_t n_bytes, float *audio_f32,
                                   int16_t *pcm16, int pcm_policy);
/* The same call in two halves, for block-streaming callers that want the PCIe copies of one block under the kernels of its
 * neighbours: submit enqueues one block (copy in, kernels, copy out) and returns; wait blocks until the OLDEST submitted block's
 * outputs are complete in the host buffers passed to its submit.  At most two blocks are in flight (a third submit waits
 * for the oldest first).  Buffers must stay valid until their block has been waited for; page-locked buffers (fmrx_host_alloc)
 * make the copies truly asynchronous.  fmrx_pipeline_process = submit + wait.  Results are identical to the synchronous calls'. */
FMRX_API int fmrx_pipeline_submit(fmrx_pipeline *pl, const uint8_t *iq, size_t n_bytes, float *audio_f32, int16_t *pcm16,
                                  int pcm_policy);
FMRX_API int fmrx_pipeline_wait(fmrx_pipeline *pl);
/* Same, device-resident: d_iq is DEVICE memory (16-byte aligned), outputs are
 * DEVICE memory or NULL; asynchronous on `stream` (a hipStream_t).  This is
 * the entry point the throughput figures are measured on.  (With option
 * "overlap_calls" the input must be complete at the call, not merely ordered
 * in front of it on `stream`: see the options above.) */
FMRX_API int fmrx_pipeline_process_dev(fmrx_pipeline *pl, const uint8_t *d_iq, size_t n_bytes, float *d_audio_f32,
                                       int16_t *d_pcm16, int pcm_policy, void *stream);
/* Copies of the last block's device intermediates to host, for parity tests
 * and diagnostics.  which: see FMRX_TAP_*.  out may be NULL to query *n. */
#define FMRX_TAP_IF_I 0
#define FMRX_TAP_IF_Q 1
#define FMRX_TAP_DEMOD 2
#define FMRX_TAP_MONO 3
#define FMRX_TAP_CARRIER 4
#define FMRX_TAP_STEREO_BPF 5
#define FMRX_TAP_PLL 6
#define FMRX_TAP_MIXER 7
#define FMRX_TAP_STEREO_FINAL 8
FMRX_API int fmrx_pipeline_read_tap(fmrx_pipeline *pl, int which, float *out, size_t *n);
/* carried state, serialised: floats in the order
 *   I_state[rf_taps-1], Q_state[rf_taps-1], prev_i, prev_q, state_mono[Ha]
 *   (+ stereo: state_stereo[St-1], state_carrier[St-1], state_stereofilt[Ha],
 *    state_allpass[(St-1)/2], state_PLL[6])
 * where Ha = audio history in INPUT samples (audio_taps-1, or
 * (audio_taps-1)/upsamp for modes 2,3).  n = number of floats. */
FMRX_API size_t fmrx_pipeline_state_size(const fmrx_pipeline *pl);
FMRX_API int fmrx_pipeline_get_state(fmrx_pipeline *pl, float *state, size_t n);
FMRX_API int fmrx_pipeline_set_state(fmrx_pipeline *pl, const float *state, size_t n);
/* wall-clock free: device time of the kernels of the last process call, ms,
 * per stage (HIP events on the pipeline's stream).  t[4] = {front_end, audio,
 * stereo_extra, total}. */
FMRX_API int fmrx_pipeline_last_timing(fmrx_pipeline *pl, float *t);
/* sums of the same four figures over the most recent profiled calls (at most
 * max_calls, at most the 128 the handle keeps); *count = calls summed */
FMRX_API int fmrx_pipeline_timing_sum(fmrx_pipeline *pl, float *t, int *count, int max_calls);
/* per-stage HIP events behind last_timing / timing_sum: 0 = off, 1 = around every process call,
 * k > 1 = around every k-th call (an event record costs a few microseconds of stream time) */
FMRX_API int fmrx_pipeline_set_profiling(fmrx_pipeline *pl, int on);
/* The front-end kernels consume the IF I/Q samples in registers, and the fused
 * mono kernel (modes 0/1, one channel, large blocks) the discriminator output
 * too: neither is written to memory.  on = 1 selects the kernels that store
 * them, so that FMRX_TAP_IF_I / FMRX_TAP_IF_Q / FMRX_TAP_DEMOD can be read
 * after a process call (diagnostics and tests; default 0; read_tap returns
 * FMRX_EINVAL for a tap that was not stored). */
FMRX_API int fmrx_pipeline_set_keep_intermediates(fmrx_pipeline *pl, int on);
/* Stereo only.  In the specialised pipeline the pilot PLL (fmPLL, src/filter.cpp:32-80) runs parallel
 * in time (segments with warm-up, checked against the neighbouring segment within a tolerance, serial
 * repair where the loop was not locked); it agrees with the serial recurrence to within the float32
 * grid of its phase argument, not bit for bit (see set_force_generic for the bit-exact mode).  Cumulative since creation: segments that had to
 * be re-run serially, and the largest phase / integrator difference accepted as
 * "merged" at a segment boundary. */
FMRX_API int fmrx_pipeline_pll_diagnostics(fmrx_pipeline *pl, unsigned *repaired_segments, float *max_dphase,
                                           float *max_dinteg);
/* force the parameter-generic kernels (1) or allow the specialised ones (0).  on = 1 is the BIT-EXACT
 * mode: every stage keeps the reference's float32 evaluation order and fmPLL runs as the serial
 * recurrence with glibc's functions, so mono AND stereo audio equal the reference's bit for bit for
 * any stream length (the specialised kernels reorder sums, i.e. differ by ulps, which the stereo
 * recurrence amplifies to its float32 phase grid: DESIGN.md section 2). */
FMRX_API int fmrx_pipeline_set_force_generic(fmrx_pipeline *pl, int on);
/* per-handle run-time option, names as for fmrx_set_option */
FMRX_API int fmrx_pipeline_set_option(fmrx_pipeline *pl, const char *name, long value);

/* ------------------------------------------------------------------ */
/* many mono channels per device call                                   */
/* ------------------------------------------------------------------ */
/* The reference runs one receiver per process (one PARAMS / STATES set, src/project.cpp:460-468); a bank of N
 * receivers is N processes.  fmrx_channels processes the current block of N independent channels in one call
 * (fmrx_channels_create: mono, modes 0 and 1, ONE kernel launch; fmrx_channels_create_ex below: mono or stereo, exact or fast): what a live multi-channel receiver needs, where one 51 200-sample block per
 * channel and launch would leave the chip idle.  A channel's whole carried state is its last ~1 200 input
 * samples, kept as raw bytes in front of its block (csrc/channels.hip); results equal fmrx_pipeline's for the
 * same stream (audio to within float32 summation order, <= 2e-6; PCM +-1 LSB).
 *   block_bytes: bytes per channel and call (a multiple of 16 and of 2*rf_decim*audio_decim; the reference's
 *   p->block_bytes qualifies).  Input: either host memory, channel-major [n_channels][block_bytes]
 *   (fmrx_channels_process), or written by the caller straight into device memory: channel c's block goes to
 *   d_first_block + c*pitch_bytes (fmrx_channels_input_layout), then fmrx_channels_process_dev.
 *   Output: [n_channels][n_audio] float and/or s16, n_audio = fmrx_channels_n_audio(). */
typedef struct fmrx_channels fmrx_channels;
FMRX_API int fmrx_channels_create(fmrx_channels **out, const fmrx_params *p, int n_channels, size_t block_bytes, int device);
/* The same bank with the two choices the reference's command line has (src/project.cpp:390-419: <mode> <channels>) and the
 * numerics mode spelled out:
 *   audio_channels  1 = mono (RF_MONO, src/project.cpp:311-382), 2 = stereo (RF_STEREO, :154-309)
 *   exact           1 = every stage in the reference's float32 evaluation order and fmPLL (src/filter.cpp:32-80) as the
 *                   serial recurrence with glibc's sinf / cosf / atan2f, ONE LANE PER CHANNEL (64 receivers per wave):
 *                   audio is the compiled reference's bit for bit, per channel, for any stream length.  This is the way
 *                   to run stereo within the 1e-4 bound at speed: the recurrence cannot be cut in time without leaving
 *                   the reference's trajectory, but receivers are independent (one STATES set
 *                   each, src/project.cpp:455-468).
 *                   0 = the specialised kernels: mono, fmrx_channels_create's bank (modes 0, 1: one fused kernel); stereo, the matrix-core front end, one fused
 *                   multiply-add per tap in the band-pass pair and the audio FIRs, and the PLL's fast recurrence (closed-form phase
 *                   detector, hardware sine / cosine) walked by one lane per channel -- the error bound of the default
 *                   single-stream stereo path (1e-4 for a stream's first 0.13 s, 0.06 ulp(trigArg(t)) after; mono sum 2e-6)
 *                   at several times the exact bank's rate.
 * Modes: exact banks cover all four modes (0, 1 integer decimation; 2, 3 the rational resampler convolveBlockResampleFIR,
 * src/filter.cpp:191-223, in its own evaluation order; a block must then end on an output boundary: n_if * upsamp % decim == 0,
 * as the reference's own block sizes do), and so do the fast banks: in modes 2 and 3 they run the matrix-core front end (stereo: the fast
 * band-pass pair and PLL) in front of the same batched resampler (mono, exact = 0, modes 2 / 3: audio within 2e-6 of the reference).
 * Outputs: audio_f32 [n_channels][audio_channels][n_audio] (stereo: left, then right), pcm16
 * [n_channels][n_audio][audio_channels] (stereo: interleaved L,R as the writer at src/project.cpp:292-302). */
FMRX_API int fmrx_channels_create_ex(fmrx_channels **out, const fmrx_params *p, int n_channels, int audio_channels, int exact,
                                     size_t block_bytes, int device);
/* exact banks: one channel's intermediates of the last call (FMRX_TAP_DEMOD, _CARRIER, _STEREO_BPF, _PLL [n_if + 1]) */
FMRX_API int fmrx_channels_read_tap(fmrx_channels *c, int channel, int which, float *out, size_t *n);
FMRX_API int fmrx_channels_destroy(fmrx_channels *c);
FMRX_API size_t fmrx_channels_n_audio(const fmrx_channels *c);
FMRX_API int fmrx_channels_input_layout(const fmrx_channels *c, uint8_t **d_first_block, size_t *pitch_bytes);
/* back to the start-of-stream state: one channel, or all of them (channel < 0) */
FMRX_API int fmrx_channels_reset(fmrx_channels *c, int channel);
/* device-resident channel-major blocks [n_channels][block_bytes] -> the slots (one strided device copy, async on `stream`) */
FMRX_API int fmrx_channels_load_dev(fmrx_channels *c, const uint8_t *d_iq, void *stream);
FMRX_API int fmrx_channels_process(fmrx_channels *c, const uint8_t *iq, float *audio_f32, int16_t *pcm16, int pcm_policy);
FMRX_API int fmrx_channels_process_dev(fmrx_channels *c, float *d_audio_f32, int16_t *d_pcm16, int pcm_policy, void *stream);

/* ------------------------------------------------------------------ */
/* RDS path                                          */
/* ------------------------------------------------------------------ */
/* The reference has this path only as a float64 Python / NumPy model: model/fmMonoBlock.py:238-296 on top of
 * model/fmSupportLib.py (it never reached its C++).  Here: float64 HIP kernels for the signal chain -- 54-60 kHz channel
 * band-pass, squarer + 114 kHz band-pass, PLL (ncoScale 0.5, phaseAdjust 3pi/8, bandwidth 0.002), I/Q mixers, rational
 * resampler to sps x 2375 Hz, root-raised-cosine matched filter -- and host C++ for the bit recovery the model does in
 * Python (CDR, Manchester, differential decoding, frame synchronisation).  Input: the discriminator output (fm_demod) of
 * the front end, one block per call, state carried by the handle. */
typedef struct fmrx_rds_params {
    int if_Fs;    /* rate of fm_demod, Hz */
    int taps;     /* band-pass filters: 151 (model/fmMonoBlock.py:114) */
    int upsamp, decim; /* resampler: 247/960 (mode 0), 817/1920 (mode 2) (:74-75, :83-84) */
    int sps;      /* samples per symbol at the resampler output: 26 / 43 */
    int rrc_taps; /* 101 */
} fmrx_rds_params;
typedef struct fmrx_rds fmrx_rds;
FMRX_API int fmrx_rds_mode_params(int mode, fmrx_rds_params *p);   /* the model defines RDS rates for modes 0 and 2 */
FMRX_API int fmrx_rds_create(fmrx_rds **out, const fmrx_rds_params *p, size_t max_block, int device);
FMRX_API int fmrx_rds_destroy(fmrx_rds *r);
FMRX_API int fmrx_rds_reset(fmrx_rds *r);
FMRX_API size_t fmrx_rds_n_out(const fmrx_rds *r, size_t n);   /* n*upsamp/decim */
/* One block of fm_demod (host, n samples; n*upsamp % decim == 0) -> rrc_i / rrc_q [n_out] (matched-filter output, in-phase
 * and quadrature; either may be NULL), bits [<= n_out/sps + 2] = the differentially decoded bits of this block, *n_bits,
 * offset_type [8] = the last offset word the frame synchroniser recognised over the bits kept so far ("A", "B", "C",
 * "C_apos", "D" or " "), exactly as model/fmMonoBlock.py:276-297 reports them per block. */
FMRX_API int fmrx_rds_process(fmrx_rds *r, const float *fm_demod, size_t n, double *rrc_i, double *rrc_q, uint8_t *bits,
                              size_t *n_bits, char *offset_type);
/* the signal chain only, on a device-resident fm_demod (e.g. the pipeline's FMRX_TAP_DEMOD buffer); async on `stream` */
FMRX_API int fmrx_rds_process_dev(fmrx_rds *r, const float *d_demod, size_t n, void *stream);
#define FMRX_RDS_TAP_CHANNEL 0
#define FMRX_RDS_TAP_CARRIER 1
#define FMRX_RDS_TAP_PLL_I 2
#define FMRX_RDS_TAP_PLL_Q 3
#define FMRX_RDS_TAP_RESAMPLED_I 4
#define FMRX_RDS_TAP_RRC_I 5
#define FMRX_RDS_TAP_RRC_Q 6
#define FMRX_RDS_TAP_PLL_STATE 7
FMRX_API int fmrx_rds_read_tap(fmrx_rds *r, int which, double *out, size_t *n);
/* the model's primitives on host buffers (float64): bandPass / impResponse (fmSupportLib.py:358, 376; Python argument
 * order taps, Fs, ...), impulseResponseRootRaisedCosine (:251), CDR incl. Manchester decoding (:103-219; state4 =
 * {pair[0], pair[1], start, prev_size} in/out), diff_decoding (:241), framesync (:30-100) */
FMRX_API int fmrx_rds_band_pass(int taps, double Fs, double Fb, double Fe, double *h);
FMRX_API int fmrx_rds_imp_response(int taps, double Fs, double Fc, double *h);
FMRX_API int fmrx_rds_rrc(double Fs, int taps, double *h);
FMRX_API int fmrx_rds_cdr(const double *x, size_t n, int sps, int block_count, double *state4, uint8_t *bits, size_t *n_bits);
FMRX_API int fmrx_rds_diff_decode(const uint8_t *in, size_t n, uint8_t *out);
FMRX_API int fmrx_rds_frame_sync(const uint8_t *bits, size_t n, char *offset_type, size_t *next_index);

/* ------------------------------------------------------------------ */
/* fused front end (the hot kernel) as a stage of its own               */
/* ------------------------------------------------------------------ */
/* Fused: u8 I/Q -> (u8-128)/128 -> rf low-pass FIR -> decimate, I and Q
 * together.  Replaces readStdinBlockData's conversion (src/iofunc.cpp:133),
 * the I/Q split (src/project.cpp:98-105) and the two convolveBlockFastFIR
 * calls of RF_FrontEnd (src/project.cpp:111,121; src/filter.cpp:158-188).
 *
 * Host buffers, synchronous:
 *   iq[2*n_samples]   interleaved u8 I,Q
 *   hist              in/out, 2*(taps-1) bytes: the taps-1 complex samples that
 *                     precede the block (the reference's I_state/Q_state, kept
 *                     as raw u8); NULL = start of stream (silence), no carry
 *   if_i, if_q        out, n_samples/decim floats each (either may be NULL)
 * force_generic != 0 runs the parameter-generic kernel (reference evaluation
 * order, bit-compatible) instead of the specialised one. */
FMRX_API int fmrx_fe_fir_decim_u8(const uint8_t *iq, size_t n_samples, const float *h, size_t taps, unsigned decim,
                                  uint8_t *hist, float *if_i, float *if_q, int force_generic);

/* Device buffers, asynchronous on a HIP stream: a reusable plan holds the tap
 * tables on the device. */
typedef struct fmrx_fe_plan fmrx_fe_plan;
FMRX_API int fmrx_fe_plan_create(fmrx_fe_plan **out, const float *h, size_t taps, unsigned decim);
FMRX_API int fmrx_fe_plan_destroy(fmrx_fe_plan *plan);
/* 1 when a specialised (register-window, packed-FMA) kernel exists for the
 * plan's (taps, decim); 0 when it will run the generic kernel */
FMRX_API int fmrx_fe_plan_is_specialised(const fmrx_fe_plan *plan);
/* bytes of history kept in front of a block: 2*(taps-1) rounded up to a multiple
 * of 16, plus 16*decim (so the fused kernel can recompute the previous block's
 * last IF samples); the LAST 2*(taps-1) bytes are the reference's I_state/Q_state */
FMRX_API size_t fmrx_fe_plan_history_bytes(const fmrx_fe_plan *plan);
/* d_iq: DEVICE, 16-byte aligned, 2*n_samples bytes.  d_hist: DEVICE,
 * history_bytes bytes, or NULL for silence.  d_if: DEVICE, interleaved float
 * I,Q, n_samples/decim pairs.  stream: hipStream_t (NULL = default stream). */
FMRX_API int fmrx_fe_run_dev(const fmrx_fe_plan *plan, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist,
                             float *d_if, int force_generic, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FMRX_H */

// fmrx_internal.hpp -- shared declarations for libfmrx.so (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "fmrx.h"

namespace fmrx {

// ---- error plumbing -------------------------------------------------------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define FMRX_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::fmrx::fail(FMRX_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                           \
    } while (0)

#define FMRX_TRY(expr)              \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != FMRX_OK) return rc_; \
    } while (0)

// true when at least one HIP device is usable; otherwise sets the error and
// the caller returns FMRX_ENODEV.  There is deliberately no CPU path.
int require_device();

// ---- run-time options ----------------------------------------------------------
// Process-wide defaults: the built-in values, overridden ONCE (first use) by the FMRX_* environment
// variables named below, changed afterwards only through fmrx_set_option().  A pipeline handle copies
// the defaults when it is created and keeps its own set (fmrx_pipeline_set_option); nothing on a
// per-block launch path reads the environment.
struct Options {
    int fe_variant = 0;            // "fe_variant" / FMRX_FE_VARIANT: 0 = matrix-core kernels ("mfma"), 1 = vector-ALU kernels ("valu")
    long fused_min_audio = 65536;  // "fused_min_audio" / FMRX_FUSED_MIN_AUDIO: audio samples per call from which the fused mono kernel runs
    int resample_l2 = 0;           // "resample_l2" / FMRX_RESAMPLE_L2: 1 = L2-table resampler kernel even for large calls
    int resample_exact = 0;        // "resample_exact" / FMRX_RESAMPLE_EXACT: 1 = the pipeline's resampler keeps the reference's rounding sequence
                                   //   (the bit-exact LDS-table kernel instead of the matrix-core one)
    int overlap_calls = 0;         // "overlap_calls" / FMRX_OVERLAP_CALLS: 1 = the caller vouches that a process_dev call's input is complete when the call
                                   //   is made (data resident in HBM): the stereo pipeline then runs front end, PLL and output stage of consecutive
                                   //   calls on three internal streams, one call apart each; the caller's stream still waits for each call's output
    int resample_chains = 0;       // "resample_chains" / FMRX_RESAMPLE_CHAINS: workgroups per XCD and tile group of the matrix-core resampler
                                   //   (0 = as many as are resident at once); A/B knob
    int fe_wgs_per_cu = 0;         // "fe_wgs_per_cu" / FMRX_FE_WGS_PER_CU: cap on resident workgroups per CU of the front-end kernels (0 = auto)
    int pll_warmup = -1;           // "pll_warmup" / FMRX_PLL_WARMUP: warm-up samples per lane of the parallel PLL (-1 = built-in)
    int pll_segment = -1;          // "pll_segment" / FMRX_PLL_SEGMENT: samples per lane (-1 = built-in)
    int pll_align = 0;             // "pll_align" / FMRX_PLL_ALIGN: 1 = lanes start on a multiple of the loop's period, 0 = exactly W early (default)
    int pll_start = 1;             // "pll_start" / FMRX_PLL_START: where the parallel PLL's lanes start: 1 (default) = the state of the locked loop as a
                                   //   linear system of the input's signs, then 64 true warm-up steps; 0 = the block's initial state plus drift, 512 steps
    int pll_head = -1;             // "pll_head" / FMRX_PLL_HEAD: samples of a stream's first call walked serially (-1 = built-in)
    int pll_mode = 0;              // "pll_mode" / FMRX_PLL_MODE: stereo PLL of the specialised pipeline: 0 = parallel in time, fast math
                                   //   (default); 1 = serial, fast math; 2 = serial, glibc math (the cause-by-cause variants of DESIGN 2)
    int bank_streams = 3;          // "bank_streams" / FMRX_BANK_STREAMS: fast stereo banks: 3 = front end | band-pass pair + output stage | PLL lanes on
                                   //   three internal streams (default), 2 = the front end on the same stream as the other wide kernels
    int bank_fused = 0;            // "bank_fused" / FMRX_BANK_FUSED: fast stereo banks: 1 = front end + band-pass pair in one kernel (int8 + f32 matrix
                                   //   cores: measured slower, kernels_fe_mfma.hip), 0 = two kernels (matrix-core front end, vector-ALU band-pass pair; default)
    int bank_fe_wgs_fused = 2;     // "bank_fe_wgs_fused": workgroups per CU of that fused kernel
    int bank_fe_wgs = 1;           // "bank_fe_wgs" / FMRX_BANK_FE_WGS: workgroups per CU of the bank's matrix-core front end (1 leaves registers and
                                   //   LDS for the kernels that run next to it; 0 = as many as fit)
    int demod = 0;                 // "demod" / FMRX_DEMOD: 0 = the C++ reference's discriminator (fmDemod, src/filter.cpp:248-266; default),
                                   //   1 = the Python model's arctangent demodulator (fmDemodArctan, model/fmSupportLib.py:502-531), float64
    int fused_tune = 0;            // "fused_tune", "fe_mfma_tune": ablation variants, honoured only by a -DFMRX_TUNING build
    int fe_mfma_tune = 0;
};
Options &default_options();
std::mutex &options_mutex();     // guards writes to the process-wide defaults and the copy a new handle takes
Options options_snapshot();      // the defaults, copied under the mutex: what handle constructors use
// name -> field; returns false for an unknown name
bool option_ref(Options &o, const char *name, long **as_long, int **as_int);
int set_option_in(Options &o, const char *name, long value);   // validates; FMRX_EINVAL for unknown names / values

// ---- small RAII device buffer ----------------------------------------------
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) return fail(FMRX_ENOMEM, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
        n = count;
        return FMRX_OK;
    }
    int ensure(size_t count) { return count <= n ? FMRX_OK : alloc(count); }
    size_t bytes() const { return n * sizeof(T); }
};

// ---- front-end fast path (kernels_fe.hip) -----------------------------------
// Device-side tap table for the register-window FIR: phase-major, padded,
// pre-scaled.  See kernels_fe.hip for the layout.
struct AudioPlan;
struct FePlan {
    int taps = 0, decim = 0;
    bool fast = false;         // a specialised kernel exists for (taps, decim)
    int hist_bytes = 0;        // bytes of u8 history the kernel reads before the block (multiple of 16)
    DevBuf<float> table;       // fast-path table
    DevBuf<float> h;           // plain taps (generic path)
    // matrix-core kernel (kernels_fe_mfma.hip): taps as int8 digit fragments + the scale that undoes them
    bool mfma = false;
    DevBuf<int32_t> a_img;
    float scale_lo = 0.0f;
    DevBuf<uint8_t> silence;   // hist_bytes bytes of 128: the history of a stream that starts here
};
constexpr int kFeMfmaDigits = 3;   // base-256 digits per tap: 24-bit fixed point
int fe_plan_init(FePlan &pl, const float *h, int taps, int decim);
// Matrix-core front end + discriminator: same contract as fe_demod_launch; d_demod may be NULL when
// only the IF stream is wanted (d_if != NULL).
int fe_mfma_plan_init(FePlan &pl, const float *h, int taps, int decim);
bool fe_mfma_available(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist);
int fe_mfma_launch(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, const float *d_prev,
                   float *d_demod, float *d_if, float *d_prev_out, uint8_t *d_hist_next, const Options &o, hipStream_t stream,
                   const float *d_dhist_src = nullptr, float *d_dhist_dst = nullptr, int dhist_n = 0);
// (d_dhist_src -> d_dhist_dst, dhist_n floats: the discriminator history copied in front of this block's output by the kernel itself)
// the same kernel over a bank of receivers' slots (channels_stereo.hip): see kernels_fe_mfma.hip
int fe_mfma_bank_lead(const FePlan &pl);   // bytes of the stream a row needs in front of its block
int fe_mfma_bank_launch(const FePlan &pl, const uint8_t *d_slots, long total_bytes, long in_pitch, long block_off, int n_channels,
                        long k_lo, long k_hi, float *d_demod, long out_pitch, long out_off, int wgs_per_cu_cap, hipStream_t stream);
// front end + both stereo band-pass filters in one kernel over the bank's rows (int8 + f32 matrix cores): kernels_fe_mfma.hip
bool fe_bpf_bank_available(const FePlan &pl, int stereo_taps);
int fe_bpf_tables_init(DevBuf<float> &st_img, DevBuf<float> &car_img, const float *h_st, const float *h_car, int taps);
int fe_bpf_bank_launch(const FePlan &pl, int stereo_taps, const float *d_st_img, const float *d_car_img, const uint8_t *d_slots,
                       long total_bytes, long in_pitch, long block_off, int n_channels, long k_lo, long k_hi, float *d_demod,
                       long out_pitch, long out_off, float *d_bpf, long bpf_pitch, int8_t *d_car8, long car_pitch, int wgs_per_cu_cap,
                       hipStream_t stream);
// d_hist: hist_bytes bytes whose LAST 2*(taps-1) hold the previous samples.
// Writes n_samples/decim float2 (I,Q) to d_if.
int fe_launch(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, float *d_if,
              const Options &o, hipStream_t stream, bool force_generic);
int fe_hist_bytes(int taps, int decim);
// Fused front end + discriminator (the pipeline's kernel).  d_demod[n/decim] is
// written; d_if (interleaved I,Q) and d_prev_out (float2 = IF[n/decim-1]) are
// optional; d_prev_override (float2) replaces the recomputed IF[-1] when given.
bool fe_fused_available(const FePlan &pl, const uint8_t *d_iq, size_t n_samples);
// d_hist_next (optional, hist_bytes bytes, requires 2*n_samples >= hist_bytes): the kernel also leaves
// the block's last bytes there for the next block.
int fe_demod_launch(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist,
                    const float *d_prev_override, float *d_demod, float *d_if, float *d_prev_out,
                    uint8_t *d_hist_next, const Options &o, hipStream_t stream);

// ---- audio fast path (kernels_audio.hip) --------------------------------------
struct AudioPlan {
    int taps = 0, decim = 0;
    bool fast = false;
    DevBuf<float> table;
    DevBuf<float> h;
    DevBuf<float> mfma_table;  // Toeplitz image for the fused mono kernel (kernels_fe_mfma.hip)
};
// Fused mono chain (modes 0/1): u8 I/Q -> audio / PCM in one kernel; the discriminator output stays
// on chip except its last tail_keep samples, written to d_demod_tail[n_if - tail_keep .. n_if) for the
// next block.  d_dhist_end: one past the previous block's last discriminator sample; d_prev: its
// last IF sample (float2).
int audio_mfma_table_init(AudioPlan &pl, const float *h, int taps, int decim);
bool mono_fused_available(const FePlan &fe, const AudioPlan &au, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist);
int mono_fused_launch(const FePlan &fe, const AudioPlan &au, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist,
                      const float *d_prev, const float *d_dhist_end, float *d_demod_tail, int tail_keep, float *d_prev_out,
                      float *d_audio, int16_t *d_pcm, int wrap, uint8_t *d_hist_next, const Options &o, hipStream_t stream);
int audio_plan_init(AudioPlan &pl, const float *h, int taps, int decim);
// y[k] = sum_n h[n] * x[decim*k - n - delay]; x points at the block start and
// x[-(taps-1+delay+3) .. -1] must be readable history (the specialised kernel
// loads 16-byte chunks).
// Optionally also writes s16 PCM (d_pcm != nullptr).  d_hist_end (optional, specialised kernel only):
// one past the last sample of the previous block; when given, samples before d_x[0] are read from
// d_hist_end[-1], d_hist_end[-2], ... instead of d_x[-1], d_x[-2], ...
bool audio_fast_available(const AudioPlan &pl, const float *d_x);
int audio_fir_launch(const AudioPlan &pl, const float *d_x, const float *d_hist_end, size_t n_in, int delay, float *d_y,
                     int16_t *d_pcm, int wrap, hipStream_t stream, bool force_generic);

// ---- rational resampler (kernels_resample.hip) ------------------------------------
struct ResamplePlan {
    int taps = 0, decim = 0, upsamp = 0;
    int J = 0, JP = 0;         // taps per polyphase row, padded row length
    int span = 0;              // input samples one 256-output tile stages in LDS
    bool fast = false;
    // LDS-resident table kernel: the taps are applied in npass passes of W taps (all phases x W taps
    // fit LDS); tiles of 1024 outputs stage span_l inputs
    int npass = 0, W = 0, span_l = 0;
    DevBuf<float> table;       // polyphase-major taps [upsamp][JP]
    DevBuf<float> h;           // plain taps (generic path)
    // matrix-core kernel (pipeline path): tap image [tile][lane][K-step], K index 0 of every tile, tile groups
    bool mfma = false;
    int mfma_ks4 = 0, mfma_ngroups = 0, mfma_pieces = 0;
    bool mfma_reach_ok = false;   // piece staging stays within kResampleFront / kResampleBack of the block (else: element staging)
    mutable int mfma_wgs_per_cu[2] = {0, 0};   // resident workgroups per CU of the kernel instance in use (piece / element staging), asked once
    DevBuf<float> mfma_img;
    DevBuf<int> mfma_top, mfma_groups;   // per tile: K index 0's input offset; per group: m0, m1, lo, pieces
};
int resample_plan_init(ResamplePlan &pl, const float *h, int taps, int decim, int upsamp);
// exact: only the kernels that keep the reference's rounding sequence (the primitive's contract)
// d_pcm: also pack s16 PCM (d_y may then be null where resample_mfma_available(): that kernel writes either or both)
// margins: the caller vouches for finite, readable samples kResampleFront floats in front of d_x - delay and kResampleBack
// behind its n_in samples (the matrix-core kernel then stages every block alike; only taps that are zero meet those samples)
constexpr int kResampleFront = 256, kResampleBack = 1024;
int resample_launch(const ResamplePlan &pl, const float *d_x, size_t n_in, int delay, float *d_y, const Options &o,
                    hipStream_t stream, bool force_generic, bool exact = false, int16_t *d_pcm = nullptr, int wrap = 0,
                    bool margins = false);
bool resample_mfma_available(const ResamplePlan &pl, const float *d_x, size_t n_in, int delay, const Options &o);

// ---- stereo band-pass pair (kernels_stereo.hip) ------------------------------------
struct BpfPairPlan {
    int taps = 0;
    bool fast = false;
    DevBuf<float> table;        // interleaved reversed tap pairs (h_stereo, h_carrier)
    DevBuf<float> h_st, h_car;  // plain taps (generic path)
};
int bpf_pair_plan_init(BpfPairPlan &pl, const float *h_stereo, const float *h_carrier, int taps);
int bpf_pair_launch(const BpfPairPlan &pl, const float *d_x, size_t n, float *d_st, float *d_car, hipStream_t stream,
                    bool force_generic);

// everything behind the PLL of modes 0/1 in one kernel: mixer, both audio FIRs (mono branch on the all-passed
// discriminator output, stereo branch on the mixer output), L/R combine, PCM (kernels_stereo.hip)
bool stereo_out_available(int taps, int decim);
int stereo_out_launch(const float *d_demod, const float *d_bpf, const float *d_nco, const float *d_mix_tail_in,
                      float *d_mix_tail_out, int hm, size_t n_if, int delay, const float *d_h, int taps, int decim, float *d_mono,
                      float *d_st, float *d_left, float *d_right, int16_t *d_pcm, int wrap, float *d_mixer, hipStream_t stream);

// ---- generic kernels (kernels_generic.hip) ----------------------------------
// y[k] = sum_{n<taps} h[n]*x[decim*k - n], sequential mul+add in n (bit-compatible
// with the reference's evaluation order).  x[-(taps-1)..-1] must be readable.
int k_fir_generic(const float *d_x, size_t n_out, const float *d_h, int taps, int decim, float *d_y, hipStream_t s);
// same on interleaved u8 I/Q with (u-128)/128 fused; writes float2 (I,Q)
int k_fe_generic(const uint8_t *d_iq, const uint8_t *d_hist, int hist_bytes, size_t n_samples, const float *d_h,
                 int taps, int decim, float *d_if, hipStream_t s);
// polyphase resampler in stream form; d_x[-(hist)..-1] readable, hist=(taps-1)/upsamp
int k_resample_generic(const float *d_x, size_t n_in, const float *d_h, int taps, int decim, int upsamp, float *d_y,
                       hipStream_t s);
// demod[k] from interleaved IF (I,Q); IF[-1] = *d_prev (float2). Also stores IF[n-1] to d_prev_out when non-null.
// fast != 0: the arithmetic of the fused audio kernel (demod_fast) instead of the reference order
int k_fm_demod_if(const float *d_if, size_t n, const float *d_prev, float *d_prev_out, float *d_demod, int fast,
                  hipStream_t s);
int k_fm_demod_planar(const float *d_i, const float *d_q, size_t n, float prev_i, float prev_q, float *d_demod,
                      hipStream_t s);
// the model's arctangent demodulator (fmSupportLib.py:502-531), float64: out[k] = unwrap(atan2(Q[k], I[k]) - phase of the sample in
// front); planar double in / double out with the running phase `prev_phase` in front of sample 0 (stage API), and interleaved
// float IF in / float out with IF[-1] = *d_prev (pipeline option "demod" = 1)
int k_fm_demod_arctan_planar(const double *d_i, const double *d_q, size_t n, double prev_phase, double *d_out, hipStream_t s);
int k_fm_demod_arctan_if(const float *d_if, size_t n, const float *d_prev, float *d_demod, hipStream_t s);
int k_u8_to_f32(const uint8_t *d_raw, size_t n, float *d_out, hipStream_t s);
int k_deinterleave(const float *d_iq, size_t n_pairs, float *d_i, float *d_q, hipStream_t s);
int k_split_if(const float *d_if, size_t n, float *d_i, float *d_q, hipStream_t s);
int k_pcm16(const float *d_a, size_t n, int16_t *d_out, int wrap, hipStream_t s);
int k_pcm16_stereo(const float *d_l, const float *d_r, size_t n, int16_t *d_out, int wrap, hipStream_t s);
int k_all_pass(const float *d_in, size_t n, const float *d_state, size_t nstate, float *d_out, hipStream_t s);
// ---- pilot PLL (kernels_pll.hip) ----
// serial form.  fast == 0: the reference's recurrence with glibc's sinf/cosf/atan2f (glibc_libm.hpp) -- bit-identical
// to the reference given bit-identical input; fast != 0: shared double-precision argument reduction + hardware sin/cos
int k_fm_pll(const float *d_in, size_t n, float *d_out, float *d_state, float freq, float Fs, float ncoScale,
             float phaseAdjust, float normBandwidth, int fast, hipStream_t s);
// parallel-in-time form (fast math): segments with warm-up, each checked against its predecessor's end state
// within the merge tolerance below, serial repair where the loop was not locked.  Agrees with the serial
// trajectory to within the float32 grid of trigArg, not bit for bit (kernels_pll.hip).  d_scratch:
// pll_parallel_scratch_floats(n) floats; d_scratch[2] (as u32) counts segments that needed a repair (diagnostic).
// lane shape: L samples per lane, started at the last multiple of the loop's period that is >= W samples early
constexpr int kPllSegment = 64, kPllWarmup = 512, kPllSegmentMin = 32;
constexpr int kPllWarmupLti = 64;
// merge tolerance on the integrator in that mode, in Ki * ulp(trigArg): lanes 64 true steps from a noise-free start still differ
// from their neighbours by the loop's response to the float32 grid (2-3 Ki ulp), which the envelope of DESIGN section 2 contains
constexpr float kPllIntegTolUlpsLti = 6.0f;   // true steps a lane runs in front of its segment when it starts from the linear system's state (pll_start = 1)
constexpr int kPllHead = 1024;   // samples of a stream's first call walked serially (acquisition) before the lanes take over
// merge tolerance between a lane's warmed-up state and the true state (see kernels_pll.hip)
constexpr float kPllTolPhase = 1e-2f, kPllTolInteg = 1e-4f;
size_t pll_parallel_scratch_floats(size_t n);
int k_fm_pll_parallel(const float *d_in, size_t n, float *d_out, float *d_state, float freq, float Fs, float ncoScale,
                      float phaseAdjust, float normBandwidth, float *d_scratch, const Options &o, hipStream_t s,
                      double off_hint = -1.0,    // off_hint: IF samples of the stream in front of this call (the state's trigOffset), < 0 = unknown
                      int phases = 3, float *d_lti = nullptr);
// phases: 1 = only what depends on the input alone (the linear system's chunk records), 2 = the lanes and the repair, 3 = both;
// d_lti: where the chunk records live (pll_parallel_lti_floats(n) floats, 8-byte aligned) if not inside d_scratch -- a caller
// that runs phase 1 of its next call on another stream while phase 2 of this one reads them keeps two
size_t pll_parallel_lti_floats(size_t n);
// many channels, lane = channel, the exact serial recurrence (kernels_pll.hip); rows [channel][pitch], state 8 floats per channel
int k_fm_pll_channels(const float *d_in, long pitch_in, size_t n, int n_ch, float *d_trig, long pitch_trig, float *d_state,
                      float *d_nco0, float freq, float Fs, float ncoScale, float phaseAdjust, float normBandwidth, hipStream_t s,
                      bool flat = true,    // flat: the branch-free forms of glibc's functions (same values; false = A/B)
                      bool exact = true,   // false: the fast recurrence (closed-form phase detector, hardware sine / cosine) of the specialised path
                      bool in8 = false);   // fast only: d_in holds signed bytes (the input's sign; pitch_in in bytes)
int k_mix(const float *d_bpf, const float *d_pll, size_t n, float *d_mix, hipStream_t s);
int k_combine(const float *d_st, const float *d_mono, size_t n, float *d_l, float *d_r, hipStream_t s);
int k_upsample(const float *d_x, size_t n, float *d_xu, int up, hipStream_t s);
int k_downsample(const float *d_in, size_t n_out, float *d_out, int ds, hipStream_t s);
int k_fill_u8(uint8_t *d, size_t n, uint8_t v, hipStream_t s);
// diagnostics: out[i] = sinf / cosf / atan2f (fn 0 / 1 / 2) of a[i] (, b[i]) as the device evaluates glibc_libm.hpp
int k_libm_eval(int fn, const float *d_a, const float *d_b, size_t n, float *d_out, hipStream_t s);

// ---- measurement aid (kernels_diag.hip): one pure streaming read of the buffer; method 0 registers (non-temporal), 1 LDS-DMA ring
int k_stream_read(const void *d_buf, size_t bytes, int method, unsigned *d_sink, hipStream_t s);

// ---- diagnostics (kernels_psd.hip) ---------------------------------------------------
// d_seg_db: (n/nfft)*(nfft/2) floats of scratch; d_freq, d_psd: nfft/2 floats
int k_estimate_psd(const float *d_x, size_t n, float Fs, int nfft, float *d_seg_db, float *d_freq, float *d_psd, hipStream_t s);

// ---- banks of receivers in the reference's evaluation order (channels_stereo.hip) ----
struct StereoBank;
bool stereo_bank_supported(const fmrx_params &p, int audio_channels);
int stereo_bank_create(StereoBank **out, const fmrx_params &p, int n_channels, int audio_channels, int exact, size_t block_bytes);
void stereo_bank_destroy(StereoBank *b);
size_t stereo_bank_n_audio(const StereoBank *b);
uint8_t *stereo_bank_first_block(const StereoBank *b);
size_t stereo_bank_pitch(const StereoBank *b);
int stereo_bank_reset(StereoBank *b, int channel);
int stereo_bank_process_dev(StereoBank *b, float *d_audio, int16_t *d_pcm, int wrap, hipStream_t s);
int stereo_bank_read_tap(StereoBank *b, int channel, int which, float *out, size_t *n);

// ---- host-side coefficient design (coeff.cpp) --------------------------------
void design_lpf(float Fs, float Fc, int taps, float *h);
void design_bpf(float Fs, float Fb, float Fe, int taps, float *h);

}  // namespace fmrx

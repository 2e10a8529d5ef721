/*
 * fm_oracle.h -- CPU restatement of the reference FM-receiver hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 * The shipped path (libfmrx.so, HIP kernels) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * the reference's own C++ sources compiled in the build container
 * (oracle/_ref/libfmref.so, recipe: oracle/Makefile) by
 * tests/test_oracle_vs_ref.py, and against the committed golden vectors in
 * tests/golden/ (generated from the compiled reference by
 * tests/golden/make_golden.py).  The reference has no golden vectors or
 * known-answer tests of its own for this path.
 *
 * Each function cites the reference file:line it restates.  All arithmetic
 * is IEEE binary32 with the evaluation order of the reference built with
 * g++ -O3 on x86-64 (no FMA contraction): compile with -ffp-contract=off.
 */
#ifndef FM_ORACLE_H
#define FM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- coefficient design (host) ------------------------------------- */
/* src/filter.cpp:103-114 impulseResponseLPF */
void fmo_impulse_response_lpf(float Fs, float Fc, unsigned short num_taps, float *h);
/* src/filter.cpp:83-99 bandPass */
void fmo_band_pass(float Fs, float Fb, float Fe, unsigned short num_taps, float *h);

/* ---- I/O conversions ------------------------------------------------ */
/* src/iofunc.cpp:128-135 readStdinBlockData (the conversion, not the read) */
void fmo_u8_to_f32(const uint8_t *raw, size_t n, float *out);
/* src/project.cpp:98-105 de-interleave (even -> I, odd -> Q) */
void fmo_deinterleave(const float *iq, size_t n_pairs, float *I, float *Q);
/* src/threadMonoOnly.cpp:185-191 PCM pack. wrap!=0: int32 truncation then low
 * 16 bits (what the compiled reference does on overflow); wrap==0: saturate. */
void fmo_pcm16(const float *audio, size_t n, int16_t *out, int wrap);

/* ---- FIR family ------------------------------------------------------ */
/* src/filter.cpp:118-130 convolveFIR; y has n+taps-1 elements */
void fmo_convolve_fir(float *y, const float *x, size_t n, const float *h, size_t taps);
/* src/filter.cpp:133-154 convolveBlockFIR; state has taps-1 elements (in/out) */
void fmo_convolve_block_fir(float *y, const float *x, size_t n, const float *h, size_t taps, float *state);
/* src/filter.cpp:158-188 convolveBlockFastFIR; y has n/decim elements.
 * The reference's extra out-of-bounds iteration at m==n is not
 * reproduced: its result lands past the end of y and is discarded. */
void fmo_convolve_block_fast_fir(float *y, const float *x, size_t n, const float *h, size_t taps,
                                 float *state, unsigned decim);
/* src/filter.cpp:191-223 convolveBlockResampleFIR; y has (n*upsamp)/decim
 * elements; state has taps-1 elements in the UPSAMPLED index space, of which
 * only slots == upsamp-1 (mod upsamp) are ever read or written. */
void fmo_convolve_block_resample_fir(float *y, const float *x, size_t n, const float *h, size_t taps,
                                     float *state, unsigned decim, unsigned upsamp);
/* src/filter.cpp:227-234 upsample; xu has n*up elements */
void fmo_upsample(const float *x, size_t n, float *xu, int up);
/* src/filter.cpp:237-245 downsample; returns number of outputs written
 * (ceil(n / (float)ds)) */
size_t fmo_downsample(float *out, const float *in, size_t n, unsigned short ds);

/* ---- demod / stereo -------------------------------------------------- */
/* src/filter.cpp:248-266 fmDemod */
void fmo_fm_demod(float *out, const float *I, const float *Q, size_t n, float *prev_i, float *prev_q);
/* src/filter.cpp:14-29 allPass; state has nstate elements, requires n >= nstate */
void fmo_all_pass(const float *in, size_t n, float *state, size_t nstate, float *out);
/* src/filter.cpp:32-80 fmPLL; nco_out has n+1 elements; state has 6 floats
 * {integrator, phaseEst, feedbackI, feedbackQ, lastOut, trigOffset} */
void fmo_fm_pll(const float *in, size_t n, float *nco_out, float *state, float freq, float Fs,
                float ncoScale, float phaseAdjust, float normBandwidth);

/* ---- mode table + whole pipelines ----------------------------------- */
typedef struct {
    int mode;
    int rf_Fs, if_Fs;
    float audio_Fs;
    int rf_decim, audio_decim, audio_upsamp;
    int rf_taps, audio_taps /* already multiplied by upsamp for modes 2,3 */, stereo_taps;
    int block_bytes; /* src/project.cpp:55-57 */
} fmo_params;

/* src/project.cpp:424-427 mode table; base_audio_taps is 101
 * (threadMonoOnly.cpp:229-232) or 13 (project.cpp:424-427).
 * Returns 0, or -1 for an invalid mode. */
int fmo_mode_params(int mode, int rf_taps, int base_audio_taps, int stereo_taps, fmo_params *p);

typedef struct fmo_pipeline fmo_pipeline;
/* channels: 1 = RF_FrontEnd + RF_MONO (project.cpp:40-152, 311-382),
 *           2 = RF_FrontEnd + RF_STEREO (project.cpp:154-309) */
fmo_pipeline *fmo_pipeline_create(const fmo_params *p, int channels);
void fmo_pipeline_destroy(fmo_pipeline *pl);
/* Process n_bytes of interleaved u8 I/Q (any length meeting the reference's
 * divisibility preconditions).  Outputs (each may be NULL):
 *   demod   [n_if]        FM discriminator output
 *   audio_l [n_audio]     mono audio (channels==1) or left
 *   audio_r [n_audio]     right (channels==2 only)
 * Returns n_audio. */
size_t fmo_pipeline_process(fmo_pipeline *pl, const uint8_t *iq, size_t n_bytes,
                            float *if_i, float *if_q, float *demod, float *audio_l, float *audio_r);
/* number of IF / audio samples produced for n_bytes of input */
size_t fmo_pipeline_n_if(const fmo_pipeline *pl, size_t n_bytes);
size_t fmo_pipeline_n_audio(const fmo_pipeline *pl, size_t n_bytes);
/* access to stereo intermediates of the LAST processed block (valid until the
 * next call); which: 0 carrier_filt, 1 stereo_filt, 2 pll (n_if+1), 3 mixer,
 * 4 allpass, 5 mono_filt, 6 stereo_final.  Returns length. */
size_t fmo_pipeline_intermediate(const fmo_pipeline *pl, int which, const float **ptr);

/* ---- diagnostics ----------------------------------- */
/* src/fourier.cpp:44-128 estimatePSD: Bartlett average of Hann-windowed
 * nfft-point DFTs (src/fourier.cpp:15-23), in dB; nfft = NFFT = 512
 * (include/dy4.h:27).  freq[nfft/2], psd[nfft/2]; needs n >= nfft.
 * Returns the number of segments averaged. */
int fmo_estimate_psd(float *freq, float *psd, const float *samples, size_t n, float Fs, int nfft);

/* out[i] = sinf(a[i]) (fn 0), cosf(a[i]) (fn 1), atan2f(a[i], b[i]) (fn 2) of this host's C library:
 * what fmPLL's std::sin / std::cos / std::atan2 (src/filter.cpp:55,69-71) resolve to.  Lets a GPU test
 * compare the device's restatement of these functions (csrc/glibc_libm.hpp) with the real ones. */
void fmo_libm(int fn, const float *a, const float *b, size_t n, float *out);

/* ---- deterministic synthetic FM multiplex ---------------- */
/* Fills iq[2*n_samples] with constant-envelope stereo-multiplex FM at rf_Fs,
 * starting at absolute sample index start (so consecutive calls continue the
 * same stream).  Pure function of (rf_Fs, seed, start). */
void fmo_synth_fm_u8(uint8_t *iq, size_t n_samples, double rf_Fs, uint64_t seed, uint64_t start);

#ifdef __cplusplus
}
#endif
#endif /* FM_ORACLE_H */

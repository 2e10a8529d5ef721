/*
 * fm_oracle.c -- CPU restatement of the reference FM-receiver hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see fm_oracle.h).  Parity status: PINNED against
 * the compiled reference (oracle/_ref) and tests/golden/.
 *
 * Written from the behaviour of /root/reference (file:line cited per
 * function), not copied from it: raw pointers + explicit lengths instead of
 * std::vector, the stream-form index arithmetic made explicit, and the
 * reference's out-of-bounds iteration dropped.  Build: oracle/Makefile
 * (gcc -O3 -ffp-contract=off, mirrors src/Makefile:4 "-O3", no -march, no
 * -ffast-math).
 */
#include "fm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* include/dy4.h:23 -- a double literal; every expression touching it is
 * evaluated in double and rounded to float only when stored. */
#define FMO_PI 3.14159265358979323846

/* ------------------------------------------------------------------ */
/* coefficient design                                                   */
/* ------------------------------------------------------------------ */

/* src/filter.cpp:103-114.  Two float stores per tap: the sinc value, then the
 * product with the sin^2 window (window uses i*pi/T, not T-1). */
void fmo_impulse_response_lpf(float Fs, float Fc, unsigned short num_taps, float *h)
{
    const int T = (int)num_taps;
    const int c = (T - 1) / 2;
    const float norm_fc = Fc / (Fs / 2); /* float / (float / int->float) */
    for (int i = 0; i < T; i++) {
        float v;
        if (i == c) {
            v = norm_fc;
        } else {
            const double arg = FMO_PI * norm_fc * (i - c);
            v = (float)(norm_fc * (sin(arg) / arg));
        }
        const double s = sin(i * FMO_PI / T);
        /* std::pow(double,int 2): g++ -O3 folds to s*s (verified against the
         * compiled reference bit-for-bit in tests/test_oracle_vs_ref.py) */
        v = (float)(v * (s * s));
        h[i] = v;
    }
}

/* src/filter.cpp:83-99.  Three float stores per tap. */
void fmo_band_pass(float Fs, float Fb, float Fe, unsigned short num_taps, float *h)
{
    const int T = (int)num_taps;
    const int c = (T - 1) / 2;
    const float norm_center = ((Fe + Fb) / 2) / (Fs / 2);
    const float norm_pass = (Fe - Fb) / (Fs / 2);
    for (int i = 0; i < T; i++) {
        float v;
        if (i == c) {
            v = norm_pass;
        } else {
            const double arg = FMO_PI * norm_pass / 2 * (i - c);
            v = (float)(norm_pass * (sin(arg) / arg));
        }
        v = (float)(v * cos(i * (FMO_PI) * norm_center));
        const double s = sin(i * FMO_PI / T);
        v = (float)(v * s * s); /* (v*s)*s, left to right, in double */
        h[i] = v;
    }
}

/* ------------------------------------------------------------------ */
/* I/O conversions                                                      */
/* ------------------------------------------------------------------ */

/* src/iofunc.cpp:133  float(((unsigned char)raw - 128) / 128.0): exact. */
void fmo_u8_to_f32(const uint8_t *raw, size_t n, float *out)
{
    for (size_t k = 0; k < n; k++)
        out[k] = (float)(((int)raw[k] - 128) / 128.0);
}

/* src/project.cpp:98-105: the toggling predicate sends element 0 to I. */
void fmo_deinterleave(const float *iq, size_t n_pairs, float *I, float *Q)
{
    for (size_t k = 0; k < n_pairs; k++) {
        I[k] = iq[2 * k];
        Q[k] = iq[2 * k + 1];
    }
}

/* src/threadMonoOnly.cpp:185-191: NaN -> 0 else (short)(a*16384).  The C++
 * cast is undefined out of range; the compiled reference (cvttss2si + low
 * word) wraps, which wrap!=0 reproduces. */
void fmo_pcm16(const float *audio, size_t n, int16_t *out, int wrap)
{
    for (size_t k = 0; k < n; k++) {
        const float a = audio[k];
        if (isnan(a)) {
            out[k] = 0;
            continue;
        }
        const float s = a * 16384;
        if (wrap) {
            int32_t v;
            if (s >= 2147483648.0f || s < -2147483648.0f)
                v = INT32_MIN; /* cvttss2si "integer indefinite" */
            else
                v = (int32_t)s;
            out[k] = (int16_t)(uint16_t)(uint32_t)v;
        } else {
            if (s >= 32767.0f) out[k] = 32767;
            else if (s <= -32768.0f) out[k] = -32768;
            else out[k] = (int16_t)s;
        }
    }
}

/* ------------------------------------------------------------------ */
/* FIR family                                                           */
/* ------------------------------------------------------------------ */

/* src/filter.cpp:118-130.  With unsigned m,n the guard reduces to
 * 0 <= m-n < n_x; terms are accumulated for n ascending from +0.0f. */
void fmo_convolve_fir(float *y, const float *x, size_t n, const float *h, size_t taps)
{
    const size_t ny = n + taps - 1;
    for (size_t m = 0; m < ny; m++) {
        float acc = 0.0f;
        for (size_t k = 0; k < taps; k++) {
            if (k <= m && m - k < n)
                acc += h[k] * x[m - k];
        }
        y[m] = acc;
    }
}

/* stream sample j of the current block: x[j] for j>=0, else the carried
 * state (state[j + ns]), ns = taps-1  (src/filter.cpp:141-144, 168-174) */
static inline float fmo_tap_src(const float *x, const float *state, long ns, long j)
{
    return j >= 0 ? x[j] : state[j + ns];
}

/* src/filter.cpp:148-153 / 182-187: state <- last ns samples of x */
static void fmo_refresh_state(float *state, size_t ns, const float *x, size_t n)
{
    for (size_t k = 0; k < ns; k++)
        state[k] = x[n - ns + k];
}

/* src/filter.cpp:133-154 */
void fmo_convolve_block_fir(float *y, const float *x, size_t n, const float *h, size_t taps, float *state)
{
    const long ns = (long)taps - 1;
    for (long m = 0; m < (long)n; m++) {
        float acc = 0.0f;
        for (long k = 0; k < (long)taps; k++)
            acc += h[k] * fmo_tap_src(x, state, ns, m - k);
        y[m] = acc;
    }
    fmo_refresh_state(state, (size_t)ns, x, n);
}

/* src/filter.cpp:158-188 (loop bound taken as m < n/decim*decim: see header) */
void fmo_convolve_block_fast_fir(float *y, const float *x, size_t n, const float *h, size_t taps,
                                 float *state, unsigned decim)
{
    const long ns = (long)taps - 1;
    const size_t ny = n / decim;
    for (size_t o = 0; o < ny; o++) {
        const long m = (long)(o * decim);
        float acc = 0.0f;
        for (long k = 0; k < (long)taps; k++)
            acc += h[k] * fmo_tap_src(x, state, ns, m - k);
        y[o] = acc;
    }
    fmo_refresh_state(state, (size_t)ns, x, n);
}

/* src/filter.cpp:191-223.  m walks the upsampled timeline in steps of decim;
 * only taps n == m (mod upsamp) meet a non-zero upsampled sample.  In-block
 * the sample is x[(m-n)/upsamp]; before the block it is
 * state[m-n+ns] (upsampled index space).  Gain: y += y*upsamp (:213). */
void fmo_convolve_block_resample_fir(float *y, const float *x, size_t n, const float *h, size_t taps,
                                     float *state, unsigned decim, unsigned upsamp)
{
    const long ns = (long)taps - 1;
    const long U = (long)upsamp, D = (long)decim;
    const long total = (long)n * U;
    for (long m = 0; m < total; m += D) {
        const long phase = m % U;
        float acc = 0.0f;
        for (long k = phase; k < (long)taps; k += U) {
            if (m - k >= 0)
                acc += h[k] * x[(m - k) / U];
            else
                acc += h[k] * state[m - k + ns];
        }
        acc += acc * (float)upsamp; /* int -> float conversion then float mul */
        y[m / D] = acc;
    }
    /* :218-222  k = U-1; for i = U*n - ns; i < U*n - U; i += U: state[k] = x[i/U + 1] */
    long k = U - 1;
    for (long i = U * (long)n - ns; i < U * (long)n - U; i += U) {
        state[k] = x[(i / U) + 1];
        k += U;
    }
}

/* src/filter.cpp:227-234 */
void fmo_upsample(const float *x, size_t n, float *xu, int up)
{
    for (size_t i = 0; i < n * (size_t)up; i++)
        xu[i] = (i % (size_t)up) == 0 ? x[i / (size_t)up] : 0.0f;
}

/* src/filter.cpp:237-245: size = ceil(n / (float)ds) evaluated in float */
size_t fmo_downsample(float *out, const float *in, size_t n, unsigned short ds)
{
    const size_t ny = (size_t)ceil((float)n / (float)ds);
    for (size_t i = 0; i < ny; i++)
        out[i] = in[i * ds];
    return ny;
}

/* ------------------------------------------------------------------ */
/* demod / stereo                                                       */
/* ------------------------------------------------------------------ */

/* src/filter.cpp:248-266.  All float; den = I*I + Q*Q (two rounded products,
 * one rounded add); numerator I*(Q-Qp) - Q*(I-Ip); one divide. */
void fmo_fm_demod(float *out, const float *I, const float *Q, size_t n, float *prev_i, float *prev_q)
{
    float pi = *prev_i, pq = *prev_q;
    for (size_t k = 0; k < n; k++) {
        const float i = I[k], q = Q[k];
        const float den = i * i + q * q;
        if (den == 0) {
            out[k] = 0;
        } else {
            out[k] = (i * (q - pq) - q * (i - pi)) / den;
        }
        pi = i;
        pq = q;
    }
    if (n) {
        *prev_i = I[n - 1];
        *prev_q = Q[n - 1];
    }
}

/* src/filter.cpp:14-29: out = [state, in[0 .. n-ns)], state <- in[n-ns .. n) */
void fmo_all_pass(const float *in, size_t n, float *state, size_t nstate, float *out)
{
    /* out may not alias in; build output before refreshing the state */
    for (size_t k = 0; k < nstate; k++)
        out[k] = state[k];
    for (size_t k = nstate; k < n; k++)
        out[k] = in[k - nstate];
    for (size_t k = 0; k < nstate; k++)
        state[k] = in[n - nstate + k];
}

/* src/filter.cpp:32-80.  float recurrences; trigArg evaluated in double
 * (2*PI is double) and rounded to float; atan2f/cosf/sinf float overloads. */
void fmo_fm_pll(const float *in, size_t n, float *nco_out, float *state, float freq, float Fs,
                float ncoScale, float phaseAdjust, float normBandwidth)
{
    const float Cp = 2.666f; /* float Cp = 2.666 (double literal rounded) */
    const float Ci = 3.555f;
    const float Kp = normBandwidth * Cp;
    const float Ki = (normBandwidth * normBandwidth) * Ci;

    float integrator = state[0];
    float phaseEst = state[1];
    float feedbackI = state[2];
    float feedbackQ = state[3];
    nco_out[0] = state[4];
    float trigOffset = state[5];

    const float f_ratio = freq / Fs; /* (freq/Fs) is a float division */
    for (size_t k = 0; k < n; k++) {
        const float errorI = in[k] * feedbackI;
        const float errorQ = in[k] * (-1 * feedbackQ);
        const float errorD = atan2f(errorQ, errorI);
        integrator = integrator + Ki * errorD;
        phaseEst = phaseEst + Kp * errorD + integrator;
        trigOffset += 1;
        const float trigArg = (float)(2 * FMO_PI * f_ratio * trigOffset + phaseEst);
        feedbackI = cosf(trigArg);
        feedbackQ = sinf(trigArg);
        nco_out[k + 1] = cosf(trigArg * ncoScale + phaseAdjust);
    }
    state[0] = integrator;
    state[1] = phaseEst;
    state[2] = feedbackI;
    state[3] = feedbackQ;
    state[4] = nco_out[n];
    state[5] = trigOffset;
}

/* ------------------------------------------------------------------ */
/* mode table + pipelines                                               */
/* ------------------------------------------------------------------ */

/* src/project.cpp:424-427 (values), :55-57 (block size) */
int fmo_mode_params(int mode, int rf_taps, int base_audio_taps, int stereo_taps, fmo_params *p)
{
    memset(p, 0, sizeof(*p));
    p->mode = mode;
    p->rf_taps = rf_taps;
    p->stereo_taps = stereo_taps;
    switch (mode) {
    case 0: p->rf_Fs = 2400000; p->if_Fs = 240000; p->audio_Fs = 48000.0f; p->rf_decim = 10; p->audio_decim = 5;    p->audio_upsamp = 0;   break;
    case 1: p->rf_Fs = 1440000; p->if_Fs = 288000; p->audio_Fs = 48000.0f; p->rf_decim = 5;  p->audio_decim = 6;    p->audio_upsamp = 0;   break;
    case 2: p->rf_Fs = 2400000; p->if_Fs = 240000; p->audio_Fs = 44100.0f; p->rf_decim = 10; p->audio_decim = 800;  p->audio_upsamp = 147; break;
    case 3: p->rf_Fs = 960000;  p->if_Fs = 320000; p->audio_Fs = 44100.0f; p->rf_decim = 3;  p->audio_decim = 3200; p->audio_upsamp = 441; break;
    default: return -1;
    }
    p->audio_taps = p->audio_upsamp ? base_audio_taps * p->audio_upsamp : base_audio_taps;
    if (mode == 0 || mode == 1)
        p->block_bytes = 1024 * p->rf_decim * p->audio_decim * 2;
    else
        p->block_bytes = 7 * p->audio_decim * p->rf_decim * 2;
    return 0;
}

struct fmo_pipeline {
    fmo_params p;
    int channels;
    float *rf_coeff, *audio_coeff, *carrier_coeff, *stereo_coeff;
    float *i_state, *q_state;
    float prev_i, prev_q;
    float *state_mono, *state_stereo, *state_carrier, *state_stereofilt, *state_allpass;
    float state_pll[6];
    /* scratch for the last block */
    size_t cap_in;
    float *iq, *I, *Q, *If, *Qf, *demod;
    float *allpass, *stereo_filt, *carrier_filt, *pll, *mixer, *mono_filt, *stereo_final;
    size_t n_if_last, n_audio_last;
};

static float *fmo_zeros(size_t n)
{
    return (float *)calloc(n ? n : 1, sizeof(float));
}

fmo_pipeline *fmo_pipeline_create(const fmo_params *p, int channels)
{
    fmo_pipeline *pl = (fmo_pipeline *)calloc(1, sizeof(*pl));
    pl->p = *p;
    pl->channels = channels;
    /* project.cpp:50  impulseResponseLPF(rf_Fs, 100000, rf_taps) (ints -> float) */
    pl->rf_coeff = fmo_zeros((size_t)p->rf_taps);
    fmo_impulse_response_lpf((float)p->rf_Fs, (float)100000, (unsigned short)p->rf_taps, pl->rf_coeff);
    /* project.cpp:321-323: Fs = if_fs or if_fs*audio_upsamp (int product -> float) */
    pl->audio_coeff = fmo_zeros((size_t)p->audio_taps);
    const int audio_design_fs = p->audio_upsamp ? p->if_Fs * p->audio_upsamp : p->if_Fs;
    fmo_impulse_response_lpf((float)audio_design_fs, (float)16000, (unsigned short)p->audio_taps, pl->audio_coeff);
    pl->i_state = fmo_zeros((size_t)p->rf_taps - 1);
    pl->q_state = fmo_zeros((size_t)p->rf_taps - 1);
    pl->state_mono = fmo_zeros((size_t)p->audio_taps - 1);
    if (channels == 2) {
        /* project.cpp:172-173, 446-458 */
        pl->carrier_coeff = fmo_zeros((size_t)p->stereo_taps);
        pl->stereo_coeff = fmo_zeros((size_t)p->stereo_taps);
        fmo_band_pass((float)p->if_Fs, (float)18.5e3, (float)19.5e3, (unsigned short)p->stereo_taps, pl->carrier_coeff);
        fmo_band_pass((float)p->if_Fs, (float)22e3, (float)54e3, (unsigned short)p->stereo_taps, pl->stereo_coeff);
        pl->state_stereo = fmo_zeros((size_t)p->stereo_taps - 1);
        pl->state_carrier = fmo_zeros((size_t)p->stereo_taps - 1);
        pl->state_stereofilt = fmo_zeros((size_t)p->audio_taps - 1);
        pl->state_allpass = fmo_zeros((size_t)((p->stereo_taps - 1) / 2));
        const float init[6] = {0.0f, 0.0f, 1.0f, 0.0f, 1.0f, 0.0f};
        memcpy(pl->state_pll, init, sizeof(init));
    }
    return pl;
}

void fmo_pipeline_destroy(fmo_pipeline *pl)
{
    if (!pl) return;
    float *ptrs[] = {pl->rf_coeff, pl->audio_coeff, pl->carrier_coeff, pl->stereo_coeff, pl->i_state, pl->q_state,
                     pl->state_mono, pl->state_stereo, pl->state_carrier, pl->state_stereofilt, pl->state_allpass,
                     pl->iq, pl->I, pl->Q, pl->If, pl->Qf, pl->demod, pl->allpass, pl->stereo_filt, pl->carrier_filt,
                     pl->pll, pl->mixer, pl->mono_filt, pl->stereo_final};
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++)
        free(ptrs[i]);
    free(pl);
}

size_t fmo_pipeline_n_if(const fmo_pipeline *pl, size_t n_bytes)
{
    return (n_bytes / 2) / (size_t)pl->p.rf_decim;
}

size_t fmo_pipeline_n_audio(const fmo_pipeline *pl, size_t n_bytes)
{
    const size_t n_if = fmo_pipeline_n_if(pl, n_bytes);
    if (pl->p.audio_upsamp)
        return (n_if * (size_t)pl->p.audio_upsamp) / (size_t)pl->p.audio_decim;
    return n_if / (size_t)pl->p.audio_decim;
}

static void fmo_reserve(fmo_pipeline *pl, size_t n_bytes)
{
    if (n_bytes <= pl->cap_in) return;
    const size_t ns = n_bytes / 2, n_if = fmo_pipeline_n_if(pl, n_bytes);
    size_t n_audio = fmo_pipeline_n_audio(pl, n_bytes);
#define FMO_RE(ptr, cnt) do { free(ptr); ptr = fmo_zeros(cnt); } while (0)
    FMO_RE(pl->iq, n_bytes);
    FMO_RE(pl->I, ns);
    FMO_RE(pl->Q, ns);
    FMO_RE(pl->If, n_if);
    FMO_RE(pl->Qf, n_if);
    FMO_RE(pl->demod, n_if);
    FMO_RE(pl->mono_filt, n_audio + 1);
    if (pl->channels == 2) {
        FMO_RE(pl->allpass, n_if);
        FMO_RE(pl->stereo_filt, n_if);
        FMO_RE(pl->carrier_filt, n_if);
        FMO_RE(pl->pll, n_if + 1);
        FMO_RE(pl->mixer, n_if);
        FMO_RE(pl->stereo_final, n_audio + 1);
    }
#undef FMO_RE
    pl->cap_in = n_bytes;
}

/* audio-rate stage shared by mono and both stereo branches:
 * project.cpp:344-357 / 217-231 / 255-268 */
static void fmo_audio_stage(const fmo_pipeline *pl, float *y, const float *x, size_t n, float *state)
{
    if (pl->p.audio_upsamp)
        fmo_convolve_block_resample_fir(y, x, n, pl->audio_coeff, (size_t)pl->p.audio_taps, state,
                                        (unsigned)pl->p.audio_decim, (unsigned)pl->p.audio_upsamp);
    else
        fmo_convolve_block_fast_fir(y, x, n, pl->audio_coeff, (size_t)pl->p.audio_taps, state,
                                    (unsigned)pl->p.audio_decim);
}

size_t fmo_pipeline_process(fmo_pipeline *pl, const uint8_t *iq, size_t n_bytes,
                            float *if_i, float *if_q, float *demod, float *audio_l, float *audio_r)
{
    fmo_reserve(pl, n_bytes);
    const size_t ns = n_bytes / 2;
    const size_t n_if = fmo_pipeline_n_if(pl, n_bytes);
    const size_t n_audio = fmo_pipeline_n_audio(pl, n_bytes);
    pl->n_if_last = n_if;
    pl->n_audio_last = n_audio;

    /* RF_FrontEnd: project.cpp:82, 98-105, 111, 121, 128 */
    fmo_u8_to_f32(iq, n_bytes, pl->iq);
    fmo_deinterleave(pl->iq, ns, pl->I, pl->Q);
    fmo_convolve_block_fast_fir(pl->If, pl->I, ns, pl->rf_coeff, (size_t)pl->p.rf_taps, pl->i_state, (unsigned)pl->p.rf_decim);
    fmo_convolve_block_fast_fir(pl->Qf, pl->Q, ns, pl->rf_coeff, (size_t)pl->p.rf_taps, pl->q_state, (unsigned)pl->p.rf_decim);
    fmo_fm_demod(pl->demod, pl->If, pl->Qf, n_if, &pl->prev_i, &pl->prev_q);
    if (if_i) memcpy(if_i, pl->If, n_if * sizeof(float));
    if (if_q) memcpy(if_q, pl->Qf, n_if * sizeof(float));
    if (demod) memcpy(demod, pl->demod, n_if * sizeof(float));

    if (pl->channels == 1) {
        /* RF_MONO: project.cpp:344-357 */
        fmo_audio_stage(pl, pl->mono_filt, pl->demod, n_if, pl->state_mono);
        if (audio_l) memcpy(audio_l, pl->mono_filt, n_audio * sizeof(float));
        return n_audio;
    }

    /* RF_STEREO: project.cpp:194-280 */
    fmo_all_pass(pl->demod, n_if, pl->state_allpass, (size_t)((pl->p.stereo_taps - 1) / 2), pl->allpass);
    fmo_convolve_block_fir(pl->stereo_filt, pl->demod, n_if, pl->stereo_coeff, (size_t)pl->p.stereo_taps, pl->state_stereo);
    fmo_convolve_block_fir(pl->carrier_filt, pl->demod, n_if, pl->carrier_coeff, (size_t)pl->p.stereo_taps, pl->state_carrier);
    fmo_audio_stage(pl, pl->mono_filt, pl->allpass, n_if, pl->state_mono);
    /* project.cpp:237  fmPLL(carrier_filt, PLL, state, 19e3, if_fs, 2.0, 0.0, 0.01) */
    fmo_fm_pll(pl->carrier_filt, n_if, pl->pll, pl->state_pll, (float)19e3, (float)pl->p.if_Fs, (float)2.0, (float)0.0, (float)0.01);
    /* project.cpp:246-248  mixer[z] = stereo_filt[z]*PLL[z]*2  (float*float, then *int) */
    for (size_t z = 0; z < n_if; z++)
        pl->mixer[z] = pl->stereo_filt[z] * pl->pll[z] * 2;
    fmo_audio_stage(pl, pl->stereo_final, pl->mixer, n_if, pl->state_stereofilt);
    /* project.cpp:277-280 */
    for (size_t s = 0; s < n_audio; s++) {
        if (audio_l) audio_l[s] = pl->stereo_final[s] + pl->mono_filt[s];
        if (audio_r) audio_r[s] = pl->mono_filt[s] - pl->stereo_final[s];
    }
    return n_audio;
}

size_t fmo_pipeline_intermediate(const fmo_pipeline *pl, int which, const float **ptr)
{
    switch (which) {
    case 0: *ptr = pl->carrier_filt; return pl->n_if_last;
    case 1: *ptr = pl->stereo_filt; return pl->n_if_last;
    case 2: *ptr = pl->pll; return pl->n_if_last + 1;
    case 3: *ptr = pl->mixer; return pl->n_if_last;
    case 4: *ptr = pl->allpass; return pl->n_if_last;
    case 5: *ptr = pl->mono_filt; return pl->n_audio_last;
    case 6: *ptr = pl->stereo_final; return pl->n_audio_last;
    default: *ptr = NULL; return 0;
    }
}

void fmo_libm(int fn, const float *a, const float *b, size_t n, float *out)
{
    for (size_t i = 0; i < n; i++) out[i] = fn == 0 ? sinf(a[i]) : fn == 1 ? cosf(a[i]) : atan2f(a[i], b[i]);
}

/* ------------------------------------------------------------------ */
/* diagnostics: Bartlett PSD estimate                                     */
/* ------------------------------------------------------------------ */

/* src/fourier.cpp:44-128 with its DFT (:15-23) inlined.  float32 throughout
 * except where a double literal forces double: the twiddle angle
 * -2*PI*(k*m)/N is evaluated in double and rounded to float (it is the
 * imaginary part of a std::complex<float>), std::exp(complex<float>) is
 * (cosf, sinf) of that float (e^0 == 1), the bin power goes through
 * std::pow(float, 2.0) in double.  Segments are averaged in dB. */
int fmo_estimate_psd(float *freq, float *psd, const float *samples, size_t n, float Fs, int nfft)
{
    const int half = nfft / 2;
    const float df = Fs / nfft;
    /* LinearSpacedArray(freq, Fs/2, 0.0, df): N = (max-min)/step in float, i < N */
    {
        const float N = (Fs / 2 - 0.0f) / df;
        for (int i = 0; i < half; i++)
            freq[i] = (i < N) ? 0.0f + i * df : 0.0f;
    }
    float *hann = (float *)malloc(sizeof(float) * (size_t)nfft);
    for (int i = 0; i < nfft; i++) {
        const double sn = sin(i * FMO_PI / nfft);
        hann[i] = (float)pow(sn, 2.0);
    }
    const int nseg = (int)floor((double)((float)n / (float)nfft));
    float *acc = (float *)calloc((size_t)half, sizeof(float));
    float *w = (float *)malloc(sizeof(float) * (size_t)nfft);
    for (int sgm = 0; sgm < nseg; sgm++) {
        for (int i = 0; i < nfft; i++)
            w[i] = samples[(size_t)sgm * nfft + i] * hann[i];
        for (int m = 0; m < half; m++) {
            float re = 0.0f, im = 0.0f;
            for (int k = 0; k < nfft; k++) {
                const float ang = (float)(-2 * FMO_PI * (unsigned)(k * m) / (unsigned)nfft);
                /* x * exp(i*ang): libstdc++ scales real and imaginary part by the real factor */
                re += w[k] * cosf(ang);
                im += w[k] * sinf(ang);
            }
            const float mag = hypotf(re, im);                 /* std::abs(complex<float>) */
            float p = (float)((1 / (Fs * nfft / 2)) * pow((double)mag, 2.0));
            p = 2 * p;
            p = 10 * log10f(p);
            acc[m] += p;    /* psd_est[k] += psd_list[...] in segment order */
        }
    }
    /* the reference sums bin k over segments l = 0.. in order, then divides */
    for (int m = 0; m < half; m++)
        psd[m] = acc[m] / nseg;
    free(w);
    free(acc);
    free(hann);
    return nseg;
}

/* ------------------------------------------------------------------ */
/* synthetic FM multiplex, closed form so that any window of  */
/* the stream can be generated independently                              */
/* ------------------------------------------------------------------ */

static inline uint64_t fmo_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* m(t) = 0.45(L+R) + 0.1 cos(2pi 19k t) + 0.45 (L-R) cos(2pi 38k t),
 * L = 0.5(cos 2pi 1k t + cos 2pi 3k t), R = cos 2pi 2k t.
 * phi(t) = 2pi * 75k * integral_0^t m, written as a sum of sines. */
static double fmo_synth_phase(double t)
{
    const double w = 2.0 * FMO_PI;
    /* cosine components of m(t): amplitude, frequency */
    static const double comp[][2] = {
        {0.45 * 0.5, 1e3}, {0.45 * 0.5, 3e3}, {0.45 * 1.0, 2e3}, /* L+R */
        {0.1, 19e3},                                               /* pilot */
        /* (L-R) cos(38k): cos a cos b = 0.5 cos(a-b) + 0.5 cos(a+b) */
        {0.45 * 0.25, 38e3 - 1e3}, {0.45 * 0.25, 38e3 + 1e3},
        {0.45 * 0.25, 38e3 - 3e3}, {0.45 * 0.25, 38e3 + 3e3},
        {-0.45 * 0.5, 38e3 - 2e3}, {-0.45 * 0.5, 38e3 + 2e3},
    };
    double integ = 0.0;
    for (size_t i = 0; i < sizeof(comp) / sizeof(comp[0]); i++)
        integ += comp[i][0] * sin(w * comp[i][1] * t) / (w * comp[i][1]);
    return w * 75e3 * integ;
}

void fmo_synth_fm_u8(uint8_t *iq, size_t n_samples, double rf_Fs, uint64_t seed, uint64_t start)
{
    for (size_t k = 0; k < n_samples; k++) {
        const uint64_t idx = start + k;
        /* the multiplex is periodic in 1 ms: when that is a whole number of
         * samples reduce the index first so t stays small and exact */
        const uint64_t per = (uint64_t)(rf_Fs / 1000.0);
        const double t = ((double)per * 1000.0 == rf_Fs) ? (double)(idx % per) / rf_Fs : (double)idx / rf_Fs;
        const double phi = fmo_synth_phase(t);
        const double vi = 0.8 * cos(phi), vq = 0.8 * sin(phi);
        const uint64_t r = fmo_splitmix64(seed ^ (idx * 0xD1342543DE82EF95ull));
        const double di = ((double)(r & 0xFFFFFFFFull) / 4294967296.0) - 0.5;
        const double dq = ((double)(r >> 32) / 4294967296.0) - 0.5;
        double qi = floor(128.0 + 127.0 * vi + di + 0.5);
        double qq = floor(128.0 + 127.0 * vq + dq + 0.5);
        qi = qi < 0 ? 0 : (qi > 255 ? 255 : qi);
        qq = qq < 0 ? 0 : (qq > 255 ? 255 : qq);
        iq[2 * k] = (uint8_t)qi;
        iq[2 * k + 1] = (uint8_t)qq;
    }
}

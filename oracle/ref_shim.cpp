/*
 * ref_shim.cpp -- extern "C" doorway onto the REAL reference functions.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with
 * the reference's own src/filter.cpp and src/iofunc.cpp, taken where they lie
 * under /root/reference (never copied), into oracle/_ref/libfmref.so by
 * oracle/Makefile.  It exists so tests can (1) validate oracle/fm_oracle.c
 * bit-for-bit against the reference and (2) generate tests/golden/ vectors.
 * It exists only in the build container: /root/reference is absent on the GPU
 * box, and oracle/_ref/ is git-ignored AND gpurun-ignored -- nothing derived from
 * the reference travels; there the tests run on the oracle and the committed fixtures.
 *
 * The reference's thread bodies (project.cpp RF_FrontEnd/RF_MONO/RF_STEREO)
 * loop forever and exit(1) at EOF, so the block pipelines below replay their
 * call sequence (project.cpp:98-128, 194-280, 344-357) with the reference's
 * primitives, single-threaded and deterministic.
 */
#include <cstdint>
#include <cstring>
#include <sstream>
#include <streambuf>
#include <vector>

#include "dy4.h"
#include "filter.h"
#include "fourier.h"
#include "iofunc.h"

namespace {
// convolveBlockFastFIR iterates once past the end (m == x.size()); give the
// vectors one element of slack so that stray iteration stays inside owned
// memory.  Values are unaffected.
std::vector<float> vec_slack(const float *p, size_t n)
{
    std::vector<float> v;
    v.reserve(n + 1);
    v.assign(p, p + n);
    return v;
}
struct membuf : std::streambuf {
    membuf(const char *b, size_t n) { char *p = const_cast<char *>(b); setg(p, p, p + n); }
};
}  // namespace

extern "C" {

void ref_impulse_response_lpf(float Fs, float Fc, unsigned short taps, float *h)
{
    std::vector<float> v;
    impulseResponseLPF(Fs, Fc, taps, v);
    std::memcpy(h, v.data(), v.size() * sizeof(float));
}

void ref_band_pass(float Fs, float Fb, float Fe, unsigned short taps, float *h)
{
    std::vector<float> v;
    bandPass(Fs, Fb, Fe, taps, v);
    std::memcpy(h, v.data(), v.size() * sizeof(float));
}

void ref_convolve_fir(float *y, const float *x, size_t n, const float *h, size_t taps)
{
    std::vector<float> vy, vx(x, x + n), vh(h, h + taps);
    convolveFIR(vy, vx, vh);
    std::memcpy(y, vy.data(), vy.size() * sizeof(float));
}

void ref_convolve_block_fir(float *y, const float *x, size_t n, const float *h, size_t taps, float *state)
{
    std::vector<float> vy, vx(x, x + n), vh(h, h + taps), vs(state, state + taps - 1);
    convolveBlockFIR(vy, vx, vh, vs);
    std::memcpy(y, vy.data(), vy.size() * sizeof(float));
    std::memcpy(state, vs.data(), vs.size() * sizeof(float));
}

void ref_convolve_block_fast_fir(float *y, const float *x, size_t n, const float *h, size_t taps, float *state,
                                 unsigned decim)
{
    std::vector<float> vy, vx = vec_slack(x, n), vh(h, h + taps), vs(state, state + taps - 1);
    vy.reserve(n / decim + 2);
    convolveBlockFastFIR(vy, vx, vh, vs, decim, false);
    std::memcpy(y, vy.data(), vy.size() * sizeof(float));
    std::memcpy(state, vs.data(), vs.size() * sizeof(float));
}

void ref_convolve_block_resample_fir(float *y, const float *x, size_t n, const float *h, size_t taps, float *state,
                                     unsigned decim, unsigned upsamp)
{
    std::vector<float> vy, vx(x, x + n), vh(h, h + taps), vs(state, state + taps - 1);
    convolveBlockResampleFIR(vy, vx, vh, vs, decim, upsamp, false);
    std::memcpy(y, vy.data(), vy.size() * sizeof(float));
    std::memcpy(state, vs.data(), vs.size() * sizeof(float));
}

void ref_upsample(const float *x, size_t n, float *xu, int up)
{
    std::vector<float> vx(x, x + n), vu;
    upsample(vx, vu, up);
    std::memcpy(xu, vu.data(), vu.size() * sizeof(float));
}

size_t ref_downsample(float *out, const float *in, size_t n, unsigned short ds)
{
    std::vector<float> vi(in, in + n), vo;
    downsample(vo, vi, ds);
    std::memcpy(out, vo.data(), vo.size() * sizeof(float));
    return vo.size();
}

void ref_fm_demod(float *out, const float *I, const float *Q, size_t n, float *prev_i, float *prev_q)
{
    std::vector<float> vo, vi(I, I + n), vq(Q, Q + n);
    fmDemod(vo, vi, vq, *prev_i, *prev_q);
    std::memcpy(out, vo.data(), vo.size() * sizeof(float));
}

void ref_all_pass(const float *in, size_t n, float *state, size_t nstate, float *out)
{
    std::vector<float> vi(in, in + n), vs(state, state + nstate), vo;
    allPass(vi, vs, vo);
    std::memcpy(out, vo.data(), vo.size() * sizeof(float));
    std::memcpy(state, vs.data(), vs.size() * sizeof(float));
}

void ref_fm_pll(const float *in, size_t n, float *nco_out, float *state, float freq, float Fs, float ncoScale,
                float phaseAdjust, float normBandwidth)
{
    std::vector<float> vi(in, in + n), vo, vs(state, state + 6);
    fmPLL(vi, vo, vs, freq, Fs, ncoScale, phaseAdjust, normBandwidth);
    std::memcpy(nco_out, vo.data(), vo.size() * sizeof(float));
    std::memcpy(state, vs.data(), 6 * sizeof(float));
}

// readStdinBlockData reads std::cin: point cin at the caller's bytes.
void ref_read_block(const uint8_t *raw, size_t n, float *out)
{
    membuf mb(reinterpret_cast<const char *>(raw), n);
    std::streambuf *old = std::cin.rdbuf(&mb);
    std::cin.clear();
    std::vector<float> v(n);
    readStdinBlockData((unsigned int)n, 0, v);
    std::cin.rdbuf(old);
    std::cin.clear();
    std::memcpy(out, v.data(), n * sizeof(float));
}

// (short)(a*16384) exactly as threadMonoOnly.cpp:188-189 is compiled by g++ -O3
void ref_pcm16(const float *a, size_t n, int16_t *out)
{
    std::vector<short int> wav(n);
    for (unsigned int k = 0; k < n; k++) {
        if (std::isnan(a[k])) wav[k] = 0;
        else wav[k] = static_cast<short int>(a[k] * 16384);
    }
    std::memcpy(out, wav.data(), n * sizeof(short int));
}

int ref_estimate_psd(float *freq, float *psd, const float *samples, size_t n, float Fs)
{
    std::vector<float> f, p, v(samples, samples + n);
    estimatePSD(f, p, v, Fs);
    std::memcpy(freq, f.data(), f.size() * sizeof(float));
    std::memcpy(psd, p.data(), p.size() * sizeof(float));
    return (int)p.size();
}

// ---- block pipeline replaying project.cpp with the reference primitives ----
struct ref_pipeline {
    int mode, channels;
    int rf_Fs, if_fs, rf_decim, audio_decim, audio_upsamp, audio_taps, stereo_taps, rf_taps;
    std::vector<float> rf_coeff, audio_coeff, carrier_coeff, stereo_coeff;
    std::vector<float> I_state, Q_state;
    float prev_i = 0.0, prev_q = 0.0;
    std::vector<float> state_mono, state_stereo, state_carrier, state_stereofilt, state_allpass, state_PLL;
    // last-block intermediates
    std::vector<float> I_filt, Q_filt, fm_demod, audio_allpass, stereo_filt, carrier_filt, audio_filt, PLL, mixer,
        stereo_final, L, R;
};

ref_pipeline *ref_pipeline_create(int mode, int channels, int rf_taps, int base_audio_taps, int stereo_taps)
{
    ref_pipeline *p = new ref_pipeline;
    p->mode = mode;
    p->channels = channels;
    p->rf_taps = rf_taps;
    p->stereo_taps = stereo_taps;
    int audio_taps = base_audio_taps;
    // the values of project.cpp:424-427
    if (mode == 1) { p->rf_Fs = 1440000; p->if_fs = 288000; p->rf_decim = 5; p->audio_decim = 6; p->audio_upsamp = 0; }
    else if (mode == 2) { p->rf_Fs = 2400000; p->if_fs = 240000; p->rf_decim = 10; p->audio_decim = 800; p->audio_upsamp = 147; audio_taps = base_audio_taps * 147; }
    else if (mode == 3) { p->rf_Fs = 960000; p->if_fs = 320000; p->rf_decim = 3; p->audio_decim = 3200; p->audio_upsamp = 441; audio_taps = base_audio_taps * 441; }
    else { p->rf_Fs = 2400000; p->if_fs = 240000; p->rf_decim = 10; p->audio_decim = 5; p->audio_upsamp = 0; }
    p->audio_taps = audio_taps;
    int rf_Fc = 100000, audio_Fc = 16000;
    impulseResponseLPF(p->rf_Fs, rf_Fc, rf_taps, p->rf_coeff);
    if (mode == 0 || mode == 1) impulseResponseLPF(p->if_fs, audio_Fc, audio_taps, p->audio_coeff);
    else impulseResponseLPF(p->if_fs * p->audio_upsamp, audio_Fc, audio_taps, p->audio_coeff);
    p->I_state.resize(rf_taps - 1, 0.0);
    p->Q_state.resize(rf_taps - 1, 0.0);
    p->state_mono.resize(audio_taps - 1, 0.0f);
    if (channels == 2) {
        bandPass(p->if_fs, 18.5e3, 19.5e3, stereo_taps, p->carrier_coeff);
        bandPass(p->if_fs, 22e3, 54e3, stereo_taps, p->stereo_coeff);
        p->state_stereo.resize(stereo_taps - 1, 0.0f);
        p->state_carrier.resize(stereo_taps - 1, 0.0f);
        p->state_stereofilt.resize(audio_taps - 1, 0.0f);
        p->state_allpass.resize(int((stereo_taps - 1) / 2), 0.0f);
        p->state_PLL = std::vector<float>{0.0, 0.0, 1.0, 0.0, 1.0, 0};
    }
    return p;
}

// The same graph at parameter values outside the reference's mode table (BASELINE configs[2]: the course
// spec's 2.5 MS/s -> 250 kS/s -> 48 / 40 kS/s, U/D = 24/125 / 4/25): project.cpp's setup with these numbers.
// audio_taps is the total count (base taps x upsamp for a resampling mode).
ref_pipeline *ref_pipeline_create_params(int rf_Fs, int if_fs, int rf_decim, int audio_decim, int audio_upsamp, int rf_taps,
                                         int audio_taps, int stereo_taps, int channels)
{
    ref_pipeline *p = new ref_pipeline;
    p->mode = audio_upsamp > 0 ? 2 : 0;   // selects the audio stage: resampler / decimating FIR
    p->channels = channels;
    p->rf_taps = rf_taps;
    p->stereo_taps = stereo_taps;
    p->rf_Fs = rf_Fs; p->if_fs = if_fs; p->rf_decim = rf_decim; p->audio_decim = audio_decim; p->audio_upsamp = audio_upsamp;
    p->audio_taps = audio_taps;
    int rf_Fc = 100000, audio_Fc = 16000;
    impulseResponseLPF(p->rf_Fs, rf_Fc, rf_taps, p->rf_coeff);
    if (audio_upsamp == 0) impulseResponseLPF(p->if_fs, audio_Fc, audio_taps, p->audio_coeff);
    else impulseResponseLPF(p->if_fs * p->audio_upsamp, audio_Fc, audio_taps, p->audio_coeff);
    p->I_state.resize(rf_taps - 1, 0.0);
    p->Q_state.resize(rf_taps - 1, 0.0);
    p->state_mono.resize(audio_taps - 1, 0.0f);
    if (channels == 2) {
        bandPass(p->if_fs, 18.5e3, 19.5e3, stereo_taps, p->carrier_coeff);
        bandPass(p->if_fs, 22e3, 54e3, stereo_taps, p->stereo_coeff);
        p->state_stereo.resize(stereo_taps - 1, 0.0f);
        p->state_carrier.resize(stereo_taps - 1, 0.0f);
        p->state_stereofilt.resize(audio_taps - 1, 0.0f);
        p->state_allpass.resize(int((stereo_taps - 1) / 2), 0.0f);
        p->state_PLL = std::vector<float>{0.0, 0.0, 1.0, 0.0, 1.0, 0};
    }
    return p;
}

void ref_pipeline_destroy(ref_pipeline *p) { delete p; }

static void audio_stage(ref_pipeline *p, std::vector<float> &y, const std::vector<float> &x, std::vector<float> &st)
{
    y.clear();
    y.reserve(x.size() + 2);
    if (p->mode == 0 || p->mode == 1) convolveBlockFastFIR(y, x, p->audio_coeff, st, p->audio_decim, false);
    else convolveBlockResampleFIR(y, x, p->audio_coeff, st, p->audio_decim, p->audio_upsamp, false);
}

size_t ref_pipeline_process(ref_pipeline *p, const uint8_t *iq, size_t n_bytes)
{
    std::vector<float> iq_data(n_bytes);
    ref_read_block(iq, n_bytes, iq_data.data());
    std::vector<float> I_in, Q_in;
    I_in.reserve(n_bytes / 2 + 1);
    Q_in.reserve(n_bytes / 2 + 1);
    for (size_t k = 0; k + 1 < n_bytes; k += 2) { I_in.push_back(iq_data[k]); Q_in.push_back(iq_data[k + 1]); }
    p->I_filt.clear(); p->I_filt.reserve(I_in.size() / p->rf_decim + 2);
    p->Q_filt.clear(); p->Q_filt.reserve(I_in.size() / p->rf_decim + 2);
    convolveBlockFastFIR(p->I_filt, I_in, p->rf_coeff, p->I_state, p->rf_decim, false);
    convolveBlockFastFIR(p->Q_filt, Q_in, p->rf_coeff, p->Q_state, p->rf_decim, false);
    fmDemod(p->fm_demod, p->I_filt, p->Q_filt, p->prev_i, p->prev_q);
    if (p->channels == 1) {
        audio_stage(p, p->audio_filt, p->fm_demod, p->state_mono);
        p->L = p->audio_filt;
        return p->audio_filt.size();
    }
    allPass(p->fm_demod, p->state_allpass, p->audio_allpass);
    convolveBlockFIR(p->stereo_filt, p->fm_demod, p->stereo_coeff, p->state_stereo);
    convolveBlockFIR(p->carrier_filt, p->fm_demod, p->carrier_coeff, p->state_carrier);
    // allPass output inherits capacity == size; FastFIR reads x[N] once (Q2)
    p->audio_allpass.reserve(p->audio_allpass.size() + 1);
    audio_stage(p, p->audio_filt, p->audio_allpass, p->state_mono);
    fmPLL(p->carrier_filt, p->PLL, p->state_PLL, 19e3, p->if_fs, 2.0, 0.0, 0.01);
    p->mixer.clear();
    p->mixer.reserve(p->stereo_filt.size() + 1);
    p->mixer.resize(p->stereo_filt.size(), 0.0);
    for (unsigned int z = 0; z < p->mixer.size(); z++) p->mixer[z] = p->stereo_filt[z] * p->PLL[z] * 2;
    audio_stage(p, p->stereo_final, p->mixer, p->state_stereofilt);
    p->L.assign(p->stereo_final.size(), 0.0f);
    p->R.assign(p->stereo_final.size(), 0.0f);
    for (unsigned int s = 0; s < p->L.size(); s++) {
        p->L[s] = p->stereo_final[s] + p->audio_filt[s];
        p->R[s] = p->audio_filt[s] - p->stereo_final[s];
    }
    return p->L.size();
}

// which: 0 carrier_filt 1 stereo_filt 2 PLL 3 mixer 4 allpass 5 audio_filt 6 stereo_final
//        7 I_filt 8 Q_filt 9 fm_demod 10 L 11 R
size_t ref_pipeline_get(ref_pipeline *p, int which, const float **ptr)
{
    std::vector<float> *v[] = {&p->carrier_filt, &p->stereo_filt, &p->PLL, &p->mixer, &p->audio_allpass, &p->audio_filt,
                               &p->stereo_final, &p->I_filt, &p->Q_filt, &p->fm_demod, &p->L, &p->R};
    if (which < 0 || which > 11) { *ptr = nullptr; return 0; }
    *ptr = v[which]->data();
    return v[which]->size();
}

}  // extern "C"

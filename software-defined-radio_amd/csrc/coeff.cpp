// coeff.cpp -- filter-coefficient design and the mode table (host code).
//
// These are host functions in the reference as well (they run once per run on
// at most 44 541 taps): impulseResponseLPF src/filter.cpp:103-114, bandPass
// src/filter.cpp:83-99, mode table src/project.cpp:424-427, block size
// src/project.cpp:55-57.  Float32 results are bit-compatible with the
// reference: intermediates in double (its PI is a double literal), one float
// rounding per assignment it makes.
#include <cmath>

#include "fmrx_internal.hpp"

namespace fmrx {

static const double kPi = 3.14159265358979323846;  // include/dy4.h:23

void design_lpf(float Fs, float Fc, int taps, float *h)
{
    const int centre = (taps - 1) / 2;
    const float norm_fc = Fc / (Fs / 2);
    for (int i = 0; i < taps; i++) {
        float v = norm_fc;
        if (i != centre) {
            const double a = kPi * norm_fc * (i - centre);
            v = static_cast<float>(norm_fc * (std::sin(a) / a));
        }
        const double w = std::sin(i * kPi / taps);  // window over taps, not taps-1
        h[i] = static_cast<float>(v * (w * w));
    }
}

void design_bpf(float Fs, float Fb, float Fe, int taps, float *h)
{
    const int centre = (taps - 1) / 2;
    const float norm_centre = ((Fe + Fb) / 2) / (Fs / 2);
    const float norm_pass = (Fe - Fb) / (Fs / 2);
    for (int i = 0; i < taps; i++) {
        float v = norm_pass;
        if (i != centre) {
            const double a = kPi * norm_pass / 2 * (i - centre);
            v = static_cast<float>(norm_pass * (std::sin(a) / a));
        }
        v = static_cast<float>(v * std::cos(i * kPi * norm_centre));
        const double w = std::sin(i * kPi / taps);
        h[i] = static_cast<float>(v * w * w);
    }
}

}  // namespace fmrx

extern "C" {

int fmrx_impulse_response_lpf(float Fs, float Fc, unsigned short num_taps, float *h)
{
    if (!h || num_taps == 0 || !(Fs > 0)) return fmrx::fail(FMRX_EINVAL, "impulse_response_lpf: bad arguments");
    fmrx::design_lpf(Fs, Fc, num_taps, h);
    return FMRX_OK;
}

int fmrx_band_pass(float Fs, float Fb, float Fe, unsigned short num_taps, float *h)
{
    if (!h || num_taps == 0 || !(Fs > 0)) return fmrx::fail(FMRX_EINVAL, "band_pass: bad arguments");
    fmrx::design_bpf(Fs, Fb, Fe, num_taps, h);
    return FMRX_OK;
}

int fmrx_mode_params(int mode, int rf_taps, int base_audio_taps, int stereo_taps, fmrx_params *p)
{
    if (!p) return fmrx::fail(FMRX_EINVAL, "mode_params: null output");
    if (mode < 0 || mode > 3) return fmrx::fail(FMRX_EINVAL, "mode_params: mode %d not in 0..3", mode);
    if (rf_taps < 2 || base_audio_taps < 2 || stereo_taps < 2)
        return fmrx::fail(FMRX_EINVAL, "mode_params: tap counts must be >= 2");
    // rows: rf_Fs, if_Fs, audio_Fs, rf_decim, audio_decim, audio_upsamp
    static const int table[4][6] = {
        {2400000, 240000, 48000, 10, 5, 0},
        {1440000, 288000, 48000, 5, 6, 0},
        {2400000, 240000, 44100, 10, 800, 147},
        {960000, 320000, 44100, 3, 3200, 441},
    };
    const int *r = table[mode];
    std::memset(p, 0, sizeof(*p));
    p->mode = mode;
    p->rf_Fs = r[0];
    p->if_Fs = r[1];
    p->audio_Fs = static_cast<float>(r[2]);
    p->rf_decim = r[3];
    p->audio_decim = r[4];
    p->audio_upsamp = r[5];
    p->rf_taps = rf_taps;
    p->stereo_taps = stereo_taps;
    p->audio_taps = r[5] ? base_audio_taps * r[5] : base_audio_taps;
    if (p->audio_taps > 65535) return fmrx::fail(FMRX_EINVAL, "mode_params: audio_taps %d > 65535", p->audio_taps);
    // 1024 audio samples per block for modes 0/1, 7 resampler periods for 2/3
    p->block_bytes = (r[5] ? 7 : 1024) * r[3] * r[4] * 2;
    return FMRX_OK;
}

}  // extern "C"

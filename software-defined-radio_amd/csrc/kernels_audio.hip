// kernels_audio.hip -- audio-rate decimating FIR on the FM-demodulated stream
// (RF_MONO / RF_STEREO: src/project.cpp:346, 219, 257 -> src/filter.cpp:158-188).
//
// Round-1 state: this stage (~10 % of the mono path's MACs) runs on
// the generic kernel; the register-window kernel of kernels_fe.hip is the
// template for its specialised version.
#include "fmrx_internal.hpp"

namespace fmrx {

int audio_plan_init(AudioPlan &pl, const float *h, int taps, int decim)
{
    pl.taps = taps;
    pl.decim = decim;
    pl.fast = false;
    FMRX_TRY(pl.h.alloc(taps));
    FMRX_HIP(hipMemcpy(pl.h.p, h, taps * sizeof(float), hipMemcpyHostToDevice));
    return FMRX_OK;
}

int audio_fir_launch(const AudioPlan &pl, const float *d_x, size_t n_in, int delay, float *d_y, hipStream_t stream,
                     bool /*force_generic*/)
{
    return k_fir_generic(d_x - delay, n_in / pl.decim, pl.h.p, pl.taps, pl.decim, d_y, stream);
}

}  // namespace fmrx

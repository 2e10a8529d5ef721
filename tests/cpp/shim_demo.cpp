// shim_demo.cpp -- a project.cpp-style caller written against the reference's own
// operator API (include/filter.h signatures), compiled against
// include/fmrx_filter.hpp instead.  It replays RF_FrontEnd + RF_MONO for mode 0
// on one block read from stdin with the reference's own readStdinBlockData (raw u8 I/Q) and
// writes float audio to argv[1].
// Exit codes: 0 ok, 2 = the library reported "no device" (expected on a CPU box).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fmrx_filter.hpp"   // the only line that differs from `#include "filter.h"`

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    try {
        const int rf_Fs = 2400000, if_fs = 240000, rf_decim = 10, audio_decim = 5, rf_taps = 101, audio_taps = 101;
        std::vector<float> rf_coeff, audio_coeff;
        impulseResponseLPF(rf_Fs, 100000, rf_taps, rf_coeff);              // project.cpp:50
        impulseResponseLPF(if_fs, 16000, audio_taps, audio_coeff);        // project.cpp:321
        std::vector<float> iq_data;
        readStdinBlockData(102400, 0, iq_data);                            // project.cpp:82 -> iofunc.cpp:128-135
        if (std::cin.rdstate() != 0) return 1;                             // project.cpp:83
        std::vector<float> I_in, Q_in;
        for (size_t k = 0; k + 1 < iq_data.size(); k += 2) { I_in.push_back(iq_data[k]); Q_in.push_back(iq_data[k + 1]); }
        std::vector<float> I_state(rf_taps - 1, 0.0f), Q_state(rf_taps - 1, 0.0f), state_mono(audio_taps - 1, 0.0f);
        float prev_i = 0.0f, prev_q = 0.0f;
        std::vector<float> I_filt, Q_filt, fm_demod, audio_filt;
        convolveBlockFastFIR(I_filt, I_in, rf_coeff, I_state, rf_decim, false);   // project.cpp:111
        convolveBlockFastFIR(Q_filt, Q_in, rf_coeff, Q_state, rf_decim, false);   // project.cpp:121
        fmDemod(fm_demod, I_filt, Q_filt, prev_i, prev_q);                          // project.cpp:128
        convolveBlockFastFIR(audio_filt, fm_demod, audio_coeff, state_mono, audio_decim, false);   // project.cpp:346
        FILE *o = std::fopen(argv[1], "wb");
        std::fwrite(audio_filt.data(), sizeof(float), audio_filt.size(), o);
        std::fclose(o);
        std::printf("ok %zu\n", audio_filt.size());
        return 0;
    } catch (const fmrx::Error &e) {
        std::printf("fmrx::Error %d: %s\n", e.code, e.what());
        return e.code == FMRX_ENODEV ? 2 : 3;
    }
}

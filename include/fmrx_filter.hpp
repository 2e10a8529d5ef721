// fmrx_filter.hpp -- header-only C++ shim: the reference's own operator API
// (include/filter.h:18-43 and include/iofunc.h:36 of
// mnigm2001/Software-Defined-Radio, std::vector<float>& signatures) on top of
// the C ABI of libfmrx.so.  A project.cpp-style caller swaps
// `#include "filter.h"` for `#include "fmrx_filter.hpp"`, links -lfmrx instead
// of filter.o, and its calls run as HIP kernels on the MI355X.
//
// Same names, argument order and meaning as the reference, including its
// quirks (bandPass(Fs,Fb,Fe,taps,out); allPass(in,state,out); the unused
// trailing printData flags).  Differences, all deliberate:
//   * the reference's unchecked preconditions throw fmrx::Error instead of
//     reading out of bounds;
//   * convolveBlockFastFIR does not perform the reference's stray iteration
//     past the end of y.
// Define FMRX_FILTER_NO_GLOBAL to keep the names inside namespace fmrx only.
#pragma once
#include <cstdint>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "fmrx.h"

namespace fmrx {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc)
{
    if (rc != FMRX_OK) throw Error(rc, fmrx_last_error());
}

// include/filter.h:24
inline void impulseResponseLPF(float Fs, float Fc, unsigned short int num_taps, std::vector<float> &h)
{
    h.clear();
    h.resize(num_taps, 0.0f);
    check(fmrx_impulse_response_lpf(Fs, Fc, num_taps, h.data()));
}

// include/filter.h:20
inline void bandPass(float Fs, float Fb, float Fe, unsigned short int N_taps, std::vector<float> &coeff)
{
    coeff.clear();
    coeff.resize(N_taps, 0.0f);
    check(fmrx_band_pass(Fs, Fb, Fe, N_taps, coeff.data()));
}

// include/filter.h:26
inline void convolveFIR(std::vector<float> &y, const std::vector<float> &x, const std::vector<float> &h)
{
    y.clear();
    y.resize(x.size() + h.size() - 1, 0.0f);
    check(fmrx_convolve_fir(y.data(), x.data(), x.size(), h.data(), h.size()));
}

// include/filter.h:28-29
inline void convolveBlockFIR(std::vector<float> &y, const std::vector<float> &x, const std::vector<float> &h,
                             std::vector<float> &state)
{
    if (state.size() + 1 != h.size()) throw Error(FMRX_EINVAL, "convolveBlockFIR: state must hold h.size()-1 samples");
    y.clear();
    y.resize(x.size(), 0.0f);
    check(fmrx_convolve_block_fir(y.data(), x.data(), x.size(), h.data(), h.size(), state.data()));
}

// include/filter.h:31-32
inline void convolveBlockFastFIR(std::vector<float> &y, const std::vector<float> &x, const std::vector<float> &h,
                                 std::vector<float> &state, const unsigned int audio_decim, const bool /*printData*/ = false)
{
    if (state.size() + 1 != h.size()) throw Error(FMRX_EINVAL, "convolveBlockFastFIR: state must hold h.size()-1 samples");
    if (audio_decim == 0) throw Error(FMRX_EINVAL, "convolveBlockFastFIR: audio_decim must be >= 1");
    y.clear();
    y.resize(x.size() / audio_decim, 0.0f);
    check(fmrx_convolve_block_fast_fir(y.data(), x.data(), x.size(), h.data(), h.size(), state.data(), audio_decim));
}

// include/filter.h:34-35
inline void convolveBlockResampleFIR(std::vector<float> &y, const std::vector<float> &x, const std::vector<float> &h,
                                     std::vector<float> &state, const unsigned int audio_decim,
                                     const unsigned int audio_upsamp, bool /*printData*/ = false)
{
    if (state.size() + 1 != h.size()) throw Error(FMRX_EINVAL, "convolveBlockResampleFIR: state must hold h.size()-1 samples");
    if (audio_decim == 0 || audio_upsamp == 0) throw Error(FMRX_EINVAL, "convolveBlockResampleFIR: zero rate");
    y.clear();
    y.resize((x.size() * audio_upsamp) / audio_decim, 0.0f);
    check(fmrx_convolve_block_resample_fir(y.data(), x.data(), x.size(), h.data(), h.size(), state.data(), audio_decim,
                                           audio_upsamp));
}

// include/filter.h:37
inline void upsample(const std::vector<float> &x, std::vector<float> &xu, const int up_rate)
{
    xu.clear();
    xu.resize(x.size() * up_rate, 0.0f);
    check(fmrx_upsample(x.data(), x.size(), xu.data(), up_rate));
}

// include/filter.h:39
inline void downsample(std::vector<float> &output, const std::vector<float> &input, const unsigned short int ds_coeff)
{
    if (ds_coeff == 0) throw Error(FMRX_EINVAL, "downsample: zero factor");
    output.clear();
    output.resize(input.size() / ds_coeff + 1, 0.0f);
    size_t n = 0;
    check(fmrx_downsample(output.data(), &n, input.data(), input.size(), ds_coeff));
    output.resize(n);
}

// include/filter.h:41
inline void fmDemod(std::vector<float> &fm_demod, const std::vector<float> &I, const std::vector<float> &Q, float &prev_i,
                    float &prev_q)
{
    if (I.size() != Q.size()) throw Error(FMRX_EINVAL, "fmDemod: I and Q differ in length");
    fm_demod.clear();
    fm_demod.resize(I.size(), 0.0f);
    check(fmrx_fm_demod(fm_demod.data(), I.data(), Q.data(), I.size(), &prev_i, &prev_q));
}

// include/filter.h:18 (input, state, output)
inline void allPass(const std::vector<float> &input_block, std::vector<float> &state_block, std::vector<float> &output_block)
{
    output_block.clear();
    output_block.resize(input_block.size(), 0.0f);
    check(fmrx_all_pass(input_block.data(), input_block.size(), state_block.data(), state_block.size(), output_block.data()));
}

// include/filter.h:22
inline void fmPLL(const std::vector<float> &PLLIn, std::vector<float> &ncoOut, std::vector<float> &state, float freq,
                  float Fs, float ncoScale, float phaseAdjust, float normBandwidth)
{
    if (state.size() < 6) throw Error(FMRX_EINVAL, "fmPLL: state needs 6 elements");
    ncoOut.clear();
    ncoOut.resize(PLLIn.size() + 1, 0.0f);
    check(fmrx_fm_pll(PLLIn.data(), PLLIn.size(), ncoOut.data(), state.data(), freq, Fs, ncoScale, phaseAdjust, normBandwidth));
}

// include/filter.h:43 / src/filter.cpp:270 -- a host-side copy helper in the
// reference; kept for source compatibility (no device work).
inline void setVec(const std::vector<float> &vec1, std::vector<float> &vec2, int begin, int end, int mode = 1)
{
    int k = 0;
    for (int i = begin; i < end; i++, k++) {
        if (mode == 1) vec2[k] = vec1[i];
        else if (mode == 2) vec2[i] = vec1[k];
    }
}

// include/fourier.h estimatePSD (src/fourier.cpp:44-128); NFFT = 512 (include/dy4.h:27)
inline void estimatePSD(std::vector<float> &freq, std::vector<float> &psd_est, const std::vector<float> &samples, const float Fs)
{
    const int nfft = 512;
    freq.assign(nfft / 2, 0.0f);
    psd_est.assign(nfft / 2, 0.0f);
    check(fmrx_estimate_psd(freq.data(), psd_est.data(), samples.data(), samples.size(), Fs, nfft));
}

// include/logfunc.h logVector (src/logfunc.cpp:23-43): the same gnuplot text format.  Host-side
// file output, no device work; the reference hard-codes the directory "../data/Graphing/", here
// `filename` is used as given (append ".dat" yourself or pass the reference's relative path).
inline void logVector(const std::string &filename, const std::vector<float> &x, const std::vector<float> &y)
{
    std::ofstream fd(filename);
    if (!fd) throw Error(FMRX_EINVAL, "logVector: cannot open " + filename);
    fd << "#\tx_axis\ty_axis\n";
    for (size_t i = 0; i < x.size(); i++) {
        fd << "\t " << x[i] << "\t";
        if (i < y.size()) fd << y[i];
        fd << "\n";
    }
}

// include/iofunc.h:36 -- the conversion of readStdinBlockData on bytes already
// read (the stdin read itself stays with the caller).
inline void convertBlockData(const std::vector<uint8_t> &raw, std::vector<float> &block_data)
{
    block_data.resize(raw.size());
    check(fmrx_u8_to_f32(raw.data(), raw.size(), block_data.data()));
}

// include/iofunc.h:36, src/iofunc.cpp:128-135 under its own name and signature: reads num_samples bytes
// from std::cin and converts them on the device.  Like the reference it ignores block_id, leaves the
// stream state for the caller to test (src/project.cpp:83), and on a short read converts the bytes it
// got with the rest of the block as zero bytes (the reference's zero-initialised raw_data, Q6).
// Define FMRX_FILTER_NO_IOFUNC to keep the reference's own iofunc.cpp in the link instead.
#ifndef FMRX_FILTER_NO_IOFUNC
inline void readStdinBlockData(unsigned int num_samples, unsigned int block_id, std::vector<float> &block_data)
{
    (void)block_id;
    std::vector<uint8_t> raw(num_samples, 0);
    std::cin.read(reinterpret_cast<char *>(raw.data()), num_samples * sizeof(char));
    convertBlockData(raw, block_data);
}
#endif

}  // namespace fmrx

#ifndef FMRX_FILTER_NO_GLOBAL
using fmrx::allPass;
using fmrx::bandPass;
using fmrx::convolveBlockFastFIR;
using fmrx::convolveBlockFIR;
using fmrx::convolveBlockResampleFIR;
using fmrx::convolveFIR;
using fmrx::downsample;
using fmrx::estimatePSD;
using fmrx::fmDemod;
using fmrx::fmPLL;
using fmrx::impulseResponseLPF;
#ifndef FMRX_FILTER_NO_IOFUNC
using fmrx::readStdinBlockData;
#endif
using fmrx::setVec;
using fmrx::upsample;
#endif

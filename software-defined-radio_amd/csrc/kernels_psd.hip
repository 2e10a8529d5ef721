// kernels_psd.hip -- Bartlett power-spectral-density estimate (diagnostics).
//
// Replaces estimatePSD (include/fourier.h, src/fourier.cpp:44-128) with its
// O(N^2) DFT (src/fourier.cpp:15-23) -- the tool the reference's authors used to
// validate every stage (report p.3-6) -- so GPU outputs can be inspected the same
// way without leaving the device.  SURVEY 8(f) rank 3.
//
// One thread per (segment, frequency bin): Hann-windowed nfft-point DFT bin,
// accumulated in the reference's order (k ascending, float32, separate multiply
// and add); |X|^2 scaling, x2 for the negative frequencies, 10 log10; then the
// segments are averaged in dB, in segment order.  The twiddle angle is the
// reference's float32 value fl(-2*PI*(k*m)/N) (up to 3.2e3 rad: its rounding IS
// part of the reference's result); sine and cosine of it come from the device
// math library (this is a diagnostic, accuracy over speed).
#include "fmrx_internal.hpp"

#pragma clang fp contract(off)

namespace fmrx {

namespace {

__global__ void psd_segments_kernel(const float *__restrict__ x, int nfft, int nseg, float Fs, float *__restrict__ seg_db)
{
    const int half = nfft / 2;
    const long gid = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long>(nseg) * half) return;
    const int sg = static_cast<int>(gid / half), m = static_cast<int>(gid % half);
    const float *xs = x + static_cast<long>(sg) * nfft;
    const double PI = 3.14159265358979323846;
    float re = 0.0f, im = 0.0f;
    for (int k = 0; k < nfft; k++) {
        const double sn = sin(k * PI / nfft);
        const float hann = static_cast<float>(sn * sn);                 // std::pow(std::sin(i*PI/N), 2.0)
        const float w = xs[k] * hann;
        const float ang = static_cast<float>(-2 * PI * static_cast<unsigned>(k * m) / static_cast<unsigned>(nfft));
        float s, c;
        sincosf(ang, &s, &c);   // library accuracy: the spectrum's deep nulls amplify twiddle errors
        const float pr = w * c, pi = w * s;
        re = re + pr;
        im = im + pi;
    }
    const float mag = hypotf(re, im);
    float p = static_cast<float>((1 / (Fs * nfft / 2)) * (static_cast<double>(mag) * static_cast<double>(mag)));
    p = 2 * p;
    seg_db[gid] = 10 * log10f(p);
}

__global__ void psd_average_kernel(const float *__restrict__ seg_db, int half, int nseg, float Fs, int nfft,
                                   float *__restrict__ freq, float *__restrict__ psd)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= half) return;
    float acc = 0.0f;
    for (int l = 0; l < nseg; l++) acc = acc + seg_db[static_cast<long>(l) * half + m];
    psd[m] = acc / nseg;
    const float df = Fs / nfft;
    const float N = (Fs / 2 - 0.0f) / df;              // LinearSpacedArray(freq, Fs/2, 0.0, df)
    freq[m] = (m < N) ? 0.0f + m * df : 0.0f;
}

}  // namespace

int k_estimate_psd(const float *d_x, size_t n, float Fs, int nfft, float *d_seg_db, float *d_freq, float *d_psd, hipStream_t s)
{
    const int nseg = static_cast<int>(n / static_cast<size_t>(nfft));
    const int half = nfft / 2;
    const long total = static_cast<long>(nseg) * half;
    hipLaunchKernelGGL(psd_segments_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, d_x, nfft, nseg,
                       Fs, d_seg_db);
    hipLaunchKernelGGL(psd_average_kernel, dim3((half + 255) / 256), dim3(256), 0, s, d_seg_db, half, nseg, Fs, nfft, d_freq,
                       d_psd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch psd kernels: %s", hipGetErrorString(e));
    return FMRX_OK;
}

}  // namespace fmrx

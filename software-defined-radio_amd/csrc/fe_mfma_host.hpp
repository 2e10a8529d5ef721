// fe_mfma_host.hpp -- host-side (plain C++, no HIP) construction of the matrix-core operands used by
// kernels_fe_mfma.hip: the fixed-point digit image of the front-end taps and the Toeplitz image of the
// audio taps.  Header-only so that tests/cpp/fe_mfma_host_test.cpp can check it with g++ on a box
// without a GPU (tests/test_host_cpu.py): digits in range and reconstructing the quantised tap, and an
// emulation of the MFMA's dot products reproducing the FIR.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace fmrx {

// shape of the front-end MFMA tile for (taps T, decimation D); mirrors MfCfg in kernels_fe_mfma.hip
struct FeMfmaShape {
    int col_out = 8;        // IF outputs per column (x {I,Q} = 16 rows)
    int front = 0;          // bytes of a column's window in front of its first output's sample (multiple of 16)
    int win = 0;            // bytes a column's rows touch
    int ksteps = 0;         // K-steps of 64 bytes
};
inline FeMfmaShape fe_mfma_shape(int T, int D)
{
    FeMfmaShape s;
    s.front = (2 * (T - 1) + 15) / 16 * 16;
    s.win = s.front + 2 * D * (s.col_out - 1) + 2;
    s.ksteps = (s.win + 63) / 64;
    return s;
}

// Largest s with max|round(h*2^s)| <= 127*256^(ndig-1): every balanced base-256 digit then fits int8.
// Returns false for tap sets the fixed-point form does not handle (non-finite, all zero, absurd range).
inline bool fe_mfma_scale(const float *h, int taps, int ndig, int *s_out)
{
    double maxabs = 0.0;
    for (int k = 0; k < taps; k++) {
        if (!std::isfinite(h[k])) return false;
        maxabs = std::fmax(maxabs, std::fabs(static_cast<double>(h[k])));
    }
    if (maxabs == 0.0 || maxabs < 1e-30 || maxabs > 1e30) return false;
    const double limit = 127.0 * std::pow(256.0, ndig - 1);
    int s = static_cast<int>(std::floor(std::log2(limit / maxabs)));
    while (std::ldexp(maxabs, s) > limit) s--;
    *s_out = s;
    return true;
}

// q = round(h * 2^s) as ndig balanced base-256 digits, least significant first: q = sum dig[d] * 256^d
inline void fe_mfma_digits(float h, int s, int ndig, int8_t *dig)
{
    long q = std::llround(std::ldexp(static_cast<double>(h), s));
    for (int d = 0; d < ndig; d++) {
        const long v = ((q + 128) & 255) - 128;           // in [-128, 127]
        dig[d] = static_cast<int8_t>(v);
        q = (q - v) / 256;
    }
}

// Digit image of the taps in A-operand order of v_mfma_i32_16x16x64_i8: [kstep][digit][lane][16 bytes].
// Lane (row m = lane&15, quarter g = lane>>4) holds the coefficients row m applies to window bytes
// 64*kstep + 16*g + 0..15.  Row m -> output r = 2*(m/4) + (m%4)/2 of the column, channel c = m%2 (0 = I):
// the C layout (row = 4*(lane>>4) + reg) then hands lane (col, g) the outputs 2g, 2g+1 as (I,Q,I,Q).
// Output r, channel c, tap k meets window byte front + 2*D*r + c - 2*k.
inline void fe_mfma_build_image(const float *h, int T, int D, int s, int ndig, std::vector<int8_t> &img)
{
    const FeMfmaShape sh = fe_mfma_shape(T, D);
    img.assign(static_cast<size_t>(sh.ksteps) * ndig * 64 * 16, 0);
    int8_t dig[8];
    for (int j = 0; j < sh.ksteps; j++)
        for (int lane = 0; lane < 64; lane++)
            for (int b = 0; b < 16; b++) {
                const int m = lane & 15, g = lane >> 4;
                const int r = 2 * (m / 4) + (m % 4) / 2, c = m % 2;
                const int p = 64 * j + 16 * g + b;                      // window byte
                const int e = sh.front + 2 * D * r + c - p;             // = 2k for the tap that meets it
                if (e < 0 || (e & 1) || e / 2 > T - 1) continue;
                fe_mfma_digits(h[e / 2], s, ndig, dig);
                for (int d = 0; d < ndig; d++)
                    img[((static_cast<size_t>(j) * ndig + d) * 64 + lane) * 16 + b] = dig[d];
            }
}

// Toeplitz image of the audio taps in A-operand order of v_mfma_f32_16x16x4_f32: [kstep][lane], lane
// (row i = lane&15, k = lane>>4) holds the tap that output i of a column applies to window sample
// w = 16*(kstep/4) + 4*k + kstep%4 (so that a lane's B operands of 4 consecutive K-steps are 4
// consecutive samples), i.e. h[decim*i + taps-1 - w], or 0 outside the filter.
inline int audio_mfma_ksteps(int taps, int decim) { return ((taps - 1) + 15 * decim + 1 + 15) / 16 * 4; }
inline void audio_mfma_build_table(const float *h, int taps, int decim, std::vector<float> &tab)
{
    const int ak = audio_mfma_ksteps(taps, decim);
    tab.assign(static_cast<size_t>(ak) * 64, 0.0f);
    for (int j = 0; j < ak; j++)
        for (int lane = 0; lane < 64; lane++) {
            const int w = 16 * (j / 4) + 4 * (lane >> 4) + j % 4;
            const int k = decim * (lane & 15) + taps - 1 - w;
            if (k >= 0 && k < taps) tab[static_cast<size_t>(j) * 64 + lane] = h[k];
        }
}

}  // namespace fmrx

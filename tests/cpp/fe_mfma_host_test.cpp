// Host-only check of software-defined-radio_amd/csrc/fe_mfma_host.hpp (no GPU, no HIP): the operands
// that kernels_fe_mfma.hip feeds the matrix cores.  Emulates what v_mfma_i32_16x16x64_i8 and
// v_mfma_f32_16x16x4_f32 compute from them (a dot product over K per (row, column)) and compares
// with the FIR they are meant to be.
//   usage: fe_mfma_host_test <taps.f32> <T> <D> <audio_taps.f32> <TA> <DA>
// exit 0 and prints "ok ..." on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "fe_mfma_host.hpp"

static std::vector<float> load(const char *path, int n)
{
    std::vector<float> v(n);
    FILE *f = std::fopen(path, "rb");
    if (!f || std::fread(v.data(), sizeof(float), n, f) != static_cast<size_t>(n)) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
    std::fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc != 7) return 2;
    const int T = std::atoi(argv[2]), D = std::atoi(argv[3]), TA = std::atoi(argv[5]), DA = std::atoi(argv[6]);
    const std::vector<float> h = load(argv[1], T), ha = load(argv[4], TA);
    const int ND = 3;
    int fails = 0;

    // ---- front end: digits ----
    int s = 0;
    if (!fmrx::fe_mfma_scale(h.data(), T, ND, &s)) { std::puts("scale failed"); return 1; }
    double worst_q = 0;
    for (int k = 0; k < T; k++) {
        int8_t dig[3];
        fmrx::fe_mfma_digits(h[k], s, ND, dig);
        const long q = dig[0] + 256L * dig[1] + 65536L * dig[2];
        if (q != std::llround(std::ldexp(static_cast<double>(h[k]), s))) fails++;                 // digits reconstruct q exactly
        worst_q = std::fmax(worst_q, std::fabs(std::ldexp(static_cast<double>(q), -s) - h[k]));  // |tap error| <= 2^-(s+1)
    }
    if (worst_q > std::ldexp(0.5, -s) * 1.0000001) fails++;

    // ---- front end: emulate the MFMAs of one tile column on random bytes ----
    const fmrx::FeMfmaShape sh = fmrx::fe_mfma_shape(T, D);
    std::vector<int8_t> img;
    fmrx::fe_mfma_build_image(h.data(), T, D, s, ND, img);
    if (sh.front % 16 || sh.win > 64 * sh.ksteps || 2 * (T - 1) > sh.front) fails++;
    std::mt19937 rng(1234);
    std::vector<uint8_t> win(64 * sh.ksteps);
    double worst_fe = 0;
    for (int trial = 0; trial < 20; trial++) {
        for (auto &b : win) b = static_cast<uint8_t>(rng() & 255);
        for (int m = 0; m < 16; m++) {                       // 16 rows of the tile
            long acc[3] = {0, 0, 0};
            for (int j = 0; j < sh.ksteps; j++)
                for (int g = 0; g < 4; g++)                  // the 4 lanes (m, g) hold K bytes 16g..16g+15 of the step
                    for (int b = 0; b < 16; b++) {
                        const int lane = m + 16 * g, p = 64 * j + 16 * g + b;
                        const int x = static_cast<int8_t>(win[p] ^ 0x80);            // what the kernel feeds: u8 ^ 0x80 as int8
                        for (int d = 0; d < ND; d++) acc[d] += static_cast<long>(img[((static_cast<size_t>(j) * ND + d) * 64 + lane) * 16 + b]) * x;
                    }
            for (int d = 0; d < ND; d++)
                if (std::labs(acc[d]) >= (1L << 31)) fails++;                           // int32 accumulators do not overflow
            if (std::labs(acc[1] * 256 + acc[0]) >= (1L << 31)) fails++;
            // the kernel's epilogue: (acc2*65536 + (acc1*256 + acc0)) * 2^-(s+7), two float roundings
            const float scale_lo = static_cast<float>(std::ldexp(1.0, -s - 7)), scale_hi = scale_lo * 65536.0f;
            const float flo = static_cast<float>(static_cast<int>(acc[1] * 256 + acc[0])) * scale_lo;
            const float got = std::fmaf(static_cast<float>(static_cast<int>(acc[2])), scale_hi, flo);
            // what row m is meant to be: output r of the column, channel c, FIR over the window in double
            const int r = 2 * (m / 4) + (m % 4) / 2, c = m % 2;
            double want = 0;
            for (int k = 0; k < T; k++) want += static_cast<double>(h[k]) * ((static_cast<int>(win[sh.front + 2 * D * r + c - 2 * k]) - 128) / 128.0);
            worst_fe = std::fmax(worst_fe, std::fabs(got - want));
        }
    }
    if (worst_fe > 6e-7) fails++;        // float32 ulp at |y| <= 1.5 is 1.2e-7; 24-bit taps add <= T * 2^-(s+1)

    // ---- audio: emulate the f32 MFMAs of one column on random samples ----
    std::vector<float> tab;
    fmrx::audio_mfma_build_table(ha.data(), TA, DA, tab);
    const int ak = fmrx::audio_mfma_ksteps(TA, DA);
    std::vector<float> x(4 * ak);
    std::uniform_real_distribution<float> ud(-1.0f, 1.0f);
    double worst_au = 0;
    for (int trial = 0; trial < 20; trial++) {
        for (auto &v : x) v = ud(rng);
        for (int i = 0; i < 16; i++) {
            double got = 0;
            for (int j = 0; j < ak; j++)
                for (int kq = 0; kq < 4; kq++) got += static_cast<double>(tab[static_cast<size_t>(j) * 64 + i + 16 * kq]) * x[16 * (j / 4) + 4 * kq + j % 4];
            double want = 0;             // y[i] = sum_k ha[k] * x[DA*i + TA-1 - k], window sample 0 = x[-(TA-1)] of output 0
            for (int k = 0; k < TA; k++) want += static_cast<double>(ha[k]) * x[DA * i + TA - 1 - k];
            worst_au = std::fmax(worst_au, std::fabs(got - want));
        }
    }
    if (worst_au > 1e-12) fails++;       // same products, double accumulation on both sides
    if ((TA - 1) + 15 * DA + 1 > 4 * ak) fails++;

    std::printf("%s T=%d D=%d s=%d ksteps=%d front=%d: tap quantisation %.2e, FE tile vs double FIR %.2e; audio TA=%d DA=%d K-steps=%d: %.2e\n",
                fails ? "FAIL" : "ok", T, D, s, sh.ksteps, sh.front, worst_q, worst_fe, TA, DA, ak, worst_au);
    return fails ? 1 : 0;
}

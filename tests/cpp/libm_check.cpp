// libm_check.cpp -- pins csrc/glibc_libm.hpp (the restatement of glibc 2.35's sinf / cosf / atan2f
// that the device PLL uses) against the C library of the machine it runs on.  Test infrastructure.
//
//   libm_check sincos <first> <count> [threads]   every float bit pattern first .. first+count-1
//   libm_check atan2 <pairs> <seed> [threads]     random pairs over all exponents, PLL-shaped pairs
//                                                 (v*-sin t, v*cos t), and the special values
// Prints "<name> checked <n> mismatches <m>" per function and the first few mismatches; exit 1 if any.
#include <atomic>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "glibc_libm.hpp"

using namespace fmrx::glibc235;

static bool same(float a, float b)
{
    if (a != a && b != b) return true;   // any NaN == any NaN
    return f2u(a) == f2u(b);
}

static std::atomic<uint64_t> bad_s{0}, bad_c{0}, bad_a{0}, shown{0};
static std::atomic<uint64_t> flat_sc{0}, flat_a{0};   // arguments that also went through the branch-free variants

static void report(const char *what, float x, float y, float got, float want)
{
    if (shown.fetch_add(1) < 12)
        fprintf(stderr, "%s(%a [%08x], %a [%08x]) = %a, libm %a\n", what, x, f2u(x), y, f2u(y), got, want);
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const unsigned nt = argc > 4 ? std::atoi(argv[4]) : std::thread::hardware_concurrency();
    std::vector<std::thread> th;
    if (!std::strcmp(argv[1], "sincos")) {
        const uint64_t first = std::strtoull(argv[2], nullptr, 0), count = std::strtoull(argv[3], nullptr, 0);
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([=] {
                uint32_t w24[24];
                inv_pio4_table(w24);
                uint64_t nflat = 0;
                for (uint64_t i = first + t; i < first + count; i += nt) {
                    const float x = u2f(static_cast<uint32_t>(i));
                    float s, c;
                    sincosf_glibc(x, &s, &c);
                    const float ws = sinf(x), wc = cosf(x);
                    if (!same(s, ws)) { bad_s++; report("sinf", x, 0, s, ws); }
                    if (!same(c, wc)) { bad_c++; report("cosf", x, 0, c, wc); }
                    if (sincosf_large_ok(x)) {               // the branch-free variant, wherever it is defined
                        sincosf_large_flat(x, w24, &s, &c);
                        if (!same(s, ws)) { bad_s++; report("sinf (flat)", x, 0, s, ws); }
                        if (!same(c, wc)) { bad_c++; report("cosf (flat)", x, 0, c, wc); }
                        nflat++;
                    }
                    if (sincosf_mid_ok(x)) {                 // and the form with the table look-up folded into selects
                        sincosf_mid_flat(x, &s, &c);
                        if (!same(s, ws)) { bad_s++; report("sinf (mid)", x, 0, s, ws); }
                        if (!same(c, wc)) { bad_c++; report("cosf (mid)", x, 0, c, wc); }
                        nflat++;
                    }
                }
                flat_sc += nflat;
            });
        for (auto &t : th) t.join();
        printf("sinf checked %" PRIu64 " mismatches %" PRIu64 "\ncosf checked %" PRIu64 " mismatches %" PRIu64 "\n", count,
               bad_s.load(), count, bad_c.load());
        printf("branch-free sincosf checked %" PRIu64 "\n", flat_sc.load());
        return bad_s || bad_c ? 1 : 0;
    }
    if (!std::strcmp(argv[1], "atan2")) {
        const uint64_t pairs = std::strtoull(argv[2], nullptr, 0), seed = std::strtoull(argv[3], nullptr, 0);
        std::atomic<uint64_t> n{0};
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([=, &n] {
                std::mt19937_64 g(seed + 7919 * t);
                uint64_t done = 0;
                uint64_t nflat = 0;
                auto chk = [&](float y, float x) {
                    const float got = atan2f_glibc(y, x), want = atan2f(y, x);
                    if (!same(got, want)) { bad_a++; report("atan2f", y, x, got, want); }
                    if (atan2f_flat_ok(y, x)) {              // the branch-free variant, wherever it is defined
                        const float gf = atan2f_flat(y, x);
                        if (!same(gf, want)) { bad_a++; report("atan2f (flat)", y, x, gf, want); }
                        nflat++;
                    }
                    done++;
                };
                for (uint64_t i = t; i < pairs; i += nt) {
                    const uint64_t r = g();
                    switch (i % 4) {
                    case 0:   // any two bit patterns
                        chk(u2f(static_cast<uint32_t>(r)), u2f(static_cast<uint32_t>(r >> 32)));
                        break;
                    case 1: { // the PLL's phase detector: atan2f(v * -sin t, v * cos t)
                        const float tt = static_cast<float>((r & 0xffffff) * 0.03);
                        const float v = u2f(0x30000000u + static_cast<uint32_t>((r >> 24) % 0x10000000u)) * ((r >> 60) & 1 ? -1.0f : 1.0f);
                        chk(v * (-1 * sinf(tt)), v * cosf(tt));
                        break;
                    }
                    case 2: { // ratios near the range boundaries of atanf's argument reduction
                        const float b[6] = {0.4375f, 0.6875f, 1.1875f, 2.4375f, 1.0f, 33554432.0f};
                        const float q = u2f(f2u(b[r % 6]) + static_cast<int>((r >> 8) % 65) - 32);
                        const float x = u2f(0x3f000000u + static_cast<uint32_t>((r >> 16) & 0xffffff));
                        chk(q * x, ((r >> 50) & 1) ? -x : x);
                        break;
                    }
                    default: { // same exponent neighbourhood
                        const float x = u2f(static_cast<uint32_t>(r));
                        const float y = u2f((f2u(x) & 0x7f800000u) + static_cast<uint32_t>((r >> 32) & 0x03ffffff) - 0x01000000u);
                        chk(((r >> 63) & 1) ? -y : y, x);
                    }
                    }
                }
                if (t == 0) {
                    const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 3.4e38f, -3.4e38f,
                                        1e-30f, -1e-30f, 1e30f, 0.5f, 2.0f};
                    for (float y : sp)
                        for (float x : sp) chk(y, x);
                }
                n += done;
                flat_a += nflat;
            });
        for (auto &t : th) t.join();
        printf("atan2f checked %" PRIu64 " mismatches %" PRIu64 "\n", n.load(), bad_a.load());
        printf("branch-free atan2f checked %" PRIu64 "\n", flat_a.load());
        return bad_a ? 1 : 0;
    }
    return 2;
}

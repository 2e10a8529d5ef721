// glibc_libm.hpp -- sinf / cosf / atan2f with the results of glibc 2.35 (x86-64, FMA-capable
// host), restated for host and device so that the pilot PLL (kernels_pll.hip) can reproduce
// the reference's recurrence bit for bit.
//
// Why this exists.  fmPLL (src/filter.cpp:52-72 of the reference) feeds std::cos / std::sin /
// std::atan2 of float arguments back into a float32 recurrence.  The reference's arithmetic
// therefore includes its C library: the third-party dependency is GNU libm, glibc 2.35
// (Ubuntu 2.35-0ubuntu3.x in this image, here and on the GPU box), which is not part of
// /root/reference.  Its algorithms for these three functions are published:
//   * sinf / cosf: "sincosf" of ARM optimized-routines (Szabolcs Nagy, 2018), in glibc since
//     2.28 as sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h; x86-64 builds select the
//     variant compiled with -mfma on CPUs that have FMA (ifunc), whose double-precision
//     polynomial steps are fused: the fma() calls below mark exactly those steps.
//     Argument reduction: |x| < 120 multiply by 2/pi * 2^24 and round (reduce_fast);
//     otherwise a 192-bit 4/pi table and integer arithmetic (reduce_large), exact.
//   * atan2f / atanf: the float versions of Sun's fdlibm (e_atan2f.c, s_atanf.c), plain
//     float32 operations in source order, no FMA (x86-64 baseline build).
// This file restates those algorithms (nothing is copied from glibc's text; the constants
// are the published ones).  It is PINNED by test, not by reading: tests/cpp/libm_check.cpp
// compares every one of the 2^32 float arguments of sinf/cosf and >10^9 argument pairs of
// atan2f against the C library of this image on the CPU (tests/test_libm_exact.py), and
// tests/test_gpu_parity.py::test_device_libm_is_glibc runs the same comparison for the
// device build on the GPU box.
//
// Host + device: every function is FMRX_HD.  Compile with FP contraction OFF (the pragma
// below does it for clang/hipcc; g++ needs -ffp-contract=off).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FMRX_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define FMRX_HD inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace fmrx {
namespace glibc235 {

FMRX_HD uint32_t f2u(float f)
{
#if defined(__HIPCC__)
    return __builtin_bit_cast(uint32_t, f);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
#endif
}
FMRX_HD float u2f(uint32_t u)
{
#if defined(__HIPCC__)
    return __builtin_bit_cast(float, u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}
FMRX_HD double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- sinf / cosf -----------------------------------------------------------------------------
// polynomial coefficients of sin and cos on [-pi/4, pi/4] (s_sincosf_data.c: __sincosf_table[0];
// entry [1] holds the negated cosine coefficients, i.e. the negated result: rounding is symmetric)
constexpr double kS1 = -0x1.555545995a603p-3, kS2 = 0x1.1107605230bc4p-7, kS3 = -0x1.994eb3774cf24p-13;
constexpr double kC0 = 0x1p0, kC1 = -0x1.ffffffd0c621cp-2, kC2 = 0x1.55553e1068f19p-5, kC3 = -0x1.6c087e89a359dp-10,
                 kC4 = 0x1.99343027bf8c3p-16;
constexpr double kHpiInv = 0x1.45F306DC9C883p+23;   // 2/pi * 2^24
constexpr double kHpi = 0x1.921FB54442D18p0;        // pi/2
constexpr double kPi63 = 0x1.921FB54442D18p-62;     // pi / 2^63

// sin(x) for the reduced argument; xs = x * (+-1), x2 = x*x
FMRX_HD float sin_poly(double xs, double x2)
{
    const double x3 = xs * x2;
    const double s1 = fma64(kS3, x2, kS2);
    const double x7 = x3 * x2;
    const double s = fma64(x3, kS1, xs);
    return static_cast<float>(fma64(s1, x7, s));
}
// cos(x) for the reduced argument
FMRX_HD float cos_poly(double x2)
{
    const double x4 = x2 * x2;
    const double c2 = fma64(kC4, x2, kC3);
    const double c1 = fma64(kC1, x2, kC0);
    const double x6 = x4 * x2;
    const double c = fma64(x4, kC2, c1);
    return static_cast<float>(fma64(c2, x6, c));
}

// 4/pi in 32-bit windows, 8 bits apart (s_sincosf_data.c: __inv_pio4[24]): window i is bytes i .. i+3 of
// 00 00 00 a2 f9 83 6e 4e 44 15 29 fc 27 57 d1 f5 34 dd c0 db 62 95 99 3c 43 90 41 (2/pi, 192 bits, behind
// three zero bytes), big-endian.  Kept as 7 words so that a window is two words and a funnel shift.
FMRX_HD uint32_t inv_pio4(int i)
{
    const uint32_t w[8] = {0x000000a2u, 0xf9836e4eu, 0x441529fcu, 0x2757d1f5u, 0x34ddc0dbu, 0x6295993cu, 0x43904100u, 0u};
    const int q = i >> 2, s = (i & 3) * 8;
    const uint64_t pair = (static_cast<uint64_t>(w[q]) << 32) | w[q + 1];
    return static_cast<uint32_t>(pair >> (32 - s));
}

struct Reduced {
    double x;    // reduced argument in [-pi/4, pi/4]
    int n;       // quadrant (low two bits matter)
};

// |x| in [120, inf): multiply the 24-bit mantissa by 96 bits of 4/pi, exactly (integers)
FMRX_HD Reduced reduce_large(uint32_t xi)
{
    const int idx = static_cast<int>((xi >> 26) & 15);
    const int shift = static_cast<int>((xi >> 23) & 7);
    uint32_t m = (xi & 0xffffffu) | 0x800000u;
    m <<= shift;
    uint64_t res0 = static_cast<uint32_t>(m * inv_pio4(idx));          // 32-bit product, as in the original
    const uint64_t res1 = static_cast<uint64_t>(m) * inv_pio4(idx + 4);
    const uint64_t res2 = static_cast<uint64_t>(m) * inv_pio4(idx + 8);
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    Reduced r;
    r.x = static_cast<double>(static_cast<int64_t>(res0)) * kPi63;
    r.n = static_cast<int>(n);
    return r;
}

// |x| in [0.75, 120): n = round(x * 2/pi) through a scaled float->int conversion, x - n*pi/2 fused
FMRX_HD Reduced reduce_fast(double x)
{
    const double r = x * kHpiInv;
    const int n = (static_cast<int32_t>(r) + 0x800000) >> 24;
    Reduced o;
    o.n = n;
    o.x = fma64(-static_cast<double>(n), kHpi, x);
    return o;
}

FMRX_HD double quadrant_sign(int m) { return ((m + 1) & 2) ? -1.0 : 1.0; }   // {1, -1, -1, 1}[m & 3]

// both values from one argument reduction; `want` bit 0: sine, bit 1: cosine
FMRX_HD void sincosf_glibc(float y, float *sn, float *cs)
{
    const uint32_t xi = f2u(y);
    const uint32_t top = (xi >> 20) & 0x7ff;
    const double xd = static_cast<double>(y);
    if (top < 0x3f4) {                 // |y| < 0.75 (the original compares the top 12 bits with those of pi/4)
        const double x2 = xd * xd;
        if (top < 0x398) {             // |y| < 2^-12
            *sn = y;
            *cs = 1.0f;
            return;
        }
        *sn = sin_poly(xd, x2);
        *cs = cos_poly(x2);
        return;
    }
    Reduced r;
    int m;
    if (top < 0x42f) {                 // |y| < 120
        r = reduce_fast(xd);
        m = r.n;
    } else if (top < 0x7f8) {
        r = reduce_large(xi);
        m = r.n + static_cast<int>(xi >> 31);
    } else {                           // inf / NaN -> NaN
        *sn = *cs = y - y;
        return;
    }
    const double x2 = r.x * r.x;
    const float sp = sin_poly(r.x * quadrant_sign(m), x2);
    float cp = cos_poly(x2);
    if (m & 2) cp = -cp;
    // sine: quadrant n even -> sine polynomial, odd -> cosine polynomial; cosine: the other way round.
    // The sign of the cosine branch depends on (m & 2) for the sine and on the same for the cosine:
    //   sinf: sinf_poly(x*s, x2, table[(m>>1)&1], n)      cosf: sinf_poly(x*s, x2, table[(m>>1)&1], n ^ 1)
    if (r.n & 1) {
        *sn = cp;
        *cs = sp;
    } else {
        *sn = sp;
        *cs = cp;
    }
}

FMRX_HD float sinf_glibc(float y)
{
    float s, c;
    sincosf_glibc(y, &s, &c);
    return s;
}
FMRX_HD float cosf_glibc(float y)
{
    float s, c;
    sincosf_glibc(y, &s, &c);
    return c;
}

// ---- atanf / atan2f (fdlibm, float) -------------------------------------------------------------
FMRX_HD float atanf_glibc(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const uint32_t hx = f2u(x);
    const uint32_t ix = hx & 0x7fffffffu;
    int id;
    if (ix >= 0x4c000000u) {           // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;
        return (hx >> 31) ? -atanhi[3] - atanlo[3] : atanhi[3] + atanlo[3];
    }
    if (ix < 0x3ee00000u) {            // |x| < 0.4375
        if (ix < 0x31000000u) return x;   // |x| < 2^-29
        id = -1;
    } else {
        x = u2f(ix);                   // fabsf
        if (ix < 0x3f980000u) {        // |x| < 1.1875
            if (ix < 0x3f300000u) {    // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {                   // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000u) {    // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {                   // 2.4375 <= |x| < 2^25
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx >> 31) ? -r : r;
}

FMRX_HD float atan2f_glibc(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const uint32_t hx = f2u(x), hy = f2u(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;    // NaN
    if (hx == 0x3f800000u) return atanf_glibc(y);              // x == 1
    const int m = static_cast<int>((hy >> 31) & 1u) | static_cast<int>((hx >> 30) & 2u);   // 2*sign(x) + sign(y)
    if (iy == 0) {                                             // y == +-0
        switch (m) {
        case 0:
        case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000u) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (static_cast<int>(iy) - static_cast<int>(ix)) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                     // |y/x| > 2^60
    else if ((hx >> 31) && k < -60) z = 0.0f;                  // |y|/x < -2^60
    else z = atanf_glibc(u2f(f2u(y / x) & 0x7fffffffu));       // fabsf(y/x): one IEEE divide
    switch (m) {
    case 0: return z;
    case 1: return u2f(f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// ---- the same functions without control flow, for the common arguments ---------------------------------------
// A wave that walks 64 receivers' PLLs (kernels_pll.hip: pll_channels_kernel) has every lane in another branch of the
// functions above -- five argument ranges in atanf, four quadrants in atan2f, sine / cosine polynomial by quadrant --
// and pays for all of them.  These variants compute the same value by the same float operations, with every choice
// a select.  They are defined for the ordinary arguments only (*_ok); the caller sends the whole wave through
// the general functions when any lane holds anything else.  Pinned like the functions above: libm_check.cpp
// compares them with the C library over the same argument sets.
FMRX_HD bool atan2f_flat_ok(float y, float x)
{
    const uint32_t hx = f2u(x), ix = hx & 0x7fffffffu, iy = f2u(y) & 0x7fffffffu;
    // finite and non-zero, both; x == 1 takes atanf(y) in the original, a different sequence of operations
    return ix - 1u < 0x7f7fffffu && iy - 1u < 0x7f7fffffu && hx != 0x3f800000u;
}

FMRX_HD float atan2f_flat(float y, float x)
{
    const float pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const uint32_t hx = f2u(x), hy = f2u(y);
    const int k = (static_cast<int>(hy & 0x7fffffffu) - static_cast<int>(hx & 0x7fffffffu)) >> 23;
    const float q = u2f(f2u(y / x) & 0x7fffffffu);             // fabsf(y/x): one IEEE divide
    // atanf(q), q >= 0 (possibly 0 or +inf after the division)
    const uint32_t iq = f2u(q);
    const bool big = iq >= 0x4c000000u, small = iq < 0x3ee00000u, tiny = iq < 0x31000000u;
    const bool r0 = iq < 0x3f300000u, r1 = iq < 0x3f980000u, r2 = iq < 0x401c0000u;
    const float n0 = 2.0f * q - 1.0f, d0 = 2.0f + q;
    const float n1 = q - 1.0f, d1 = q + 1.0f;
    const float n2 = q - 1.5f, d2 = 1.0f + 1.5f * q;
    const float num = small ? q : r0 ? n0 : r1 ? n1 : r2 ? n2 : -1.0f;
    const float den = small ? 1.0f : r0 ? d0 : r1 ? d1 : r2 ? d2 : q;
    const float hi = r0 ? hi0 : r1 ? hi1 : r2 ? hi2 : hi3;
    const float lo = r0 ? lo0 : r1 ? lo1 : r2 ? lo2 : lo3;
    const float xr = num / den;                                // the range's reduced argument (q itself below 7/16: q / 1)
    const float z2 = xr * xr;
    const float w = z2 * z2;
    const float s1 = z2 * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float ps = xr * (s1 + s2);
    const float res_small = xr - ps;
    const float res_range = hi - ((ps - lo) - xr);
    float z = small ? (tiny ? q : res_small) : res_range;
    z = big ? hi3 + lo3 : z;
    z = k > 60 ? pi_o_2 + 0.5f * pi_lo : (((hx >> 31) && k < -60) ? 0.0f : z);
    const float t = z - pi_lo;
    const float zneg = u2f(f2u(z) ^ 0x80000000u);
    const bool xneg = (hx >> 31) != 0, yneg = (hy >> 31) != 0;
    return xneg ? (yneg ? t - pi : pi - t) : (yneg ? zneg : z);
}

// 4/pi in the 24 windows reduce_large can ask for (inv_pio4(0..23)); a device kernel keeps the table in LDS
FMRX_HD void inv_pio4_table(uint32_t *w24)
{
    for (int i = 0; i < 24; i++) w24[i] = inv_pio4(i);
}

FMRX_HD bool sincosf_large_ok(float y)
{
    const uint32_t top = (f2u(y) >> 20) & 0x7ffu;
    return top >= 0x42fu && top < 0x7f8u;                      // 120 <= |y| < inf
}

FMRX_HD void sincosf_large_flat(float y, const uint32_t *w24, float *sn, float *cs)
{
    const uint32_t xi = f2u(y);
    const int idx = static_cast<int>((xi >> 26) & 15);
    const int shift = static_cast<int>((xi >> 23) & 7);
    uint32_t m = (xi & 0xffffffu) | 0x800000u;
    m <<= shift;
    uint64_t res0 = static_cast<uint32_t>(m * w24[idx]);
    const uint64_t res1 = static_cast<uint64_t>(m) * w24[idx + 4];
    const uint64_t res2 = static_cast<uint64_t>(m) * w24[idx + 8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    const double rx = static_cast<double>(static_cast<int64_t>(res0)) * kPi63;
    const int rn = static_cast<int>(n);
    const int mq = rn + static_cast<int>(xi >> 31);
    const double x2 = rx * rx;
    const double xs = ((mq + 1) & 2) ? -rx : rx;
    const float sp = sin_poly(xs, x2);
    const float cp0 = cos_poly(x2);
    const float cp = (mq & 2) ? -cp0 : cp0;
    *sn = (rn & 1) ? cp : sp;
    *cs = (rn & 1) ? sp : cp;
}

// 120 <= |y| < 2^25 (the PLL's trigArg from a stream's 240th sample to its 67 millionth): reduce_large then only asks for three
// of 4/pi's 24 windows per operand, so the table look-up (a lane-varying LDS read on the recurrence's dependent chain) becomes two
// selects between constants.  Same integers, same floats as sincosf_large_flat.
FMRX_HD bool sincosf_mid_ok(float y)
{
    const uint32_t a = f2u(y) & 0x7fffffffu;
    return a >= 0x42f00000u && a < 0x4c000000u;
}

FMRX_HD void sincosf_mid_flat(float y, float *sn, float *cs)
{
    const uint32_t xi = f2u(y);
    const int idx = static_cast<int>((xi >> 26) & 15);         // 0, 1 or 2 here
    const int shift = static_cast<int>((xi >> 23) & 7);
    uint32_t m = (xi & 0xffffffu) | 0x800000u;
    m <<= shift;
    // inv_pio4(0..2), (4..6), (8..10): windows of 00 00 00 a2 f9 83 6e 4e 44 15 29 fc 27 57 ...
    const uint32_t w0 = idx == 0 ? 0x000000a2u : (idx == 1 ? 0x0000a2f9u : 0x00a2f983u);
    const uint32_t w4 = idx == 0 ? 0xf9836e4eu : (idx == 1 ? 0x836e4e44u : 0x6e4e4415u);
    const uint32_t w8 = idx == 0 ? 0x441529fcu : (idx == 1 ? 0x1529fc27u : 0x29fc2757u);
    uint64_t res0 = static_cast<uint32_t>(m * w0);
    const uint64_t res1 = static_cast<uint64_t>(m) * w4;
    const uint64_t res2 = static_cast<uint64_t>(m) * w8;
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    const double rx = static_cast<double>(static_cast<int64_t>(res0)) * kPi63;
    const int rn = static_cast<int>(n);
    const int mq = rn + static_cast<int>(xi >> 31);
    const double x2 = rx * rx;
    const double xs = ((mq + 1) & 2) ? -rx : rx;
    const float sp = sin_poly(xs, x2);
    const float cp0 = cos_poly(x2);
    const float cp = (mq & 2) ? -cp0 : cp0;
    *sn = (rn & 1) ? cp : sp;
    *cs = (rn & 1) ? sp : cp;
}

}  // namespace glibc235
}  // namespace fmrx

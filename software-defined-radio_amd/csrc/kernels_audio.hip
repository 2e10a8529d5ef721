// kernels_audio.hip -- the audio-rate stage: decimating low-pass FIR on the
// FM-demodulated stream plus s16 packing, one kernel.
//
// Replaces the audio convolveBlockFastFIR of RF_MONO / RF_STEREO
// (src/project.cpp:346, 219, 257 -> src/filter.cpp:158-188) and the PCM
// conversion (src/threadMonoOnly.cpp:185-191).
//
// Same register-blocked packed-FMA structure as kernels_fe.hip, with two
// changes forced by the data: (1) the input is one real stream, so the two
// halves of v_pk_fma_f32 carry two OUTPUTS that are HALF = NT*R apart (the tile
// is split into a low and a high half, staged in LDS as (lo, hi) float pairs);
// (2) float windows are too big for registers (136 samples x 2), so the tile
// is staged PHASE-MAJOR -- LDS[p][i] = pair of sample p + D*i -- and a thread
// streams one polyphase branch at a time: R+ceil(T/D)-1 consecutive pairs
// (ds_read_b128, chunk-padded so lanes 64 B apart do not share a bank).
// Staging reads the stream with coalesced 16-byte loads; samples before the
// block come from the history kept in front of it (negative indices).
//
// Numerics: one FMA per tap in polyphase order.  The generic kernel keeps the
// reference's exact order and serves as the bit-compatible path.
#include "device_math.hpp"
#include "fmrx_internal.hpp"

namespace fmrx {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));

constexpr int kQC = 12;

template <int T, int D, int R, int NT>
struct AuCfg {
    static constexpr int QT = (T + D - 1) / D;
    static constexpr int NC = (QT + kQC - 1) / kQC;
    static constexpr int HALF = NT * R;                 // outputs per half tile
    static constexpr int WL = D * (HALF - 1) + T;       // input samples one half needs
    static constexpr int NI = HALF + QT - 1;            // pairs per polyphase branch
    static constexpr int PAD = R >= 8 ? 16 : 0;         // R=8: lanes 64 B apart would be 4-way bank conflicts; R=4: 2-way, cheaper than the LDS
    static constexpr int CHB = R * 8 + PAD;             // bytes per chunk of R pairs
    static constexpr int NCHK = NI / R + 1;             // +1: the last b128 of the last thread may touch pair NI
    static constexpr int PH_BYTES = NCHK * CHB;
    static constexpr int LDS_BYTES = D * PH_BYTES;
    static constexpr int WT = D * (R - 1) + T;          // samples one thread's outputs span
    static constexpr int NPAIR = R + QT - 1;            // pairs a thread reads per branch
    static constexpr int NRD = (NPAIR + 1) / 2;         // ds_read_b128 per branch
    static constexpr int TABLE = D * NC * kQC;
    static_assert(R % 2 == 0, "pairs are read two at a time");
    static_assert(LDS_BYTES <= 64 * 1024, "tile exceeds the default LDS limit");
};

template <int T, int D, int R, int NT>
__global__ __launch_bounds__(NT) void audio_fir_kernel(const float *__restrict__ xh, long n_in, int delay,
                                                        const float *__restrict__ table, float *__restrict__ y,
                                                        int16_t *__restrict__ pcm, int wrap, long n_out)
{
    using C = AuCfg<T, D, R, NT>;
    extern __shared__ f4 lds4[];
    uint8_t *ldsb = reinterpret_cast<uint8_t *>(lds4);
    const int t = threadIdx.x;
    const long a0 = static_cast<long>(blockIdx.x) * (2 * C::HALF);   // first output of the low half
    const long gbase = D * a0 - (T - 1) - delay;                      // input index of window sample 0 (low half)

    // ---- stage both half-windows phase-major: 16-byte loads, 4 samples per lane per step.
    // xh is 16-byte aligned at index 0, so chunks start at multiples of 4 (also for
    // negative indices = history); `off` = where the window starts inside its first chunk.
    const long gal = gbase & ~3L;
    const int off = static_cast<int>(gbase - gal);
    constexpr int NCH4 = (C::WL + 3) / 4 + 1;
    constexpr int NIT = (2 * NCH4 + NT - 1) / NT;
    f4 v[NIT];
    // all of a thread's loads first (NIT independent 16-byte loads in flight), then the scatter
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int cc = t + it * NT;
        const int half = cc >= NCH4 ? 1 : 0;
        const long g0 = gal + static_cast<long>(half) * (D * C::HALF) + 4L * (cc - half * NCH4);
        v[it] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        if (cc < 2 * NCH4 && g0 < n_in) v[it] = *reinterpret_cast<const f4 *>(xh + g0);
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int cc = t + it * NT;
        if (cc < 2 * NCH4) {
            const int half = cc >= NCH4 ? 1 : 0;
            const int c = cc - half * NCH4;
            const long g0 = gal + static_cast<long>(half) * (D * C::HALF) + 4L * c;
            int j = 4 * c - off;                 // window index of v.x
            int p = (j + 4 * D) % D;             // j >= -3, keep the operand non-negative
            int i = (j + 4 * D) / D - 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float val = (g0 + e < n_in) ? v[it][e] : 0.0f;
                if (j >= 0 && j < C::WL)
                    *reinterpret_cast<float *>(ldsb + p * C::PH_BYTES + (i / R) * C::CHB + (i % R) * 8 + half * 4) = val;
                j++;
                p++;
                if (p == D) {
                    p = 0;
                    i++;
                }
            }
        }
    }
    __syncthreads();

    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};

#pragma unroll
    for (int p = 0; p < D; p++) {
        // the thread's pairs of branch p: i = t*R + ii, ii = 0 .. NPAIR-1
        f2 xs[2 * C::NRD];
#pragma unroll
        for (int k = 0; k < C::NRD; k++) {
            const int ii = 2 * k;
            const f4 v = *reinterpret_cast<const f4 *>(ldsb + p * C::PH_BYTES + (t + ii / R) * C::CHB + (ii % R) * 8);
            xs[ii] = (f2){v.x, v.y};
            xs[ii + 1] = (f2){v.z, v.w};
        }
#pragma unroll
        for (int c = 0; c < C::NC; c++) {
            f8 ha;
            f4 hb;
            asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(ha), "=&s"(hb)
                         : "s"(table), "i"((p * C::NC + c) * kQC * 4), "i"((p * C::NC + c) * kQC * 4 + 32));
            const float hq[kQC] = {ha[0], ha[1], ha[2], ha[3], ha[4], ha[5], ha[6], ha[7], hb[0], hb[1], hb[2], hb[3]};
#pragma unroll
            for (int s = 0; s < R + kQC - 1; s++) {
                const int ii = c * kQC + s;
                if (ii < C::NPAIR && p + D * ii < C::WT) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int q = ii - r;
                        if (q >= c * kQC && q < (c + 1) * kQC && p + D * q < T)
                            acc[r] = __builtin_elementwise_fma(xs[ii], (f2){hq[q - c * kQC], hq[q - c * kQC]}, acc[r]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) asm volatile("" : "+v"(acc[r]));
        }
    }

    // ---- R consecutive outputs in each half; f32 and/or s16 ----
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const long k = a0 + static_cast<long>(half) * C::HALF + static_cast<long>(t) * R;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float v = half ? acc[r].y : acc[r].x;
            if (k + r < n_out) {
                if (y) y[k + r] = v;
                if (pcm) pcm[k + r] = pcm_pack(v, wrap);
            }
        }
    }
}

template <int T, int D>
int launch_fast(const AudioPlan &pl, const float *d_x, size_t n_in, int delay, float *d_y, int16_t *d_pcm, int wrap,
                hipStream_t stream)
{
    constexpr int R = 4, NT = 256;
    using C = AuCfg<T, D, R, NT>;
    const long n_out = static_cast<long>(n_in / D);
    const unsigned grid = static_cast<unsigned>((n_out + 2 * C::HALF - 1) / (2 * C::HALF));
    hipLaunchKernelGGL((audio_fir_kernel<T, D, R, NT>), dim3(grid), dim3(NT), C::LDS_BYTES, stream, d_x,
                       static_cast<long>(n_in), delay, pl.table.p, d_y, d_pcm, wrap, n_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch audio_fir_kernel<%d,%d>: %s", T, D, hipGetErrorString(e));
    return FMRX_OK;
}

template <int T, int D>
void build_table(const float *h, std::vector<float> &tab)
{
    using C = AuCfg<T, D, 4, 256>;
    tab.assign(C::TABLE, 0.0f);
    for (int p = 0; p < D; p++)
        for (int q = 0; q < C::NC * kQC; q++) {
            const int m = p + D * q;
            if (m < T) tab[p * C::NC * kQC + q] = h[T - 1 - m];
        }
}

// audio taps the reference ships (101: threadMonoOnly.cpp:229-232, 13: project.cpp:424-427)
// x the audio_decim of its integer-decimation modes 0 and 1 (5, 6)
#define FMRX_AUDIO_CASES(X) X(101, 5) X(101, 6) X(13, 5) X(13, 6)

}  // namespace

int audio_plan_init(AudioPlan &pl, const float *h, int taps, int decim)
{
    pl.taps = taps;
    pl.decim = decim;
    pl.fast = false;
    FMRX_TRY(pl.h.alloc(taps));
    FMRX_HIP(hipMemcpy(pl.h.p, h, taps * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float> tab;
#define X(T_, D_)                    \
    if (taps == T_ && decim == D_) { \
        build_table<T_, D_>(h, tab); \
        pl.fast = true;              \
    }
    FMRX_AUDIO_CASES(X)
#undef X
    if (pl.fast) {
        FMRX_TRY(pl.table.alloc(tab.size()));
        FMRX_HIP(hipMemcpy(pl.table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return FMRX_OK;
}

int audio_fir_launch(const AudioPlan &pl, const float *d_x, size_t n_in, int delay, float *d_y, int16_t *d_pcm, int wrap,
                     hipStream_t stream, bool force_generic)
{
    if (n_in / pl.decim == 0) return FMRX_OK;
    // the specialised kernel reads 16-byte chunks relative to d_x: d_x must be 16-byte aligned
    if (pl.fast && !force_generic && reinterpret_cast<uintptr_t>(d_x) % 16 == 0) {
#define X(T_, D_) \
    if (pl.taps == T_ && pl.decim == D_) return launch_fast<T_, D_>(pl, d_x, n_in, delay, d_y, d_pcm, wrap, stream);
        FMRX_AUDIO_CASES(X)
#undef X
    }
    FMRX_TRY(k_fir_generic(d_x - delay, n_in / pl.decim, pl.h.p, pl.taps, pl.decim, d_y, stream));
    if (d_pcm) FMRX_TRY(k_pcm16(d_y, n_in / pl.decim, d_pcm, wrap, stream));
    return FMRX_OK;
}

}  // namespace fmrx

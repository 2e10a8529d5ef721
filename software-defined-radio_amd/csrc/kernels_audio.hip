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
// (2) float windows are too big for registers, so a thread streams one
// polyphase branch at a time from LDS: the R+ceil(T/D)-1 pairs p + D*ii, one
// ds_read_b64 each at a compile-time offset from the thread's base.  The tile
// sits in LDS in stream order with one pad pair per R*D pairs, so lanes are
// 8*(R*D+1) bytes apart and a wave's 64 reads hit 64 distinct banks.
// Staging reads the stream with coalesced 16-byte loads (all of a thread's
// loads in flight together); samples before the block come from the history
// kept in front of it (negative indices).
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

#define FMRX_TAPS_ISSUE(ha, hb, table, off) \
    asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4" : "=&s"(ha), "=&s"(hb) : "s"(table), "i"(off), "i"((off) + 32))
#define FMRX_TAPS_WAIT(ha, hb) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ha), "+s"(hb))

template <int T, int D, int R, int NT>
struct AuCfg {
    static constexpr int QT = (T + D - 1) / D;
    static constexpr int NC = (QT + kQC - 1) / kQC;
    static constexpr int HALF = NT * R;                 // outputs per half tile
    static constexpr int WL = D * (HALF - 1) + T;       // input samples one half needs
    static constexpr int PADP = R * D;                  // pairs between pad slots = one thread's stride
    static constexpr int NSLOT = WL + WL / PADP + 2;    // pair slots incl. pads; the last one is a dump slot
    static constexpr int LDS_BYTES = (NSLOT * 8 + 15) / 16 * 16;
    static constexpr int WT = D * (R - 1) + T;          // samples one thread's outputs span
    static constexpr int NPAIR = R + QT - 1;            // pairs a thread reads per branch
    static constexpr int TABLE = D * NC * kQC;
    static_assert(LDS_BYTES <= 64 * 1024, "tile exceeds the default LDS limit");
    static_assert(WL < 65536, "staging uses a 16-bit reciprocal division");
};

// byte offset of window sample j's pair slot
template <int PADP>
__device__ __forceinline__ constexpr int au_slot(int j) { return (j + j / PADP) * 8; }

template <int T, int D, int R, int NT, int G>
__device__ __forceinline__ void au_step(const uint8_t *tb, const float *__restrict__ table, f2 (&acc)[R], f8 &haA, f4 &hbA,
                                        f8 &haB, f4 &hbB)
{
    using C = AuCfg<T, D, R, NT>;
    constexpr int NG = D * C::NC;
    if constexpr (G < NG) {
        constexpr int p = G / C::NC, c = G % C::NC;
        float hq[kQC];
        if constexpr (G % 2 == 0) {
            FMRX_TAPS_WAIT(haA, hbA);
            if constexpr (G + 1 < NG) FMRX_TAPS_ISSUE(haB, hbB, table, (G + 1) * kQC * 4);
#pragma unroll
            for (int k = 0; k < 8; k++) hq[k] = haA[k];
#pragma unroll
            for (int k = 0; k < 4; k++) hq[8 + k] = hbA[k];
        } else {
            FMRX_TAPS_WAIT(haB, hbB);
            if constexpr (G + 1 < NG) FMRX_TAPS_ISSUE(haA, hbA, table, (G + 1) * kQC * 4);
#pragma unroll
            for (int k = 0; k < 8; k++) hq[k] = haB[k];
#pragma unroll
            for (int k = 0; k < 4; k++) hq[8 + k] = hbB[k];
        }
#pragma unroll
        for (int s = 0; s < R + kQC - 1; s++) {
            const int ii = c * kQC + s;
            const int jt = p + D * ii;                   // sample index inside the thread's span
            if (ii < C::NPAIR && jt < C::WT) {
                const f2 xs = *reinterpret_cast<const f2 *>(tb + au_slot<C::PADP>(jt));   // ds_read_b64, immediate offset
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int q = ii - r;
                    if (q >= c * kQC && q < (c + 1) * kQC && p + D * q < T)
                        acc[r] = __builtin_elementwise_fma(xs, (f2){hq[q - c * kQC], hq[q - c * kQC]}, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" : "+v"(acc[r]));
        au_step<T, D, R, NT, G + 1>(tb, table, acc, haA, hbA, haB, hbB);
    }
}

// loads of one tile: NIT independent 16-byte loads per thread, all in flight together
template <int T, int D, int R, int NT, int NIT>
__device__ __forceinline__ void au_fetch(const float *__restrict__ xh, const float *__restrict__ hist_end, long n_in, int delay,
                                         long tile, int t, f4 (&v)[NIT])
{
    using C = AuCfg<T, D, R, NT>;
    constexpr int NCH4 = (C::WL + 3) / 4 + 1;
    const long gbase = D * tile * (2 * C::HALF) - (T - 1) - delay;   // input index of window sample 0 (low half)
    const long gal = gbase & ~3L;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int cc = t + it * NT;
        const int half = cc >= NCH4 ? 1 : 0;
        const long g0 = gal + static_cast<long>(half) * (D * C::HALF) + 4L * (cc - half * NCH4);
        v[it] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        if (cc < 2 * NCH4 && g0 < n_in) {
            if (g0 >= 0 || !hist_end) {
                v[it] = *reinterpret_cast<const f4 *>(xh + g0);   // chunks start at multiples of 4: never straddle 0
            } else {
                // history lives at the tail of the previous block's buffer (no alignment guarantee)
                v[it] = (f4){hist_end[g0], hist_end[g0 + 1], hist_end[g0 + 2], hist_end[g0 + 3]};
            }
        }
    }
}

// Persistent workgroups: the next tile's samples are loaded into registers while
// this tile's FMAs run, so HBM latency is not in front of the barrier.
template <int T, int D, int R, int NT>
__global__ __launch_bounds__(NT) void audio_fir_kernel(const float *__restrict__ xh, const float *__restrict__ hist_end,
                                                        long n_in, int delay, const float *__restrict__ table,
                                                        float *__restrict__ y, int16_t *__restrict__ pcm, int wrap,
                                                        long n_out, long n_tiles)
{
    using C = AuCfg<T, D, R, NT>;
    extern __shared__ f4 lds4[];
    uint8_t *ldsb = reinterpret_cast<uint8_t *>(lds4);
    const int t = threadIdx.x;
    constexpr int NCH4 = (C::WL + 3) / 4 + 1;
    constexpr int NIT = (2 * NCH4 + NT - 1) / NT;
    constexpr int DUMP = (C::NSLOT - 1) * 8;             // where out-of-window elements go

    f4 v[NIT];
    long tile = blockIdx.x;
    if (tile < n_tiles) au_fetch<T, D, R, NT, NIT>(xh, hist_end, n_in, delay, tile, t, v);
    for (; tile < n_tiles; tile += gridDim.x) {
        const long a0 = tile * (2 * C::HALF);            // first output of the low half
        const long gbase = D * a0 - (T - 1) - delay;
        const long gal = gbase & ~3L;
        const int off = static_cast<int>(gbase - gal);   // where the window starts inside its first 16-byte chunk
        // ---- scatter the prefetched samples into (lo,hi) pair slots, stream order + pads ----
        // interior tiles (every chunk inside [0, n_in) or in the history, wave-uniform test) skip the
        // per-element range checks on values; chunks wholly inside the window skip the slot checks too
        const bool interior = gal + 4L * NCH4 + static_cast<long>(D) * C::HALF <= n_in;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int cc = t + it * NT;
            if (cc < 2 * NCH4) {
                const int half = cc >= NCH4 ? 1 : 0;
                const int c = cc - half * NCH4;
                const int j0 = 4 * c - off;                  // window index of v.x; >= -3
                const int jq = ((j0 + C::PADP) * (65536 / C::PADP + 1)) >> 16;   // (j0 + PADP) / PADP
                const int rem = j0 + C::PADP - jq * C::PADP;                     // j0 mod PADP
                const int addr = (j0 + jq - 1) * 8 + half * 4;                   // slot of j0: (j0 + j0/PADP)*8
                if (interior && j0 >= 0 && j0 + 3 < C::WL) {
#pragma unroll
                    for (int e = 0; e < 4; e++)              // +8 once the chunk has crossed into the next thread-stride (pad slot)
                        *reinterpret_cast<float *>(ldsb + addr + 8 * e + (rem + e >= C::PADP ? 8 : 0)) = v[it][e];
                } else {
                    const long g0 = gal + static_cast<long>(half) * (D * C::HALF) + 4L * c;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float val = (g0 + e < n_in) ? v[it][e] : 0.0f;
                        const bool ok = static_cast<unsigned>(j0 + e) < static_cast<unsigned>(C::WL);
                        const int a = addr + 8 * e + (rem + e >= C::PADP ? 8 : 0);
                        *reinterpret_cast<float *>(ldsb + (ok ? a : DUMP)) = val;
                    }
                }
            }
        }
        __syncthreads();
        const long next = tile + gridDim.x;
        if (next < n_tiles) au_fetch<T, D, R, NT, NIT>(xh, hist_end, n_in, delay, next, t, v);

        f2 acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
        {
            const uint8_t *tb = ldsb + t * ((C::PADP + 1) * 8);   // slot of this thread's first sample
            f8 haA, haB;
            f4 hbA, hbB;
            FMRX_TAPS_ISSUE(haA, hbA, table, 0);
            au_step<T, D, R, NT, 0>(tb, table, acc, haA, hbA, haB, hbB);
        }
        __syncthreads();   // all reads of this tile done before the next scatter

        // ---- R consecutive outputs in each half; f32 and/or s16 ----
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const long k = a0 + static_cast<long>(half) * C::HALF + static_cast<long>(t) * R;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float o = half ? acc[r].y : acc[r].x;
                if (k + r < n_out) {
                    if (y) y[k + r] = o;
                    if (pcm) pcm[k + r] = pcm_pack(o, wrap);
                }
            }
        }
    }
}

template <int T, int D>
int launch_fast(const AudioPlan &pl, const float *d_x, const float *d_hist_end, size_t n_in, int delay, float *d_y,
                int16_t *d_pcm, int wrap, hipStream_t stream)
{
    constexpr int R = 4, NT = 256;
    using C = AuCfg<T, D, R, NT>;
    const long n_out = static_cast<long>(n_in / D);
    const long n_tiles = (n_out + 2 * C::HALF - 1) / (2 * C::HALF);
    long per_cu = (160 * 1024) / C::LDS_BYTES;
    if (per_cu > 4) per_cu = 4;
    const unsigned grid = static_cast<unsigned>(n_tiles < 256 * per_cu ? n_tiles : 256 * per_cu);
    hipLaunchKernelGGL((audio_fir_kernel<T, D, R, NT>), dim3(grid), dim3(NT), C::LDS_BYTES, stream, d_x, d_hist_end,
                       static_cast<long>(n_in), delay, pl.table.p, d_y, d_pcm, wrap, n_out, n_tiles);
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
        FMRX_TRY(audio_mfma_table_init(pl, h, taps, decim));
    }
    return FMRX_OK;
}

// the specialised kernel reads 16-byte chunks relative to d_x: d_x must be 16-byte aligned
bool audio_fast_available(const AudioPlan &pl, const float *d_x)
{
    return pl.fast && reinterpret_cast<uintptr_t>(d_x) % 16 == 0;
}

int audio_fir_launch(const AudioPlan &pl, const float *d_x, const float *d_hist_end, size_t n_in, int delay, float *d_y,
                     int16_t *d_pcm, int wrap, hipStream_t stream, bool force_generic)
{
    if (n_in / pl.decim == 0) return FMRX_OK;
    if (audio_fast_available(pl, d_x) && !force_generic) {
#define X(T_, D_) \
    if (pl.taps == T_ && pl.decim == D_) \
        return launch_fast<T_, D_>(pl, d_x, d_hist_end, n_in, delay, d_y, d_pcm, wrap, stream);
        FMRX_AUDIO_CASES(X)
#undef X
    }
    if (d_hist_end) return fail(FMRX_EINVAL, "audio_fir_launch: split history needs the specialised kernel");
    FMRX_TRY(k_fir_generic(d_x - delay, n_in / pl.decim, pl.h.p, pl.taps, pl.decim, d_y, stream));
    if (d_pcm) FMRX_TRY(k_pcm16(d_y, n_in / pl.decim, d_pcm, wrap, stream));
    return FMRX_OK;
}

}  // namespace fmrx

// kernels_fe.hip -- the front-end kernel: raw u8 I/Q -> (u-128)/128 -> T-tap
// low-pass FIR -> decimate by D, I and Q together, one pass over HBM.
//
// Replaces, fused: readStdinBlockData's conversion (src/iofunc.cpp:133), the
// I/Q split (src/project.cpp:98-105) and the two convolveBlockFastFIR calls of
// RF_FrontEnd (src/project.cpp:111,121 -> src/filter.cpp:158-188), which are
// ~90 % of the reference's run time.
//
// Design (CDNA4 / gfx950, wave64, VALU packed-FP32; no MFMA -- this is a 1-D
// vector FIR with a 1-column "B matrix"):
//
//  * HBM traffic is the algorithmic minimum: every input byte is read once by
//    a coalesced 16 B/lane load (the T-1 sample halo between neighbouring
//    tiles is < 0.5 % and L2-resident) and every output float2 written once.
//    Per complex input sample: 2 B in + 8/D B out (2.8 B at D = 10).
//  * A workgroup of NT threads stages the raw BYTES of its tile
//    (D*R*NT + T-1 samples) in LDS -- 41 KB at NT=256, R=8, D=10 -- so the
//    8-bit data is not inflated before it reaches registers.
//  * Each thread produces R = 8 CONSECUTIVE outputs.  Its whole input window
//    (D*(R-1)+T samples = 171 at T=101, D=10) is 22 ds_read_b128 into
//    registers; every byte is then addressed statically (v_cvt_f32_ubyteN on a
//    compile-time register), so there is no per-tap address arithmetic at all.
//  * I and Q ride in the two halves of one v_pk_fma_f32: acc(I,Q) += (xI,xQ)*h.
//    The tap is a wave-uniform SGPR operand (op_sel picks the half of an SGPR
//    pair), so taps cost no VGPRs and no LDS bandwidth.
//  * The computation is ordered by polyphase branch p = j mod D: branch p needs
//    only ceil(T/D) taps (<= 12 SGPRs, one s_load_dwordx8 + one dwordx4) and
//    the R+ceil(T/D)-1 window samples j = p + D*i, each converted once per
//    branch and reused by up to R outputs.  Per 8 outputs: 808 v_pk_fma_f32 +
//    342 v_cvt (T=101, D=10), i.e. ~70 % of VALU issue is useful FMA.
//  * (u-128)/128 costs nothing per tap: the staging pass flips the top bit of
//    every byte (u ^ 0x80 is u-128 as a signed byte; one v_xor per 4 samples,
//    once per tile), the window bytes are converted with the sign-extending
//    v_cvt_f32_i32 (SDWA byte select), and the taps are pre-scaled by 1/128
//    (exact).  Every product h[n]*(u-128)/128 is then the reference's product
//    bit for bit, silence (u = 128) gives exact zeros, and the rounding error
//    scales with the signal instead of with the 128 offset.
//
// Numerics: one fused multiply-add per tap, taps visited branch by branch
// instead of n = 0..T-1; the result differs from the reference's sequential
// multiply/add by a few float32 ulp (tests bound the RMS error; the pipeline's
// audio stays within 1e-4 RMS of the reference, SURVEY 7.3 "Summation order").
#include "fmrx_internal.hpp"

namespace fmrx {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int kQC = 12;  // taps fetched per scalar-load group

template <int T, int D, int R, int NT>
struct FeCfg {
    static constexpr int LEAD = (8 - (T - 1) % 8) % 8;          // samples in front of the window so it starts 16 B aligned
    static constexpr int W = D * (R - 1) + T;                   // samples one thread needs
    static constexpr int NB = (2 * (W + LEAD) + 15) / 16;       // ds_read_b128 per thread
    static constexpr int QT = (T + D - 1) / D;                  // taps per polyphase branch (max)
    static constexpr int NC = (QT + kQC - 1) / kQC;             // scalar-load groups per branch
    static constexpr int NOUT = NT * R;                         // outputs per workgroup
    static constexpr int TSTRIDE = 2 * D * R;                   // LDS byte stride between threads
    static constexpr int TILE_BYTES = TSTRIDE * (NT - 1) + 16 * NB;
    static constexpr int HB = 2 * (T - 1 + LEAD);               // history bytes in front of a block
    static constexpr int TABLE = D * NC * kQC;                  // floats in the tap table
    static_assert(TSTRIDE % 16 == 0, "thread windows must start 16-byte aligned");
    static_assert(HB % 16 == 0, "history must be a whole number of 16-byte chunks");
    static_assert(TILE_BYTES <= 64 * 1024, "tile exceeds the default LDS limit");
};

// Tap table: entry ((p*NC + c)*kQC + qq) = h[T-1 - p - D*(c*kQC+qq)] / 128, or 0
// when that index is out of range.  Window sample j = p + D*i of a thread meets
// output r with tap q = i - r of branch p.

// taps of one (branch, group): wave-uniform, straight into SGPRs.  Inline asm keeps
// the loads where they are written (hipcc otherwise hoists every tap load to the
// kernel entry and spills SGPRs).  Issue and wait are separate statements so the
// next group's taps can be in flight while this group's FMAs run.
#define FMRX_TAPS_ISSUE(ha, hb, table, off) \
    asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4" : "=&s"(ha), "=&s"(hb) : "s"(table), "i"(off), "i"((off) + 32))
#define FMRX_TAPS_WAIT(ha, hb) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ha), "+s"(hb))

// One (branch p, tap group c) step of a thread's arithmetic; G = p*NC + c is a
// template parameter (not a loop variable) because the scalar-load offsets must
// be assembler immediates.  Tap registers are double-buffered: set A on even
// steps, set B on odd ones; the other set is being loaded meanwhile.
template <int T, int D, int R, int NT, int G>
__device__ __forceinline__ void fe_step(const uint32_t (&raw)[FeCfg<T, D, R, NT>::NB * 4], const float *__restrict__ table,
                                        f2 (&acc)[R], f8 &haA, f4 &hbA, f8 &haB, f4 &hbB)
{
    using C = FeCfg<T, D, R, NT>;
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
        // window samples j = p + D*i that meet taps q in [c*kQC, (c+1)*kQC)
#pragma unroll
        for (int ii = 0; ii < R + kQC - 1; ii++) {
            const int i = c * kQC + ii;
            const int j = p + D * i;
            if (j < C::W) {
                const int bo = 2 * (j + C::LEAD);
                const uint32_t w = raw[bo / 4];
                f2 xs;
                // v_cvt_f32_i32_sdwa sext(w) src0_sel:BYTE_n
                if ((bo % 4) == 0) {
                    xs.x = static_cast<float>(static_cast<int8_t>(w & 0xffu));
                    xs.y = static_cast<float>(static_cast<int8_t>((w >> 8) & 0xffu));
                } else {
                    xs.x = static_cast<float>(static_cast<int8_t>((w >> 16) & 0xffu));
                    xs.y = static_cast<float>(static_cast<int8_t>(w >> 24));
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int q = i - r;
                    if (q >= c * kQC && q < (c + 1) * kQC && p + D * q < T) {
                        const float h = hq[q - c * kQC];
                        acc[r] = __builtin_elementwise_fma(xs, (f2){h, h}, acc[r]);  // v_pk_fma_f32
                    }
                }
            }
        }
        // pin this step's FMAs before the next step's (keeps the live set at one
        // branch: ~110 VGPRs, ~45 SGPRs)
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" : "+v"(acc[r]));
        fe_step<T, D, R, NT, G + 1>(raw, table, acc, haA, hbA, haB, hbB);
    }
}

// The arithmetic of one thread: R outputs from its register-resident byte window.
template <int T, int D, int R, int NT>
__device__ __forceinline__ void fe_compute(const uint32_t (&raw)[FeCfg<T, D, R, NT>::NB * 4], const float *__restrict__ table,
                                           f2 (&acc)[R])
{
    f8 haA, haB;
    f4 hbA, hbB;
    FMRX_TAPS_ISSUE(haA, hbA, table, 0);
    fe_step<T, D, R, NT, 0>(raw, table, acc, haA, hbA, haB, hbB);
}

// one 16-byte chunk of a tile window: block bytes, carried history, or silence
// beyond either end; the top bit of every byte is flipped on the way (u8 -> int8)
template <int HB>
__device__ __forceinline__ u4 fe_fetch(const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist, long n_bytes, long g)
{
    const uint32_t flip = 0x80808080u;
    u4 v = {0u, 0u, 0u, 0u};  // u8 128 == 0.0f == signed byte 0
    if (g >= 0) {
        if (g + 16 <= n_bytes) v = *reinterpret_cast<const u4 *>(x + g) ^ flip;
    } else if (hist) {
        v = *reinterpret_cast<const u4 *>(hist + (g + HB)) ^ flip;
    }
    return v;
}

template <int R>
__device__ __forceinline__ void fe_store(f2 *__restrict__ y, long kt, long n_out, const f2 (&acc)[R])
{
    if (kt + R <= n_out) {
        f4 *dst = reinterpret_cast<f4 *>(y + kt);
#pragma unroll
        for (int r = 0; r < R; r += 2) dst[r / 2] = (f4){acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y};
    } else {
#pragma unroll
        for (int r = 0; r < R; r++)
            if (kt + r < n_out) y[kt + r] = acc[r];
    }
}

// Variant 1: one tile per workgroup; latency hiding by occupancy alone.
template <int T, int D, int R, int NT>
__global__ __launch_bounds__(NT) void fe_fir_kernel(const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist,
                                                     long n_bytes, const float *__restrict__ table,
                                                     f2 *__restrict__ y, long n_out)
{
    using C = FeCfg<T, D, R, NT>;
    extern __shared__ u4 lds[];
    const int t = threadIdx.x;
    const long k0 = static_cast<long>(blockIdx.x) * C::NOUT;  // first output of this tile
    const long wbyte0 = 2L * D * k0 - C::HB;                  // byte offset of the tile window, multiple of 16

    // ---- stage the tile's raw bytes: coalesced 16 B per lane ----
    constexpr int NCHUNK = C::TILE_BYTES / 16;
#pragma unroll
    for (int c0i = 0; c0i < NCHUNK; c0i += NT) {
        const int c = c0i + t;
        if (c < NCHUNK) lds[c] = fe_fetch<C::HB>(x, hist, n_bytes, wbyte0 + 16L * c);
    }
    __syncthreads();

    // ---- the thread's window: NB x 16 B, all register-resident ----
    const u4 *lw = reinterpret_cast<const u4 *>(reinterpret_cast<const uint8_t *>(lds) + t * C::TSTRIDE);
    uint32_t raw[C::NB * 4];
#pragma unroll
    for (int i = 0; i < C::NB; i++) {
        const u4 v = lw[i];
        raw[4 * i] = v.x;
        raw[4 * i + 1] = v.y;
        raw[4 * i + 2] = v.z;
        raw[4 * i + 3] = v.w;
    }

    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
    fe_compute<T, D, R, NT>(raw, table, acc);
    fe_store<R>(y, k0 + static_cast<long>(t) * R, n_out, acc);
}

// Variant 2: persistent workgroups; the NEXT tile's bytes are fetched into
// registers while this tile's FMAs run, so HBM latency sits under VALU work
// instead of in front of a barrier.
template <int T, int D, int R, int NT>
__global__ __launch_bounds__(NT) void fe_fir_kernel_pf(const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist,
                                                        long n_bytes, const float *__restrict__ table,
                                                        f2 *__restrict__ y, long n_out, long n_tiles)
{
    using C = FeCfg<T, D, R, NT>;
    extern __shared__ u4 lds[];
    const int t = threadIdx.x;
    constexpr int NCHUNK = C::TILE_BYTES / 16;
    constexpr int NPF = (NCHUNK + NT - 1) / NT;  // 16-byte chunks each thread carries

    u4 pre[NPF];
    long tile = blockIdx.x;
    if (tile < n_tiles) {
        const long wbyte0 = 2L * D * tile * C::NOUT - C::HB;
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int c = k * NT + t;
            pre[k] = (c < NCHUNK) ? fe_fetch<C::HB>(x, hist, n_bytes, wbyte0 + 16L * c) : (u4){0u, 0u, 0u, 0u};
        }
    }
    for (; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int c = k * NT + t;
            if (c < NCHUNK) lds[c] = pre[k];
        }
        __syncthreads();
        const u4 *lw = reinterpret_cast<const u4 *>(reinterpret_cast<const uint8_t *>(lds) + t * C::TSTRIDE);
        uint32_t raw[C::NB * 4];
#pragma unroll
        for (int i = 0; i < C::NB; i++) {
            const u4 v = lw[i];
            raw[4 * i] = v.x;
            raw[4 * i + 1] = v.y;
            raw[4 * i + 2] = v.z;
            raw[4 * i + 3] = v.w;
        }
        __syncthreads();  // every window is in registers: the LDS tile may be overwritten
        const long next = tile + gridDim.x;
        if (next < n_tiles) {
            const long wbyte0 = 2L * D * next * C::NOUT - C::HB;
#pragma unroll
            for (int k = 0; k < NPF; k++) {
                const int c = k * NT + t;
                if (c < NCHUNK) pre[k] = fe_fetch<C::HB>(x, hist, n_bytes, wbyte0 + 16L * c);
            }
        }
        f2 acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
        fe_compute<T, D, R, NT>(raw, table, acc);
        fe_store<R>(y, tile * C::NOUT + static_cast<long>(t) * R, n_out, acc);
    }
}

// 1 = one tile per workgroup, 2 = persistent + register prefetch (default).
// Read per launch so one process can A/B the two (tools/fe_ab.py).
int fe_variant()
{
    const char *e = std::getenv("FMRX_FE_VARIANT");
    return e ? std::atoi(e) : 2;
}

template <int T, int D>
int launch_fast(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, float *d_if,
                hipStream_t stream)
{
    constexpr int R = 8, NT = 256;
    using C = FeCfg<T, D, R, NT>;
    const long n_out = static_cast<long>(n_samples / D);
    const long n_tiles = (n_out + C::NOUT - 1) / C::NOUT;
    if (fe_variant() == 1) {
        hipLaunchKernelGGL((fe_fir_kernel<T, D, R, NT>), dim3(static_cast<unsigned>(n_tiles)), dim3(NT), C::TILE_BYTES,
                           stream, d_iq, d_hist, static_cast<long>(2 * n_samples), pl.table.p,
                           reinterpret_cast<f2 *>(d_if), n_out);
    } else {
        // persistent: as many workgroups as fit at once (LDS allows 160 KiB / tile per CU)
        const long per_cu = (160 * 1024) / C::TILE_BYTES > 8 ? 8 : (160 * 1024) / C::TILE_BYTES;
        const long resident = 256 * (per_cu > 0 ? per_cu : 1);
        const unsigned grid = static_cast<unsigned>(n_tiles < resident ? n_tiles : resident);
        hipLaunchKernelGGL((fe_fir_kernel_pf<T, D, R, NT>), dim3(grid), dim3(NT), C::TILE_BYTES, stream, d_iq, d_hist,
                           static_cast<long>(2 * n_samples), pl.table.p, reinterpret_cast<f2 *>(d_if), n_out, n_tiles);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch fe_fir_kernel<%d,%d>: %s", T, D, hipGetErrorString(e));
    return FMRX_OK;
}

template <int T, int D>
void build_table(const float *h, std::vector<float> &tab)
{
    using C = FeCfg<T, D, 8, 256>;
    tab.assign(C::TABLE, 0.0f);
    for (int p = 0; p < D; p++)
        for (int q = 0; q < C::NC * kQC; q++) {
            const int m = p + D * q;  // distance (in samples) from the oldest sample of the output's span
            if (m < T) tab[p * C::NC * kQC + q] = h[T - 1 - m] * 0.0078125f;  // /128, exact
        }
}

// one row per specialised (taps, decim): the tap counts the reference ships
// (13 / 101 / 151) x the rf_decim of its four modes (10 / 5 / 3)
#define FMRX_FE_CASES(X) X(13, 10) X(101, 10) X(151, 10) X(13, 5) X(101, 5) X(151, 5) X(13, 3) X(101, 3) X(151, 3)

}  // namespace

int fe_hist_bytes(int taps)
{
    const int lead = (8 - (taps - 1) % 8) % 8;
    return 2 * (taps - 1 + lead);
}

int fe_plan_init(FePlan &pl, const float *h, int taps, int decim)
{
    pl.taps = taps;
    pl.decim = decim;
    pl.hist_bytes = fe_hist_bytes(taps);
    pl.fast = false;
    FMRX_TRY(pl.h.alloc(taps));
    FMRX_HIP(hipMemcpy(pl.h.p, h, taps * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float> tab;
#define X(T_, D_)                             \
    if (taps == T_ && decim == D_) {          \
        build_table<T_, D_>(h, tab);          \
        pl.fast = true;                       \
    }
    FMRX_FE_CASES(X)
#undef X
    if (pl.fast) {
        FMRX_TRY(pl.table.alloc(tab.size()));
        FMRX_HIP(hipMemcpy(pl.table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return FMRX_OK;
}

int fe_launch(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, float *d_if,
              hipStream_t stream, bool force_generic)
{
    if (n_samples / pl.decim == 0) return FMRX_OK;
    const bool aligned = (reinterpret_cast<uintptr_t>(d_iq) % 16 == 0) && ((2 * n_samples) % 16 == 0) &&
                         (!d_hist || reinterpret_cast<uintptr_t>(d_hist) % 16 == 0);
    if (pl.fast && aligned && !force_generic) {
#define X(T_, D_) \
    if (pl.taps == T_ && pl.decim == D_) return launch_fast<T_, D_>(pl, d_iq, n_samples, d_hist, d_if, stream);
        FMRX_FE_CASES(X)
#undef X
    }
    return k_fe_generic(d_iq, d_hist, pl.hist_bytes, n_samples, pl.h.p, pl.taps, pl.decim, d_if, stream);
}

}  // namespace fmrx

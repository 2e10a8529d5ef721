// kernels_fe.hip -- the front end: raw u8 I/Q -> (u-128)/128 -> T-tap low-pass
// FIR -> decimate by D (I and Q together) -> FM discriminator, one pass over HBM.
//
// Replaces, fused: readStdinBlockData's conversion (src/iofunc.cpp:133), the
// I/Q split (src/project.cpp:98-105), the two convolveBlockFastFIR calls of
// RF_FrontEnd (src/project.cpp:111,121 -> src/filter.cpp:158-188) -- ~90 % of
// the reference's run time -- and fmDemod
// (src/project.cpp:128 -> src/filter.cpp:248-266).
//
// Two kernels share one arithmetic core (fe_compute):
//   fe_demod_kernel    the pipeline's kernel: FIR + discriminator, one WAVE per
//                      tile, LDS-DMA staging, no workgroup barrier; writes demod
//                      (4/D B per sample) and, on request, the IF stream
//   fe_fir_kernel_pf   IF-only (behind fmrx_fe_run_dev): persistent workgroups,
//                      next tile prefetched into registers
//
// This file is the vector-ALU generation of the front end (the north star's "no MFMA" form): kept,
// tested, selected by the option fe_variant = valu (FMRX_FE_VARIANT=valu at start-up).  The default kernels are the matrix-core ones of
// kernels_fe_mfma.hip (int8 MFMA on the raw bytes, exact integer arithmetic), which are HBM-bound
// where this one is VALU-bound.
//
// The core (CDNA4 / gfx950, wave64, VALU packed FP32):
//
//  * Bytes stay bytes until the register file: a tile's raw I/Q bytes are staged
//    in LDS (10 KB per wave tile), never inflated to floats in memory.  HBM
//    traffic is the algorithmic minimum (PMC: 1.03x): every input byte read once
//    by a coalesced 16 B/lane access, the T-1 sample halo between tiles is < 2 %
//    and L2-resident.
//  * Each thread produces R = 8 CONSECUTIVE outputs.  Its whole input window
//    (D*(R-1)+T samples = 171 at T=101, D=10) is 22 ds_read_b128 into registers;
//    every byte is then addressed statically (v_cvt_f32_i32_sdwa sext(v)
//    src0_sel:BYTE_n on a compile-time register): no per-tap address arithmetic.
//  * I and Q ride in the two halves of one v_pk_fma_f32: acc(I,Q) += (xI,xQ)*h.
//    The tap is a wave-uniform SGPR operand (op_sel picks the half of an SGPR
//    pair): taps cost no VGPRs and no LDS bandwidth.
//  * Work is ordered by polyphase branch p = j mod D: branch p needs only
//    ceil(T/D) taps (<= 12 SGPRs per group, s_load_dwordx8 + dwordx4, double
//    buffered so the next group loads under this group's FMAs) and the
//    R+ceil(T/D)-1 window samples j = p + D*i, each converted once per branch and
//    reused by up to R outputs.  Per 8 outputs: 800 v_pk_fma_f32 + 342 v_cvt +
//    86 v_xor (T=101, D=10; tap h[0] is exactly 0 and skipped).
//  * (u-128)/128: the top bit of every byte is flipped once (u ^ 0x80 is u-128
//    as a signed byte), bytes are converted with sign extension, and the taps are
//    pre-scaled by 1/128 (exact).  Every product h[n]*(u-128)/128 is then the
//    reference's product bit for bit, silence (u = 128) gives exact zeros, and
//    the rounding error scales with the signal, not with the 128 offset.
//
// Numerics: one fused multiply-add per tap, taps visited branch by branch instead
// of n = 0..T-1: IF samples differ from the reference's sequential multiply/add
// by a few float32 ulp (measured 2e-7 relative RMS).  The discriminator keeps the
// reference's operation order except for the division (v_rcp_f32).  Every output
// is produced by the same instruction sequence wherever it is computed, so the
// result does not depend on how the stream is cut into tiles or blocks.
#include "device_math.hpp"
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
                    // m = p + D*q = T-1 is tap h[0], which the sin^2 window makes exactly 0 (fe_plan_init
                    // checks it): skipped, 8 FMAs per tile
                    if (q >= c * kQC && q < (c + 1) * kQC && p + D * q < T - 1) {
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

// Persistent workgroups: the NEXT tile's bytes are fetched into
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

// Variant 3 (the pipeline's kernel): front end + FM discriminator fused, one
// WAVE per tile.  Adds to variant 2:
//  * wave-private tiles: each of the workgroup's 4 waves stages, reads and
//    computes its own 64-thread tile in its own LDS region, so there is no
//    workgroup barrier anywhere -- a wave never waits for another wave;
//  * the discriminator (src/filter.cpp:248-266) runs in the epilogue on the IF
//    samples still in registers.  Output k needs IF[k-1]: inside a thread that
//    is the previous accumulator, across threads one lane shift (ds_bpermute).
//    Lane 0 of every tile exists only to supply IF[k-1] to lane 1: tiles overlap
//    by R outputs (1/64 redundant work) and every IF sample is produced by the
//    same instruction sequence wherever it is computed, so results do not depend
//    on how the stream is cut into tiles or blocks;
//  * the IF stream itself is written only on request (d_if != nullptr): the mono
//    and stereo chains consume demod, which costs 4/D bytes per input sample
//    instead of 8/D.
template <int T, int D, int R>
struct FeWaveCfg {
    using C = FeCfg<T, D, R, 64>;
    static constexpr int STRIDE = 63 * R;                      // new outputs per wave tile
    static constexpr int NCHUNK = C::TILE_BYTES / 16;          // 16-byte chunks of one wave tile
    static constexpr int NPF = (NCHUNK + 63) / 64;             // LDS-DMA instructions per tile (1 KiB each)
    static constexpr int WREGION = NPF * 1024;                 // LDS bytes per wave
    static constexpr int HBX = C::HB + 2 * D * R;              // history bytes in front of a block (lane 0 of tile 0)
    static constexpr int MINW = ((160 * 1024) / (4 * WREGION) >= 3 && T <= 101) ? 3 : 2;   // waves per SIMD to compile for
    static_assert(HBX % 16 == 0, "alignment");
};

__device__ const u4 g_silence = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};  // u8 128 = 0.0f

// source address of one 16-byte chunk of a tile window: block bytes, carried
// history, or the silence constant beyond either end
template <int HB>
__device__ __forceinline__ const uint8_t *fe_src(const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist, long n_bytes,
                                                 long g, bool in_tile)
{
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&g_silence);
    if (in_tile) {
        if (g >= 0) {
            if (g + 16 <= n_bytes) p = x + g;
        } else if (hist) {
            p = hist + (g + HB);
        }
    }
    return p;
}

// stage one wave tile: NPF LDS-DMA instructions, 64 lanes x 16 B each, no VGPR
// destination -- the bytes land in the wave's LDS region while the wave computes.
// Interior tiles (every byte inside the block: all but the first and last few)
// take a wave-uniform fast path: one scalar base, lane*16 + k*1024 offsets.
template <int T, int D, int R>
__device__ __forceinline__ void fe_dma_tile(const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist, long n_bytes,
                                            int w, u4 *wl, int lane)
{
    using W = FeWaveCfg<T, D, R>;
    const long wbyte0 = 2L * D * (static_cast<long>(w) * W::STRIDE - R) - W::C::HB;   // wave-uniform
    if (wbyte0 >= 0 && wbyte0 + W::NPF * 1024L <= n_bytes) {
        const uint8_t *base = x + wbyte0;               // scalar
        const int loff = lane * 16;
#pragma unroll
        for (int k = 0; k < W::NPF; k++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (loff + k * 1024)),
                                             (__attribute__((address_space(3))) void *)(wl + k * 64), 16, 0, 0);
    } else {
#pragma unroll
        for (int k = 0; k < W::NPF; k++) {
            const int c = k * 64 + lane;
            const uint8_t *src = fe_src<W::HBX>(x, hist, n_bytes, wbyte0 + 16L * c, c < W::NCHUNK);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(wl + k * 64), 16, 0, 0);
        }
    }
}

template <int T, int D, int R>
__global__ __launch_bounds__(256, (FeWaveCfg<T, D, R>::MINW)) void fe_demod_kernel(
    const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist, long n_bytes, const float *__restrict__ table,
    const float2 *__restrict__ prev_override, float *__restrict__ demod, f2 *__restrict__ y_if,
    float2 *__restrict__ prev_out, long n_out, int n_wtiles, uint8_t *__restrict__ hist_next, int hist_bytes)
{
    using W = FeWaveCfg<T, D, R>;
    using C = typename W::C;
    extern __shared__ u4 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    u4 *wl = lds + wave * (W::WREGION / 16);                   // this wave's private region
    const int n_waves = static_cast<int>(gridDim.x) * 4;
    // tile index: the same in every lane; tell the compiler (scalar address math, uniform branches)
    int w = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + wave);
    const uint32_t flip = 0x80808080u;                         // u ^ 0x80 = (u - 128) as int8

    // the stream's last bytes become the next block's history (I_state/Q_state of the reference,
    // src/filter.cpp:182-187): one wave copies them; host guarantees n_bytes >= hist_bytes
    if (hist_next && blockIdx.x == 0 && wave == 0)
        for (int i = lane; i < hist_bytes; i += 64) hist_next[i] = x[n_bytes - hist_bytes + i];

    if (w < n_wtiles) fe_dma_tile<T, D, R>(x, hist, n_bytes, w, wl, lane);
    bool two_stores = false;   // wave-uniform: the previous iteration issued exactly its two demod stores after the DMA
    for (; w < n_wtiles; w += n_waves) {
        // The tile's bytes have landed.  vmcnt counts in issue order, so when the only younger
        // operations are the previous tile's two demod stores the wait need not cover them
        // (their ~1 us write acknowledgement would otherwise sit in front of every tile).
        if (two_stores) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u4 *lw = reinterpret_cast<const u4 *>(reinterpret_cast<const uint8_t *>(wl) + lane * C::TSTRIDE);
        uint32_t raw[C::NB * 4];
#pragma unroll
        for (int i = 0; i < C::NB; i++) {
            const u4 v = lw[i];
            raw[4 * i] = v.x ^ flip;
            raw[4 * i + 1] = v.y ^ flip;
            raw[4 * i + 2] = v.z ^ flip;
            raw[4 * i + 3] = v.w ^ flip;
        }
        // every lane's window is in registers before the region is overwritten
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int next = w + n_waves;
        if (next < n_wtiles) fe_dma_tile<T, D, R>(x, hist, n_bytes, next, wl, lane);

        f2 acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
        fe_compute<T, D, R, 64>(raw, table, acc);

        // ---- discriminator on the register-resident IF samples ----
        const long kt = static_cast<long>(w) * W::STRIDE + static_cast<long>(lane - 1) * R;   // first output of this lane (lane 0: previous tile's)
        float pi = __shfl_up(acc[R - 1].x, 1, 64), pq = __shfl_up(acc[R - 1].y, 1, 64);
        if (prev_override && kt == 0) {
            const float2 po = *prev_override;
            pi = po.x;
            pq = po.y;
        }
        float d[R];
        {
            // the discriminator of device_math.hpp::demod_fast, written on (I,Q) pairs so that the
            // squares, the differences and the crossed products are one packed instruction each
            f2 pz = (f2){pi, pq};
#pragma unroll
            for (int r = 0; r < R; r++) {
                const f2 z = acc[r];
                const f2 sq = z * z;                                  // (I*I, Q*Q)
                const float den = sq.x + sq.y;
                const f2 dz = z - pz;                                 // (I-Ip, Q-Qp)
                const f2 cr = z * __builtin_shufflevector(dz, dz, 1, 0);   // (I*(Q-Qp), Q*(I-Ip))
                const float num = cr.x - cr.y;
                const float sc = den < 8.6736174e-19f ? 1.8446744e19f : 1.0f;
                const float q = (num * sc) * __builtin_amdgcn_rcpf(den * sc);
                d[r] = den == 0.0f ? 0.0f : q;
                pz = z;
            }
        }
        // interior tile without the optional IF stream: every lane >= 1 stores, 2 x dwordx4 (>= 2 store
        // instructions in any case, which is the safe direction for the counted wait above)
        two_stores = (y_if == nullptr) && (static_cast<long>(w) * W::STRIDE + 63L * R <= n_out);
        if (lane > 0 && kt < n_out) {
            if (kt + R <= n_out) {
                f4 *dd = reinterpret_cast<f4 *>(demod + kt);
#pragma unroll
                for (int r = 0; r < R; r += 4) dd[r / 4] = (f4){d[r], d[r + 1], d[r + 2], d[r + 3]};
                if (y_if) {
                    f4 *dst = reinterpret_cast<f4 *>(y_if + kt);
#pragma unroll
                    for (int r = 0; r < R; r += 2) dst[r / 2] = (f4){acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y};
                }
                if (prev_out && kt + R == n_out) *prev_out = make_float2(acc[R - 1].x, acc[R - 1].y);
            } else {
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (kt + r < n_out) {
                        demod[kt + r] = d[r];
                        if (y_if) y_if[kt + r] = acc[r];
                        if (prev_out && kt + r == n_out - 1) *prev_out = make_float2(acc[r].x, acc[r].y);
                    }
            }
        }
    }
}

template <int T, int D, int R>
int launch_fused(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, const float *d_prev_override,
                 float *d_demod, float *d_if, float *d_prev_out, uint8_t *d_hist_next, const Options &o, hipStream_t stream)
{
    using W = FeWaveCfg<T, D, R>;
    if (d_hist) d_hist += pl.hist_bytes - W::HBX;   // the kernel reads the last HBX bytes of the history
    const long n_out = static_cast<long>(n_samples / D);
    const long n_wtiles = (n_out + W::STRIDE - 1) / W::STRIDE;
    const long lds_wg = 4L * W::WREGION;
    long per_cu = (160 * 1024) / lds_wg;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    if (o.fe_wgs_per_cu >= 1 && o.fe_wgs_per_cu < per_cu) per_cu = o.fe_wgs_per_cu;   // tuning knob
    const long want = (n_wtiles + 3) / 4;
    const unsigned grid = static_cast<unsigned>(want < 256 * per_cu ? want : 256 * per_cu);
    hipLaunchKernelGGL((fe_demod_kernel<T, D, R>), dim3(grid), dim3(256), static_cast<size_t>(lds_wg), stream, d_iq, d_hist,
                       static_cast<long>(2 * n_samples), pl.table.p, reinterpret_cast<const float2 *>(d_prev_override),
                       d_demod, reinterpret_cast<f2 *>(d_if), reinterpret_cast<float2 *>(d_prev_out), n_out,
                       static_cast<int>(n_wtiles), d_hist_next, pl.hist_bytes);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch fe_demod_kernel<%d,%d>: %s", T, D, hipGetErrorString(e));
    return FMRX_OK;
}

template <int T, int D>
int launch_fast(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, float *d_if,
                hipStream_t stream)
{
    constexpr int R = 8, NT = 256;
    using C = FeCfg<T, D, R, NT>;
    const long n_out = static_cast<long>(n_samples / D);
    const long n_tiles = (n_out + C::NOUT - 1) / C::NOUT;
    {
        // persistent: as many workgroups as fit at once (LDS allows 160 KiB / tile per CU)
        const long per_cu = (160 * 1024) / C::TILE_BYTES > 8 ? 8 : (160 * 1024) / C::TILE_BYTES;
        const long resident = 256 * (per_cu > 0 ? per_cu : 1);
        const unsigned grid = static_cast<unsigned>(n_tiles < resident ? n_tiles : resident);
        hipLaunchKernelGGL((fe_fir_kernel_pf<T, D, R, NT>), dim3(grid), dim3(NT), C::TILE_BYTES, stream, d_iq, d_hist,
                           static_cast<long>(2 * n_samples), pl.table.p, reinterpret_cast<f2 *>(d_if), n_out, n_tiles);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch fe_fir_kernel_pf<%d,%d>: %s", T, D, hipGetErrorString(e));
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

// bytes of history kept in front of a block: the taps-1 samples the FIR needs,
// rounded up to 16 bytes (fe_hist_base), plus decim*kMaxR samples so that the fused
// kernel's lane 0 can recompute the previous block's last IF samples.  Every
// kernel reads the LAST bytes of this buffer that it needs.
constexpr int kMaxR = 8;   // outputs per lane of the fused kernel (12 was measured: no faster, 2 waves/SIMD)
static int fe_hist_base(int taps)
{
    const int lead = (8 - (taps - 1) % 8) % 8;
    return 2 * (taps - 1 + lead);
}
int fe_hist_bytes(int taps, int decim) { return fe_hist_base(taps) + 2 * decim * kMaxR; }

int fe_plan_init(FePlan &pl, const float *h, int taps, int decim)
{
    pl.taps = taps;
    pl.decim = decim;
    pl.hist_bytes = fe_hist_bytes(taps, decim);
    pl.fast = false;
    FMRX_TRY(pl.h.alloc(taps));
    FMRX_HIP(hipMemcpy(pl.h.p, h, taps * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float> tab;
    // the specialised kernels skip tap 0: only for designs whose h[0] is exactly 0 (every
    // impulseResponseLPF output: its window is sin^2(i*pi/T))
#define X(T_, D_)                             \
    if (taps == T_ && decim == D_ && h[0] == 0.0f) { \
        build_table<T_, D_>(h, tab);          \
        pl.fast = true;                       \
    }
    FMRX_FE_CASES(X)
#undef X
    if (pl.fast) {
        FMRX_TRY(pl.table.alloc(tab.size()));
        FMRX_HIP(hipMemcpy(pl.table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return fe_mfma_plan_init(pl, h, taps, decim);
}

int fe_launch(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, float *d_if,
              const Options &o, hipStream_t stream, bool force_generic)
{
    if (n_samples / pl.decim == 0) return FMRX_OK;
    const bool aligned = (reinterpret_cast<uintptr_t>(d_iq) % 16 == 0) && ((2 * n_samples) % 16 == 0) &&
                         (!d_hist || reinterpret_cast<uintptr_t>(d_hist) % 16 == 0);
    if (aligned && !force_generic && o.fe_variant == 0 &&
        fe_mfma_available(pl, d_iq, n_samples, d_hist ? d_hist : pl.silence.p))
        // matrix-core kernel, IF stream only (S1: 2 + 8/D bytes per sample)
        return fe_mfma_launch(pl, d_iq, n_samples, d_hist ? d_hist : pl.silence.p, nullptr, nullptr, d_if, nullptr, nullptr,
                              o, stream);
    if (pl.fast && aligned && !force_generic) {
        // the IF-only kernels read just the last 2*(taps-1+lead) bytes of the history
        const uint8_t *h1 = d_hist ? d_hist + (pl.hist_bytes - fe_hist_base(pl.taps)) : nullptr;
#define X(T_, D_) \
    if (pl.taps == T_ && pl.decim == D_) return launch_fast<T_, D_>(pl, d_iq, n_samples, h1, d_if, stream);
        FMRX_FE_CASES(X)
#undef X
    }
    return k_fe_generic(d_iq, d_hist, pl.hist_bytes, n_samples, pl.h.p, pl.taps, pl.decim, d_if, stream);
}

bool fe_fused_available(const FePlan &pl, const uint8_t *d_iq, size_t n_samples)
{
    return pl.fast && (reinterpret_cast<uintptr_t>(d_iq) % 16 == 0) && ((2 * n_samples) % 16 == 0);
}

int fe_demod_launch(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist,
                    const float *d_prev_override, float *d_demod, float *d_if, float *d_prev_out,
                    uint8_t *d_hist_next, const Options &o, hipStream_t stream)
{
    if (n_samples / pl.decim == 0) return FMRX_OK;
#define X(T_, D_) \
    if (pl.taps == T_ && pl.decim == D_) \
        return launch_fused<T_, D_, 8>(pl, d_iq, n_samples, d_hist, d_prev_override, d_demod, d_if, d_prev_out, \
                                       d_hist_next, o, stream);
    FMRX_FE_CASES(X)
#undef X
    return fail(FMRX_EINVAL, "fe_demod_launch: no specialised kernel for taps=%d decim=%d", pl.taps, pl.decim);
}

}  // namespace fmrx

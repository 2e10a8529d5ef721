// kernels_fe_mfma.hip -- the front end on the matrix cores.
//
// Replaces RF_FrontEnd's hot loop (src/project.cpp:95-128: u8 -> float, de-interleave,
// convolveBlockFastFIR x2 (src/filter.cpp:158-188), fmDemod (src/filter.cpp:248-266)) by ONE
// kernel that is bound by HBM, not by the vector ALUs.
//
// Why integers.  The samples are bytes: (u8-128) is exact in int8.  The taps are float32, but a
// tap times a byte only has to be right to a fraction of a float32 ulp of the SUM.  So the taps
// are quantised once, on the host, to 24-bit fixed point q[k] = round(h[k] * 2^s) (s chosen so
// max|q| < 2^23) and split into three signed base-256 digits q = d2*65536 + d1*256 + d0.  Three
// int8 MFMAs per K-step accumulate sum_k d_i[k]*(u8[k]-128) EXACTLY in int32; the epilogue forms
// (acc2*65536 + acc1*256 + acc0) * 2^-(s+7) with two roundings.  The result is the correctly
// rounded FIR with taps perturbed by <= 2^-(s+1) ~ 4e-9 each: closer to the real-number answer
// than the reference's own 101-term float32 accumulation (tests: <= 2e-6 relative RMS of the
// oracle, as for any reordering of that sum; silence stays exactly 0).
//
// Shape.  One MFMA tile = v_mfma_i32_16x16x64_i8:
//   rows    (M=16) = 8 consecutive IF outputs x {I, Q}
//   columns (N=16) = 16 consecutive groups of 8 outputs (the first one belongs to the previous tile)
//   K              = the raw interleaved byte stream of a column's window; a row's taps sit on
//                    every other byte (I rows on even bytes, Q rows on odd bytes), so the
//                    de-interleave costs nothing and the B operand is the input, untouched
//                    but for one XOR 0x80 per dword.
// A (the digit image of the taps, Toeplitz-shifted per row) lives in VGPRs for the whole kernel.
// Since every lane holds 16 consecutive K bytes of its row/column in both operands, the product
// does not depend on how the hardware numbers k inside a lane.
//
// The discriminator needs the IF sample in front of each output.  Inside a tile that is a lane
// shuffle; across tiles there is no carry: column 0 of every tile recomputes the previous tile's
// last column (1/16 more MFMA work, which is not the bound), so tiles are independent.
//
// Data movement.  Tiles are dealt round-robin to the resident waves, so the chip sweeps the input
// front to back.  LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction, no VGPRs) fills a
// wave-private ring of P+1 tile slots, P tiles ahead of the MFMAs; the B fragments are
// ds_read_b128 at a 16*D-byte column stride.  No workgroup barrier, no de-interleave pass; a
// tile's window overlaps its neighbour's by LEAD + K padding bytes (L2 hits), outputs are
// written once.
#include "device_math.hpp"
#include "fe_mfma_host.hpp"
#include "fmrx_internal.hpp"

#include <cmath>
#include <type_traits>
#include <utility>

namespace fmrx {
namespace {

using i4 = int __attribute__((ext_vector_type(4)));
using f2 = float __attribute__((ext_vector_type(2)));
using f4 = float __attribute__((ext_vector_type(4)));

constexpr int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int T, int D, int PF = 0>
struct MfCfg {
    static constexpr int COL_OUT = 8;                              // outputs per column (x2 channels = 16 rows)
    static constexpr int COLS = 16;                                // column 0 repeats the previous tile's last column:
    static constexpr int TILE_OUT = COL_OUT * (COLS - 1);          //   it only supplies the IF sample in front -> 120 new outputs
    static constexpr int COL_BYTES = 2 * D * COL_OUT;              // input bytes per column = column stride
    static constexpr int STRIDE_BYTES = COL_BYTES * (COLS - 1);    // input bytes from one tile to the next
    static constexpr int FRONT = (2 * (T - 1) + 15) / 16 * 16;     // bytes of a column's window in front of its first output's sample
    static constexpr int WIN = FRONT + 2 * D * (COL_OUT - 1) + 2;  // bytes a column's rows touch
    static constexpr int KSTEPS = (WIN + 63) / 64;
    static constexpr int NDIG = kFeMfmaDigits;
    static constexpr int LEAD = COL_BYTES + FRONT;                 // a tile's window starts this far in front of its first new output
    static constexpr int TILE_WIN = COL_BYTES * (COLS - 1) + 64 * KSTEPS;   // bytes a tile reads
    static constexpr int NPF = TILE_WIN / 1024;                    // full 1 KiB DMA pieces per tile
    static constexpr int REM_LANES = (TILE_WIN % 1024) / 16;       // lanes of the last, partial piece
    static constexpr int NP = NPF + (REM_LANES ? 1 : 0);
    static constexpr int SLOT = NP * 1024;                         // LDS bytes per tile
    static constexpr int P = PF ? PF : iclamp(8192 / TILE_WIN, 2, 8);   // tiles in flight beyond the one being multiplied
    static constexpr int NSLOT = P + 1;
    static constexpr int RING = NSLOT * SLOT;                      // bytes of LDS per wave
    static_assert(TILE_WIN % 16 == 0 && COL_BYTES % 16 == 0, "16-byte fragments");
    static constexpr int YOUNGER = NP * P;                         // DMA pieces younger than a tile's last one, steady state
    static_assert(YOUNGER + 2 * P <= 63, "vmcnt is 6 bits");
};

// f(integral_constant<int, 0>{}), f(integral_constant<int, 1>{}), ... in order: a loop whose index is a compile-time constant
template <class Fn, int... K>
__device__ __forceinline__ void for_each_index(std::integer_sequence<int, K...>, Fn &&f)
{
    (f(std::integral_constant<int, K>{}), ...);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA of NP * 1 KiB (the last piece REM_LANES lanes wide if not full) from byte offset s0 of the block
// to dst (wave-uniform LDS address).  Bytes before the block come from the tail of the history; addresses past
// the block are clamped to its last 16 bytes (what lands there only ever meets zero taps or outputs that are not
// stored), so every instruction is issued with its full, compile-time set of lanes: the counted waits of the
// kernels depend on that.  AUX = 2: non-temporal.
template <int NPF, int REM_LANES, int AUX = 0>
__device__ __forceinline__ void dma_window(const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist_end, long n_bytes,
                                           long s0, uint8_t *dst, int lane)
{
    constexpr int NP = NPF + (REM_LANES ? 1 : 0);
    if (s0 >= 0 && s0 + NP * 1024L <= n_bytes) {
        const uint8_t *base = x + s0;   // scalar base, lane*16 + k*1024 offsets
#pragma unroll
        for (int k = 0; k < NP; k++)
            if (k < NPF || lane < REM_LANES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (lane * 16 + k * 1024)),
                                                 (__attribute__((address_space(3))) void *)(dst + k * 1024), 16, 0, AUX);
    } else {
#pragma unroll
        for (int k = 0; k < NP; k++)
            if (k < NPF || lane < REM_LANES) {
                long off = s0 + k * 1024 + lane * 16;
                if (off > n_bytes - 16) off = n_bytes - 16;
                const uint8_t *src = off < 0 ? hist_end + off : x + off;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(dst + k * 1024), 16, 0, AUX);
            }
    }
}

// one tile's window of the S2/S1 kernel: TILE_WIN bytes from STRIDE_BYTES*tile - LEAD of the block into ring slot rs
template <class C, int AUX = 0>
__device__ __forceinline__ void mf_dma_tile(const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist_end, long n_bytes,
                                            int tile, uint8_t *ring, int rs, int lane)
{
    dma_window<C::NPF, C::REM_LANES, AUX>(x, hist_end, n_bytes, static_cast<long>(tile) * C::STRIDE_BYTES - C::LEAD,
                                          ring + rs * C::SLOT, lane);
}

// DBG (ablation variants, compiled and dispatched only in a -DFMRX_TUNING build: option fe_mfma_tune):
// 1 = no output stores, 2 = no MFMA work, 8 = non-temporal DMA, 16 = general tile loop only (no straight-line interior loop)
template <int T, int D, int MINB, int PF, int DBG = 0>
__global__ __launch_bounds__(256, MINB) void fe_mfma_kernel(
    const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist_end, long n_bytes, const i4 *__restrict__ a_img,
    float scale_lo, const float2 *__restrict__ prev_in, float *__restrict__ demod, float *__restrict__ y_if,
    float2 *__restrict__ prev_out, long n_out, int n_tiles, uint8_t *__restrict__ hist_next, int hist_bytes,
    const float *__restrict__ dhist_src, float *__restrict__ dhist_dst, int dhist_n)
{
    using C = MfCfg<T, D, PF>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    uint8_t *ring = lds_raw + wave * C::RING;                 // this wave's private ring
    // tiles are dealt round-robin over all resident waves, the four waves of a workgroup taking four
    // neighbours: at any moment the chip is reading one contiguous stretch of the input
    const int tstep = static_cast<int>(gridDim.x) * 4;
    int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + wave);

    // the stream's last bytes become the next block's history (I_state/Q_state of the reference,
    // src/filter.cpp:182-187): one wave copies them; host guarantees n_bytes >= hist_bytes
    if (hist_next && tile == 0)
        for (int i = lane; i < hist_bytes; i += 64) hist_next[i] = x[n_bytes - hist_bytes + i];
    // the discriminator history in front of this block's output (the previous block's tail lives in the other
    // buffer): copied here so that the consumers of demod[-k] need no separate copy in front of them
    if (dhist_dst && tile == 1)   // the second wave (wave 0 copies the byte history)
        for (int i = lane; i < dhist_n; i += 64) dhist_dst[i] = dhist_src[i];
    if (tile >= n_tiles) return;

    // taps: KSTEPS x NDIG fragments, resident for the whole kernel
    i4 a[C::KSTEPS][C::NDIG];
#pragma unroll
    for (int j = 0; j < C::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) a[j][d] = a_img[(j * C::NDIG + d) * 64 + lane];
    // The IF sample in front of the block: recomputed from the history like any other (column 0 of
    // tile 0), unless the caller carries one that the history does not explain (after set_state).
    float ovi = 0.0f, ovq = 0.0f;
    const bool override_prev = prev_in != nullptr;
    if (override_prev) {
        const float2 p = *prev_in;
        ovi = p.x;
        ovq = p.y;
    }
    // everything an ordinary load returns is in registers before the first DMA is issued (the
    // compiler drains vmcnt to 0 at the use of a plain load: keep that out of the streaming loop)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < C::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) asm volatile("" : "+v"(a[j][d]));
    asm volatile("" : "+v"(ovi), "+v"(ovq));

#pragma unroll
    for (int k = 0; k < C::P; k++)
        if (tile + k * tstep < n_tiles) mf_dma_tile<C, (DBG & 8) ? 2 : 0>(x, hist_end, n_bytes, tile + k * tstep, ring, k, lane);

    const int col = lane & 15, g = lane >> 4;
    const int lane_off = C::COL_BYTES * col + 16 * g;
    const float scale_hi = scale_lo * 65536.0f;
    const bool with_if = y_if != nullptr;
    const int src_lane = lane >= 16 ? lane - 16 : lane + 47;   // who holds the output in front of this lane's first
    int slot = 0, fill = C::P;                                 // ring slot of this tile / of the tile P ahead
    // ---- interior tiles as straight-line code, NSLOT tiles per loop iteration (ring slots are compile-time constants): the
    //      DMA of the tile P ahead is known to lie inside the block, the tile is full, the counted wait is the steady one.
    //      Same arithmetic as the general iteration below, minus its ~12 scalar branches per tile (the lesson of the fused
    //      kernel: a wave's time per tile is a chain of dependent issue, profiles/round2/04_fused_kernel_ab.txt). ----------
    auto fast_tile = [&](auto uc, auto modec, int t) {
        constexpr int u = decltype(uc)::value;                 // ring slot of tile t; the tile P ahead goes to (u + P) % NSLOT
        constexpr int MODE = decltype(modec)::value;           // 1 = discriminator output, 2 = IF output, 3 = both
        constexpr int FILL = (u + C::P) % C::NSLOT;
        {
            const uint8_t *src = x + (static_cast<long>(t + C::P * tstep) * C::STRIDE_BYTES - C::LEAD);   // wave-uniform
            uint8_t *dst = ring + FILL * C::SLOT;
#pragma unroll
            for (int q = 0; q < C::NP; q++)
                if (q < C::NPF || lane < C::REM_LANES)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (lane * 16 + q * 1024)),
                                                     (__attribute__((address_space(3))) void *)(dst + q * 1024), 16, 0, 0);
        }
        wait_vmcnt<C::YOUNGER + (MODE == 3 ? 2 : 1) * C::P>();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint8_t *bsrc = ring + u * C::SLOT + lane_off;
        i4 b[C::KSTEPS];
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) b[j] = *reinterpret_cast<const i4 *>(bsrc + 64 * j);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        i4 acc[C::NDIG];
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) acc[d] = (i4){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) {
            const i4 bs = b[j] ^ static_cast<int>(0x80808080u);
#pragma unroll
            for (int d = 0; d < C::NDIG; d++) acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j][d], bs, acc[d], 0, 0, 0);
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int lo = acc[0][k];
            if (C::NDIG >= 2) lo += acc[1][k] * 256;
            const float flo = static_cast<float>(lo) * scale_lo;
            v[k] = C::NDIG >= 3 ? __builtin_fmaf(static_cast<float>(acc[2][k]), scale_hi, flo) : flo;
        }
        const long o = static_cast<long>(t) * C::TILE_OUT + C::COL_OUT * (col - 1) + 2 * g;
        float d0 = 0.0f, d1 = 0.0f;
        if constexpr ((MODE & 1) != 0) {
            const float pi = __shfl(v[2], src_lane, 64), pq = __shfl(v[3], src_lane, 64);
            d0 = demod_fast_bounded(v[0], v[1], pi, pq);
            d1 = demod_fast_bounded(v[2], v[3], v[0], v[1]);
        }
        if (col > 0) {
            if constexpr ((MODE & 1) != 0) *reinterpret_cast<f2 *>(demod + o) = (f2){d0, d1};
            if constexpr ((MODE & 2) != 0) *reinterpret_cast<f4 *>(y_if + 2 * o) = (f4){v[0], v[1], v[2], v[3]};
        }
    };
    // tiles tile, tile + tstep, ... of this wave that qualify: index it in [P, it_end) (it < P: the counted wait is another)
    long it_end = 0;
    // measured (tools/fe_mfma_tune_modes.py, whole step): D = 3 0.096 -> 0.084 ms, D = 5 0.041 -> 0.038 ms; D = 10 sits at the
    // streaming rate of the LDS-DMA ring either way (0.0396 vs 0.0408 ms): general loop there
    if (!(DBG & 16) && D < 10 && scale_lo >= 8.8817842e-16f) {    // 2^-50: demod_fast_bounded's precondition
        const long nt = n_tiles, ts = tstep;
        // (a) tile + P*tstep exists and its window is read whole from the block; (b) the tile's outputs and the pair behind exist
        const long last_dma = (n_bytes - C::NP * 1024L + C::LEAD) / C::STRIDE_BYTES;   // last tile whose window fits
        long lim = (nt - 1 < last_dma ? nt - 1 : last_dma) - C::P * ts;               // tile <= lim
        const long full = (n_out - 2) / C::TILE_OUT - 1;                               // (tile+1)*TILE_OUT + 2 <= n_out
        lim = lim < full ? lim : full;
        it_end = lim >= tile ? (lim - tile) / ts + 1 : 0;
    }
    auto fast_loop = [&](auto modec, int &it) {
        for (; it + C::NSLOT <= it_end; it += C::NSLOT, tile += C::NSLOT * tstep)
            for_each_index(std::make_integer_sequence<int, C::NSLOT>{},
                           [&](auto uc) { fast_tile(uc, modec, tile + decltype(uc)::value * tstep); });
    };
    for (int it = 0; tile < n_tiles; tile += tstep, it++) {
        if (it >= C::P && slot == 0 && it + C::NSLOT <= it_end && tile > 0) {
            if (demod && with_if) fast_loop(std::integral_constant<int, 3>{}, it);
            else if (demod) fast_loop(std::integral_constant<int, 1>{}, it);
            else fast_loop(std::integral_constant<int, 2>{}, it);
            if (tile >= n_tiles) break;
        }
        const bool steady = tile + C::P * tstep < n_tiles;     // wave-uniform
        if (steady) mf_dma_tile<C, (DBG & 8) ? 2 : 0>(x, hist_end, n_bytes, tile + C::P * tstep, ring, fill, lane);
        // ---- this tile's window has landed -------------------------------------------------
        // vmcnt counts in issue order.  The operations younger than this tile's last piece are the
        // YOUNGER pieces of the P tiles behind it and the stores of the P iterations in between:
        // exactly one demod store (+ one IF store) each (those tiles came before this one, so they
        // were full).  Under-counting stores only waits longer; without P tiles behind, wait for all.
        if (steady) {
            if (it < C::P) wait_vmcnt<C::YOUNGER>();
            else if (with_if && demod) wait_vmcnt<C::YOUNGER + 2 * C::P>();
            else wait_vmcnt<C::YOUNGER + C::P>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- B fragments: 16 consecutive window bytes per lane and K-step ------------------
        const uint8_t *bsrc = ring + slot * C::SLOT + lane_off;
        i4 b[C::KSTEPS];
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) b[j] = *reinterpret_cast<const i4 *>(bsrc + 64 * j);
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): the slot may be refilled
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- 3 exact int8 products per K-step ------------------------------------------------
        i4 acc[C::NDIG];
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) acc[d] = (i4){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) {
            const i4 bs = b[j] ^ static_cast<int>(0x80808080u);   // u8 ^ 0x80 = (u8 - 128) as int8
            if (DBG & 2) {
                acc[j % C::NDIG] += bs;
                continue;
            }
#pragma unroll
            for (int d = 0; d < C::NDIG; d++) acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j][d], bs, acc[d], 0, 0, 0);
        }

        // ---- IF samples: this lane holds outputs o, o+1 as (I, Q, I, Q) -----------------------
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int lo = acc[0][k];
            if (C::NDIG >= 2) lo += acc[1][k] * 256;              // < 2^31: T*2^14*2^8
            const float flo = static_cast<float>(lo) * scale_lo;
            v[k] = C::NDIG >= 3 ? __builtin_fmaf(static_cast<float>(acc[2][k]), scale_hi, flo) : flo;
        }
        float pi = __shfl(v[2], src_lane, 64), pq = __shfl(v[3], src_lane, 64);
        const long o = static_cast<long>(tile) * C::TILE_OUT + C::COL_OUT * (col - 1) + 2 * g;
        if (override_prev && o == 0) {
            pi = ovi;
            pq = ovq;
        }
        float d0 = 0.0f, d1 = 0.0f;
        if (demod) {                                           // wave-uniform: the IF-only form skips the discriminator
            d0 = demod_fast(v[0], v[1], pi, pq);
            d1 = demod_fast(v[2], v[3], v[0], v[1]);
        }

        if (col > 0 && o < n_out && (!(DBG & 1) || d0 == 1234.5f)) {
            if (o + 1 < n_out) {
                if (demod) *reinterpret_cast<f2 *>(demod + o) = (f2){d0, d1};
                if (with_if) *reinterpret_cast<f4 *>(y_if + 2 * o) = (f4){v[0], v[1], v[2], v[3]};
                if (prev_out && o + 2 == n_out) *prev_out = make_float2(v[2], v[3]);
            } else {
                if (demod) demod[o] = d0;
                if (with_if) *reinterpret_cast<f2 *>(y_if + 2 * o) = (f2){v[0], v[1]};
                if (prev_out) *prev_out = make_float2(v[0], v[1]);
            }
        }
        slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
        fill = fill + 1 == C::NSLOT ? 0 : fill + 1;
    }
}

template <int T, int D, int MINB = 2, int PF = 0, int DBG = 0>
int launch_mfma(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, const float *d_prev,
                float *d_demod, float *d_if, float *d_prev_out, uint8_t *d_hist_next, const Options &o, hipStream_t stream,
                const float *d_dhist_src = nullptr, float *d_dhist_dst = nullptr, int dhist_n = 0)
{
    using C = MfCfg<T, D, PF>;
    if (C::LEAD > pl.hist_bytes) return fail(FMRX_EINVAL, "fe_mfma: history too short");
    const long n_out = static_cast<long>(n_samples / D);
    const long n_tiles = (n_out + C::TILE_OUT - 1) / C::TILE_OUT;
    long wgs_per_cu = (160 * 1024) / (4L * C::RING);
    if (wgs_per_cu > MINB) wgs_per_cu = MINB;
    if (o.fe_wgs_per_cu >= 1 && o.fe_wgs_per_cu < wgs_per_cu) wgs_per_cu = o.fe_wgs_per_cu;   // tuning knob
    const long want = (n_tiles + 3) / 4;
    const long grid = want < 256 * wgs_per_cu ? want : 256 * wgs_per_cu;
    hipLaunchKernelGGL((fe_mfma_kernel<T, D, MINB, PF, DBG>), dim3(static_cast<unsigned>(grid)), dim3(256), 4 * C::RING, stream, d_iq,
                       d_hist + pl.hist_bytes, static_cast<long>(2 * n_samples), reinterpret_cast<const i4 *>(pl.a_img.p),
                       pl.scale_lo, reinterpret_cast<const float2 *>(d_prev), d_demod, d_if,
                       reinterpret_cast<float2 *>(d_prev_out), n_out, static_cast<int>(n_tiles), d_hist_next, pl.hist_bytes,
                       d_dhist_src, d_dhist_dst, dhist_n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch fe_mfma_kernel<%d,%d>: %s", T, D, hipGetErrorString(e));
    return FMRX_OK;
}

// ---- the same front end over a BANK of receivers (channels_stereo.hip) -------------------------------------------
// One launch covers the IF outputs [out_off, out_off + n_row) of EVERY channel's block: tile tau = (channel tau / tiles_per_row,
// tile tau % tiles_per_row of that channel's range).  Input rows are the bank's slots [history | block] (pitch in_pitch
// bytes; the history in front of a block holds the stream's last bytes, so every window is read from one contiguous
// row and no tile is special), output rows the bank's discriminator rows.  Same arithmetic as fe_mfma_kernel's general
// loop (same integers: IF samples are bit-identical to the single-stream kernels'); no IF output, no carried IF sample (column 0
// of a row's first tile recomputes it from the history like any other).  n_row is even (host contract): every tile issues
// exactly one f2 store, which the counted waits rely on.
struct FeBankGeom {
    long in_pitch, in_off;     // bytes between channels' slots; byte offset of output 0's first sample inside a slot
    long out_pitch, out_off;   // floats between channels' rows; offset of output 0 inside a row
    long n_row;                // outputs per channel in this launch
    int tiles_per_row;
};

template <int T, int D>
__global__ __launch_bounds__(256, 2) void fe_mfma_bank_kernel(const uint8_t *__restrict__ x, long n_bytes, const i4 *__restrict__ a_img,
                                                               float scale_lo, float *__restrict__ demod, FeBankGeom bk, int n_tiles)
{
    using C = MfCfg<T, D, 0>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    uint8_t *ring = lds_raw + wave * C::RING;
    const int tstep = static_cast<int>(gridDim.x) * 4;
    int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + wave);
    if (tile >= n_tiles) return;
    i4 a[C::KSTEPS][C::NDIG];
#pragma unroll
    for (int j = 0; j < C::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) a[j][d] = a_img[(j * C::NDIG + d) * 64 + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < C::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) asm volatile("" : "+v"(a[j][d]));
    auto dma = [&](int t, int rs) {
        const int c = t / bk.tiles_per_row, j = t - c * bk.tiles_per_row;
        const long s0 = c * bk.in_pitch + bk.in_off + static_cast<long>(j) * C::STRIDE_BYTES - C::LEAD;
        dma_window<C::NPF, C::REM_LANES, 0>(x, x, n_bytes, s0, ring + rs * C::SLOT, lane);
    };
#pragma unroll
    for (int k = 0; k < C::P; k++)
        if (tile + k * tstep < n_tiles) dma(tile + k * tstep, k);
    const int col = lane & 15, g = lane >> 4;
    const int lane_off = C::COL_BYTES * col + 16 * g;
    const float scale_hi = scale_lo * 65536.0f;
    const int src_lane = lane >= 16 ? lane - 16 : lane + 47;
    int slot = 0, fill = C::P;
    for (int it = 0; tile < n_tiles; tile += tstep, it++) {
        const bool steady = tile + C::P * tstep < n_tiles;
        if (steady) dma(tile + C::P * tstep, fill);
        if (steady) {
            if (it < C::P) wait_vmcnt<C::YOUNGER>();
            else wait_vmcnt<C::YOUNGER + C::P>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint8_t *bsrc = ring + slot * C::SLOT + lane_off;
        i4 b[C::KSTEPS];
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) b[j] = *reinterpret_cast<const i4 *>(bsrc + 64 * j);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        i4 acc[C::NDIG];
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) acc[d] = (i4){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) {
            const i4 bs = b[j] ^ static_cast<int>(0x80808080u);
#pragma unroll
            for (int d = 0; d < C::NDIG; d++) acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j][d], bs, acc[d], 0, 0, 0);
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int lo = acc[0][k];
            if (C::NDIG >= 2) lo += acc[1][k] * 256;
            const float flo = static_cast<float>(lo) * scale_lo;
            v[k] = C::NDIG >= 3 ? __builtin_fmaf(static_cast<float>(acc[2][k]), scale_hi, flo) : flo;
        }
        const float pi = __shfl(v[2], src_lane, 64), pq = __shfl(v[3], src_lane, 64);
        const float d0 = demod_fast(v[0], v[1], pi, pq), d1 = demod_fast(v[2], v[3], v[0], v[1]);
        const int c = tile / bk.tiles_per_row, j = tile - c * bk.tiles_per_row;
        const long ol = static_cast<long>(j) * C::TILE_OUT + C::COL_OUT * (col - 1) + 2 * g;   // output inside this launch's range of the row
        if (col > 0 && ol < bk.n_row) *reinterpret_cast<f2 *>(demod + c * bk.out_pitch + bk.out_off + ol) = (f2){d0, d1};
        slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
        fill = fill + 1 == C::NSLOT ? 0 : fill + 1;
    }
}

// ---- front end + BOTH stereo band-pass filters over a bank of receivers, one kernel --------------------------------
// The fast stereo bank's first two stages in one pass over HBM: the int8 front end above, its discriminator output kept in LDS
// as well as written to its row, and the two convolveBlockFIR calls of RF_STEREO (src/project.cpp:202, 207 -> src/filter.cpp:133-154)
// on the f32 matrix cores -- the audio FIR's scheme of the fused mono kernel with decimation 1: rows = 16 consecutive
// outputs, columns = 15 such groups (a batch = two front-end tiles = 240 outputs), K = the TS-1+15+1 samples a column touches,
// A = the taps, Toeplitz-shifted per row (audio_mfma_build_table(h, TS, 1)), one image per filter, resident in registers.
// 2 x 32 MFMAs per batch next to the two tiles' 36 int8 ones, and the discriminator row is not read back by a second kernel.
// MEASURED, AND NOT THE DEFAULT (option bank_fused = 1 selects it): 435-492 us per 2560 IF samples x 16 384 channels against 206-260 +
// 167-172 us for the two separate kernels -- the f32 matrix cores have the vector ALUs' own peak (v_mfma_f32_16x16x4_f32: 1024
// multiply-adds in 32 cycles = a packed fma's rate; the 16 % the Toeplitz shape wastes on top), so moving the band-pass pair there buys
// only what overlaps, and a wave that multiplies its batch is not streaming.  Kept: it is correct (same envelope test), and it is
// the measurement behind DESIGN.md 4.7's statement of what the fast bank would need.
// Work: persistent waves; a wave's items are (channel, tile) pairs, channel = wave, wave + n_waves, ...; the tiles of a channel's
// row in order (the band-pass filters need the 100 samples in front: the row's history at tile 0, the previous batch after
// that).  LDS per wave: the DMA ring + 368 floats [128 history | 240 batch].  Outputs: discriminator row (f32), 22-54 kHz
// band-pass row (f32), the SIGN of the pilot band-pass output as one byte per sample (what the fast PLL reads, kernels_pll.hip).
template <int T, int D, int TS>
__global__ __launch_bounds__(256, 2) void fe_bpf_bank_kernel(const uint8_t *__restrict__ x, long n_bytes, const i4 *__restrict__ a_img,
                                                              float scale_lo, const float *__restrict__ st_img,
                                                              const float *__restrict__ car_img, float *__restrict__ demod,
                                                              float *__restrict__ bpf, long bpf_pitch, long bpf_off,
                                                              int8_t *__restrict__ car8, long car_pitch, long car_off, FeBankGeom bk,
                                                              int n_channels)
{
    using C = MfCfg<T, D, 0>;
    constexpr int AK = ((TS - 1) + 15 + 1 + 15) / 16 * 4;     // audio_mfma_ksteps(TS, 1)
    constexpr int HB = 128, BATCH = 2 * C::TILE_OUT, BUF = HB + BATCH + 32;   // (+32: the unused 16th column's window reaches past the batch)
    static_assert(TS - 1 <= HB && (HB - (TS - 1)) % 4 == 0 && C::TILE_OUT == 120, "column 0's window starts TS-1 samples in front of the batch, 16-byte aligned");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    uint8_t *ring = lds_raw + wave * (C::RING + BUF * 4);
    float *buf = reinterpret_cast<float *>(ring + C::RING);
    const int n_waves = static_cast<int>(gridDim.x) * 4;
    const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + wave);
    const int tpc = bk.tiles_per_row;
    const int my_ch = wid < n_channels ? (n_channels - wid + n_waves - 1) / n_waves : 0;   // channels of this wave
    const long n_items = static_cast<long>(my_ch) * tpc;
    if (n_items == 0) return;
    i4 a[C::KSTEPS][C::NDIG];
#pragma unroll
    for (int j = 0; j < C::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) a[j][d] = a_img[(j * C::NDIG + d) * 64 + lane];
    float ast[AK], acar[AK];
#pragma unroll
    for (int j = 0; j < AK; j++) {
        ast[j] = st_img[j * 64 + lane];
        acar[j] = car_img[j * 64 + lane];
    }
    for (int i = lane; i < BUF; i += 64) buf[i] = 0.0f;       // finite everywhere: zero taps times stale NaN bits would not be zero
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < C::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) asm volatile("" : "+v"(a[j][d]));
#pragma unroll
    for (int j = 0; j < AK; j++) asm volatile("" : "+v"(ast[j]), "+v"(acar[j]));
    auto dma = [&](long q, int rs) {
        const int cq = static_cast<int>(q / tpc), t = static_cast<int>(q - static_cast<long>(cq) * tpc);
        const long c = wid + static_cast<long>(cq) * n_waves;
        const long s0 = c * bk.in_pitch + bk.in_off + static_cast<long>(t) * C::STRIDE_BYTES - C::LEAD;
        dma_window<C::NPF, C::REM_LANES, 0>(x, x, n_bytes, s0, ring + rs * C::SLOT, lane);
    };
#pragma unroll
    for (int k = 0; k < C::P; k++)
        if (k < n_items) dma(k, k);
    const int col = lane & 15, g = lane >> 4;
    const int lane_off = C::COL_BYTES * col + 16 * g;
    const float scale_hi = scale_lo * 65536.0f;
    const int src_lane = lane >= 16 ? lane - 16 : lane + 47;
    int slot = 0, fill = C::P;
    int t = 0;                                                // tile inside the channel's row
    long c = wid;                                             // channel
    for (long q = 0; q < n_items; q++) {
        const bool steady = q + C::P < n_items;
        // tile 0 of a row: the 128 discriminator samples in front of this launch's share of the row (the row's carried history, or
        // what an earlier launch wrote): plain loads, once per channel, requested in front of this iteration's DMA so that the wait
        // for them does not include it
        float h0 = 0.0f, h1 = 0.0f;
        if (t == 0) {
            const float *hp = demod + c * bk.out_pitch + bk.out_off - HB;
            h0 = hp[lane];
            h1 = hp[64 + lane];
        }
        if (steady) dma(q + C::P, fill);
        if (t == 0) {
            buf[lane] = h0;
            buf[64 + lane] = h1;
        }
        // vmcnt counts in issue order: younger than this item's last DMA piece are the YOUNGER pieces of the P items behind it and the
        // stores of the P iterations in between -- one discriminator store each, and the two stores of a band-pass batch in at
        // least every second one (a batch closes every odd tile and every row's last).  Under-counting only waits longer.
        if (steady) {
            if (q < C::P) wait_vmcnt<C::YOUNGER>();
            else wait_vmcnt<C::YOUNGER + C::P + 2 * (C::P / 2)>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint8_t *bsrc = ring + slot * C::SLOT + lane_off;
        i4 b[C::KSTEPS];
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) b[j] = *reinterpret_cast<const i4 *>(bsrc + 64 * j);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        i4 acc[C::NDIG];
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) acc[d] = (i4){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) {
            const i4 bs = b[j] ^ static_cast<int>(0x80808080u);
#pragma unroll
            for (int d = 0; d < C::NDIG; d++) acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j][d], bs, acc[d], 0, 0, 0);
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int lo = acc[0][k];
            if (C::NDIG >= 2) lo += acc[1][k] * 256;
            const float flo = static_cast<float>(lo) * scale_lo;
            v[k] = C::NDIG >= 3 ? __builtin_fmaf(static_cast<float>(acc[2][k]), scale_hi, flo) : flo;
        }
        const float pi = __shfl(v[2], src_lane, 64), pq = __shfl(v[3], src_lane, 64);
        const float d0 = demod_fast(v[0], v[1], pi, pq), d1 = demod_fast(v[2], v[3], v[0], v[1]);
        const int ot = C::COL_OUT * (col - 1) + 2 * g;             // this lane's first output inside the tile
        const long ol = static_cast<long>(t) * C::TILE_OUT + ot;   // ... inside this launch's share of the row
        if (col > 0) {
            if (ol < bk.n_row) *reinterpret_cast<f2 *>(demod + c * bk.out_pitch + bk.out_off + ol) = (f2){d0, d1};
            *reinterpret_cast<f2 *>(buf + HB + (t & 1) * C::TILE_OUT + ot) = (f2){d0, d1};
        }
        // ---- the band-pass pair over the batch that this tile completes (two tiles; the row's last tile alone if their number is odd) ----
        if ((t & 1) || t == tpc - 1) {
            const long o0 = static_cast<long>(t & ~1) * C::TILE_OUT;   // the batch's first output inside this launch's share of the row
            f4 xs[AK / 4];
#pragma unroll
            for (int k = 0; k < AK / 4; k++) xs[k] = *reinterpret_cast<const f4 *>(buf + (HB - (TS - 1)) + 16 * col + 4 * g + 16 * k);
            f4 ys = (f4){0.0f, 0.0f, 0.0f, 0.0f}, yc = ys;
#pragma unroll
            for (int j = 0; j < AK; j++) {
                const float xv = xs[j / 4][j % 4];
                ys = __builtin_amdgcn_mfma_f32_16x16x4f32(ast[j], xv, ys, 0, 0, 0);
                yc = __builtin_amdgcn_mfma_f32_16x16x4f32(acar[j], xv, yc, 0, 0, 0);
            }
            const long ob = o0 + 16 * col + 4 * g;                 // this lane's 4 consecutive outputs
            if (col < 15 && ob < bk.n_row) {                       // n_row is a multiple of 4 (host contract): whole groups
                *reinterpret_cast<f4 *>(bpf + c * bpf_pitch + bpf_off + ob) = ys;
                uint32_t code = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float cv = yc[r];
                    const bool ord = fabsf(cv) > 1e-20f && fabsf(cv) < 1e20f;
                    code |= (ord ? (cv > 0.0f ? 0x01u : 0xffu) : 0u) << (8 * r);
                }
                *reinterpret_cast<uint32_t *>(car8 + c * car_pitch + car_off + ob) = code;
            }
            // the batch's last 128 samples become the history of the next one
            const float k0 = buf[BATCH + lane], k1 = buf[BATCH + 64 + lane];
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            buf[lane] = k0;
            buf[64 + lane] = k1;
        }
        slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
        fill = fill + 1 == C::NSLOT ? 0 : fill + 1;
        if (++t == tpc) {
            t = 0;
            c += n_waves;
        }
    }
}

// ============================================================================================
// Fused mono chain: RF_FrontEnd + RF_MONO of modes 0/1 (src/project.cpp:82-128, 330-350;
// src/threadMonoOnly.cpp:185-191) in ONE kernel: u8 I/Q in, float audio and/or s16 PCM out.  The
// discriminator output never goes to HBM (it is 1/6 of the front end's traffic and all of the audio
// kernel's input): per input sample 2 bytes are read and 0.12 written.
//
// Each wave owns a contiguous run of tiles (128 IF outputs each; the IF sample in front of a tile is
// carried in registers) and a 2048-float ring of discriminator output in LDS.  Every 2*DA tiles the
// ring holds 256 audio outputs' worth: one batch of the audio FIR as f32 MFMAs
// (v_mfma_f32_16x16x4_f32 = exact fmaf chains): rows = 16 consecutive audio outputs, columns = 16
// such groups, K = the TA-1+15*DA+1 discriminator samples a column touches, A = the audio taps,
// Toeplitz-shifted per row, resident in VGPRs.  A run that does not start the block first computes
// the tile in front of it for the TA-1 samples of audio history ("dry": nothing stored).
template <int T, int D, int TA, int DA, int PF = 0, int DRF = 0, int KPTF = 0>
struct FuCfg {
    using F = MfCfg<T, D>;
    static constexpr int TILE_OUT = 128;
    static constexpr int TILE_BYTES = F::COL_BYTES * 16;
    static constexpr int NPF = TILE_BYTES / 1024;
    static constexpr int REM_LANES = (TILE_BYTES % 1024) / 16;
    static constexpr int NP = NPF + (REM_LANES ? 1 : 0);
    static constexpr int P = PF ? PF : iclamp(6144 / TILE_BYTES, 2, 8);   // tiles in flight beyond the one being multiplied
    // one slot more than tiles in flight: the DMA of tile t+P+1 is issued at the top of iteration t, into the
    // slot tile t-1 left, before this iteration waits for its own data and reads its fragments
    static constexpr int NSLOT = P + 2;
    static constexpr int RING = NSLOT * TILE_BYTES;                // byte ring: slot t holds bytes [t*TILE - FRONT, (t+1)*TILE - FRONT)
    static constexpr int PIECE0 = NPF ? 1024 : TILE_BYTES;
    // a tile reads its own slot and the head of the next one: that head must be one DMA piece
    static_assert(F::WIN - F::COL_BYTES <= PIECE0, "window overlap must fit the next slot's first piece");
    static_assert(64 * F::KSTEPS - F::COL_BYTES <= RING - TILE_BYTES, "K padding laps the ring");
    static constexpr int YOUNGER = (NP - 1) + NP * P;              // DMA pieces younger than (next slot, piece 0) at the wait
    static_assert(YOUNGER <= 63, "vmcnt is 6 bits");
    static constexpr int AB_OUT = 256;                             // audio outputs per batch
    static constexpr int TB = AB_OUT * DA / TILE_OUT;              // tiles per batch
    static_assert(TB * TILE_OUT == AB_OUT * DA, "batches end on tile boundaries");
    static constexpr int AWIN = (TA - 1) + 15 * DA + 1;            // discriminator samples a column's 16 outputs touch
    static constexpr int AK = (AWIN + 15) / 16 * 4;                // K-steps of the 16x16x4 MFMA, in whole groups of 4
    // A finished batch is multiplied in NPH slices of KPT K-steps, one slice per following tile, so
    // that no wave ever stops streaming for a whole batch (all waves would at the same time).  Its
    // window must survive the NPH tiles that are written meanwhile:
    static_assert(AK % 4 == 0 && (TA - 1) % 4 == 0, "window in whole 16-byte groups");
    // K-steps per slice: whole groups of 4 (one ds_read_b128 per lane feeds 4 K-steps, see slice_load)
    static constexpr int KPT = KPTF ? KPTF : ((AK + TB - 3) / (TB - 2) + 3) / 4 * 4;
    static constexpr int NPH = (AK + KPT - 1) / KPT;               // slices per batch, <= TB - 2
    static constexpr int DR_MIN = AB_OUT * DA + TA - 1 + (4 * AK - AWIN) + (NPH + 1) * TILE_OUT;
    static constexpr int DR = DRF ? DRF : (DR_MIN + TILE_OUT - 1) / TILE_OUT * TILE_OUT;   // discriminator ring, floats (whole tiles)
    static_assert(NPH <= TB - 2, "a batch must be done before the next one completes");
    static_assert(AB_OUT * DA + TA - 1 + (4 * AK - AWIN) + (NPH + 1) * TILE_OUT <= DR, "ring too small for the delayed multiply");
    static_assert(DR % TILE_OUT == 0, "tiles must not wrap inside");
    static_assert(TA - 1 <= TILE_OUT - 1, "one dry tile must cover the audio history");
    // the ring's first MIRROR samples are kept a second time behind its end, so that the KPT reads of
    // a slice are one base address + immediate offsets (no wrap inside a slice)
    static constexpr int MIRROR = 4 * KPT;
    static_assert(MIRROR <= TILE_OUT, "the mirror is refreshed by the tile at position 0");
    static constexpr int LDS_WAVE = RING + (DR + MIRROR) * 4;
};

// tile slot u of the fused kernel: bytes [u*TILE - FRONT, (u+1)*TILE - FRONT) of the block into ring slot rs
template <class C, class F>
__device__ __forceinline__ void fu_dma_slot(const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist_end, long n_bytes,
                                            int u, uint8_t *ring, int rs, int lane)
{
    dma_window<C::NPF, C::REM_LANES>(x, hist_end, n_bytes, static_cast<long>(u) * C::TILE_BYTES - F::FRONT,
                                     ring + rs * C::TILE_BYTES, lane);
}

// PF / DRF / KPTF override P / DR / KPT, DBG compiles parts out -- ablation variants (tools/fused_tune.py;
// DESIGN.md section 5 quotes them), compiled and dispatched only in a -DFMRX_TUNING build (option fused_tune).  DBG bits: 1 = no audio
// work at all, 2 = the audio MFMAs replaced by one v_fma each, 4 = no byte flip (wrong results, timing
// only), 8 = audio stores compiled out, 128 = two of the three tap digits only, 256 = four of the six K-steps only (both: wrong results), 16 = per-phase clock stamps of every wave (s_memtime) written to the f32
// audio buffer as 8 longs per wave (tools/fused_phases.py reads them; the audio output is not produced).
template <int T, int D, int TA, int DA, int PF = 0, int DRF = 0, int DBG = 0, int KPTF = 0>
__global__ __launch_bounds__(256, 2) void mono_fused_kernel(
    const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist_end, long n_bytes, const i4 *__restrict__ a_img,
    float scale_lo, const float *__restrict__ au_img, const float2 *__restrict__ prev_in,
    const float *__restrict__ dhist_end, float *__restrict__ demod_tail, int tail_keep, float2 *__restrict__ prev_out,
    float *__restrict__ audio, int16_t *__restrict__ pcm, int wrap, long n_out, int n_tiles, long n_audio, int n_batches,
    int batches_per_wave, uint8_t *__restrict__ hist_next, int hist_bytes)
{
    using C = FuCfg<T, D, TA, DA, PF, DRF, KPTF>;
    using F = typename C::F;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    uint8_t *ring = lds_raw + wave * C::LDS_WAVE;
    float *dring = reinterpret_cast<float *>(ring + C::RING);
    const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + wave);

    if (hist_next && wid == 0)   // I_state/Q_state for the next block (src/filter.cpp:182-187), as raw bytes
        for (int i = lane; i < hist_bytes; i += 64) hist_next[i] = x[n_bytes - hist_bytes + i];

    const int b0 = wid * batches_per_wave;                    // this wave's audio batches [b0, b1) = tiles [t0, t1)
    const int b1 = b0 + batches_per_wave < n_batches ? b0 + batches_per_wave : n_batches;
    if (b0 >= b1) return;
    const int t0 = b0 * C::TB;
    const int t1 = b1 * C::TB < n_tiles ? b1 * C::TB : n_tiles;
    const int tb = t0 > 0 ? t0 - 1 : 0;

    // taps of both filters, resident for the whole kernel
    i4 a[F::KSTEPS][F::NDIG];
#pragma unroll
    for (int j = 0; j < F::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < F::NDIG; d++) a[j][d] = a_img[(j * F::NDIG + d) * 64 + lane];
    float au[C::AK];
#pragma unroll
    for (int j = 0; j < C::AK; j++) au[j] = au_img[j * 64 + lane];
    float ci = 0.0f, cq = 0.0f;
    // Ring position of discriminator sample i: (i - (128*tb - 128)) mod DR, so a tile never wraps
    // inside and the history in front of the run has positive positions.
    // The ring starts finite everywhere: padding taps are zeros, and 0 * (stale NaN bits) is not 0.
    for (int i = lane; i < C::DR + C::MIRROR; i += 64) dring[i] = 0.0f;
    if (t0 == 0) {
        const float2 p = *prev_in;                            // prev_i/prev_q (src/project.cpp:122-126)
        ci = p.x;
        cq = p.y;
        // state_mono (src/project.cpp:346): the previous block's last TA-1 discriminator samples
        for (int i = lane; i < TA - 1; i += 64) dring[C::TILE_OUT - (TA - 1) + i] = dhist_end[i - (TA - 1)];
    }
    // everything an ordinary load returns is in registers / LDS before the first DMA is issued (the
    // compiler drains vmcnt to 0 at the use of a plain load: keep that out of the streaming loop)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < F::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < F::NDIG; d++) asm volatile("" : "+v"(a[j][d]));
#pragma unroll
    for (int j = 0; j < C::AK; j++) asm volatile("" : "+v"(au[j]));
    asm volatile("" : "+v"(ci), "+v"(cq));

    {
        int rs = 0;
        for (int u = tb; u <= tb + C::P && u <= t1; u++) {
            fu_dma_slot<C, F>(x, hist_end, n_bytes, u, ring, rs, lane);
            rs++;
        }
    }

    // DBG & 16: where a wave's time goes, phase by phase (shader clock, summed over its tiles)
    long ph[6] = {0, 0, 0, 0, 0, 0};
    long n_stamped = 0;
    auto stamp = [&](float dep) -> long {
        long c = 0;
        if (DBG & 16) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) : "v"(dep) : "memory");
        return c;
    };
    const long run_start = stamp(0.0f);
    long rt_start = 0;
    if (DBG & 16) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_start)::"memory");
    const int col = lane & 15, g = lane >> 4;
    const int lane_off = F::COL_BYTES * col + 16 * g;
    const float scale_hi = scale_lo * 65536.0f;
    const int src_lane = lane >= 16 ? lane - 16 : lane + 47;   // who holds the output in front of this lane's first
    int wrap_adr[F::KSTEPS];                                   // B fragment addresses of the ring's last slot (its window wraps)
#pragma unroll
    for (int j = 0; j < F::KSTEPS; j++) {
        const int adr = (C::NSLOT - 1) * C::TILE_BYTES + lane_off + 64 * j;
        wrap_adr[j] = adr >= C::RING ? adr - C::RING : adr;
    }
    int slot = 0, fill = C::P + 1;                             // ring slot of tile t / of tile t+P+1
    int dpos = C::TILE_OUT;                                    // ring position of tile t's first sample (wave-uniform)
    // the batch being multiplied (at most one): its accumulators, next slice, window, outputs
    bool pend = false;
    int pend_ph = 0, pend_ws = 0;
    long pend_a0 = 0;
    f4 y0 = (f4){0.0f, 0.0f, 0.0f, 0.0f}, y1 = y0;
    // ---- audio FIR: the pending batch is multiplied one slice of KPT K-steps per tile, its LDS reads
    //      issued with the tile's own and its MFMAs behind the tile's, so they hide behind the front
    //      end's epilogue --------------------------------------------------------------------------------
    // K-step j of a batch, K index kq (= this lane's g) <-> window sample 16*(j/4) + 4*kq + j%4: a lane's
    // operands of 4 consecutive K-steps are 4 consecutive samples, one 16-byte LDS read (the tap image
    // is laid out to match, audio_mfma_table_init).
    f4 xs[C::KPT / 4];
    auto slice_load = [&]() {
        // pend_ws walks the window, 16 samples per group of K-steps; one wrap per slice thanks to the mirror
#pragma unroll
        for (int k = 0; k < C::KPT / 4; k++)
            xs[k] = (DBG & 512) ? (f4){1.0f, 2.0f, 3.0f, 4.0f} : *reinterpret_cast<const f4 *>(dring + pend_ws + 16 * k);   // < DR + MIRROR
        pend_ws += 4 * C::KPT;
        pend_ws = pend_ws >= C::DR ? pend_ws - C::DR : pend_ws;
    };
    auto slice_mfma = [&]() {
#pragma unroll
        for (int ph = 0; ph < ((DBG & 2048) ? 1 : C::NPH); ph++)
            if ((DBG & 2048) || pend_ph == ph) {   // DBG 2048: always the first slice's taps (no chain of compares; wrong results)
#pragma unroll
                for (int j = ph * C::KPT; j < (ph + 1) * C::KPT && j < C::AK; j++) {
                    const float xv = xs[(j - ph * C::KPT) / 4][j % 4];
                    if (DBG & 2) {
                        y0[j & 3] += xv * au[j];
                        continue;
                    }
                    if (j & 1) y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(au[j], xv, y1, 0, 0, 0);
                    else y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(au[j], xv, y0, 0, 0, 0);
                }
            }
    };
    auto store_batch = [&]() {
        const f4 y = y0 + y1;
        long ao = pend_a0 + 4 * g;                                               // this lane's 4 consecutive outputs
        if ((DBG & 32768) && ao + 3 < n_audio) ao = wid * 256L + 16 * col + 4 * g;   // every batch of a wave to one place (timing only)
        if ((DBG & (8 | 16)) && y[0] != 1234.5f) {
        } else if (ao + 3 < n_audio) {
            if (audio) *reinterpret_cast<f4 *>(audio + ao) = y;
            if (pcm) {
                using s4 = short __attribute__((ext_vector_type(4)));
                const s4 pk = (s4){pcm_pack_flat(y[0], wrap), pcm_pack_flat(y[1], wrap), pcm_pack_flat(y[2], wrap), pcm_pack_flat(y[3], wrap)};   // no branches
                if (DBG & 131072) __builtin_nontemporal_store(pk, reinterpret_cast<s4 *>(pcm + ao));
                else *reinterpret_cast<s4 *>(pcm + ao) = pk;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (ao + k < n_audio) {
                    if (audio) audio[ao + k] = y[k];
                    if (pcm) pcm[ao + k] = pcm_pack(y[k], wrap);
                }
        }
    };
    auto slice_done = [&]() {
        pend_ph++;
        if (pend_ph == C::NPH) {
            store_batch();
            pend = false;
        }
    };
    // a finished batch becomes the pending one: batch bt = audio outputs [256 bt, 256 bt + 256) is in the ring, column
    // `col` starts at discriminator sample DA*(256 bt + 16 col) - (TA-1); this lane reads samples 4g..4g+3 of every 16
    auto begin_batch = [&](int bt) {
        pend = true;
        pend_ph = 0;
        pend_a0 = static_cast<long>(bt) * C::AB_OUT + 16 * col;
        int sb = (bt * C::AB_OUT * DA - (TA - 1)) - (C::TILE_OUT * tb - C::TILE_OUT);   // wave-uniform, >= 0
        pend_ws = (sb + 16 * col * DA + 4 * g) % C::DR;
        y0 = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        y1 = y0;
    };
    // ---- interior batches run as straight-line code, one batch of TB tiles per loop iteration: every DMA is a steady
    //      one inside the block, no tile touches the block's tail, the slices of the previous batch ride on tiles
    //      0..NPH-1 with their taps selected at compile time.  A wave's time per tile is mostly a chain of dependent
    //      issue (profiles/round2/04_fused_kernel_ab.txt): the general loop below spends a fifth of it on its ~25
    //      scalar branches per tile.  Tiles [tb, t_fast0) and [t_fast1, t1) stay with the general loop. --------------
    int t_fast1 = 0;                                               // first tile that is not in a straight-line steady batch
    bool fast_last = false;                                        // the run's last batch [t1 - TB, t1) runs straight-line too
    if (!(DBG & 4096) && scale_lo >= 8.8817842e-16f) {             // 2^-50: demod_fast_bounded's precondition
        // tile t may issue the DMA of tile t + P + 1 without looking if that tile is read whole from the block ...
        const long in_x = (n_bytes - C::NP * 1024L + F::FRONT) / C::TILE_BYTES;              // last such tile
        const long no_tail = (n_out - tail_keep) / C::TILE_OUT - 1;                          // ... (t+1)*128 <= n_out - tail_keep
        long lim = t1 - C::P - 1;                                  // ... and tile t + P + 1 <= t1 (this wave's to fetch)
        lim = lim < in_x - C::P - 1 ? lim : in_x - C::P - 1;
        lim = lim < no_tail ? lim : no_tail;                       // tiles t <= lim qualify
        t_fast1 = lim + 1 >= C::TB ? static_cast<int>((lim + 1) / C::TB) * C::TB : 0;
        fast_last = t1 % C::TB == 0 && t1 - C::TB >= t0 && t1 <= in_x && t1 - 1 <= no_tail;
    }
    auto fast_tile = [&](auto kc, auto lastc, int t) {
        constexpr int k = decltype(kc)::value;                     // tile k of its batch
        // the run's last batch: the tiles behind it belong to the next wave, only tile t1's head is fetched
        constexpr bool STEADY = !decltype(lastc)::value || k + C::P + 1 <= C::TB;
        constexpr int WAIT = STEADY ? C::YOUNGER : (C::NP - 1) + C::NP * (C::TB - k - 1);
        constexpr bool EARLY_SLICE = (DBG & 8192) != 0;
        auto slice_k = [&]() {
#pragma unroll
            for (int j = k * C::KPT; j < (k + 1) * C::KPT && j < C::AK; j++) {
                const float xv = xs[(j - k * C::KPT) / 4][j % 4];
                if (DBG & 2) {
                    y0[j & 3] += xv * au[j];
                    continue;
                }
                if (j & 1) y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(au[j], xv, y1, 0, 0, 0);
                else y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(au[j], xv, y0, 0, 0, 0);
            }
        };
        const long ta = stamp(ci);
        if constexpr (STEADY) {
            const uint8_t *src = x + (static_cast<long>(t + C::P + 1) * C::TILE_BYTES - F::FRONT);   // wave-uniform
            uint8_t *dst = ring + fill * C::TILE_BYTES;
#pragma unroll
            for (int q = 0; q < C::NP; q++)
                if (q < C::NPF || lane < C::REM_LANES)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (lane * 16 + q * 1024)),
                                                     (__attribute__((address_space(3))) void *)(dst + q * 1024), 16, 0, 0);
        }
        wait_vmcnt<WAIT>();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const long tb_ = stamp(ci);
        i4 b[F::KSTEPS];
        if (slot != C::NSLOT - 1) {
            const uint8_t *bsrc = ring + slot * C::TILE_BYTES + lane_off;
#pragma unroll
            for (int j = 0; j < F::KSTEPS; j++) b[j] = *reinterpret_cast<const i4 *>(bsrc + 64 * j);
        } else {
#pragma unroll
            for (int j = 0; j < F::KSTEPS; j++) b[j] = *reinterpret_cast<const i4 *>(ring + wrap_adr[j]);
        }
        if constexpr (k < C::NPH) slice_load();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const long tc = stamp(__builtin_bit_cast(float, b[0][0]));
        i4 acc[F::NDIG];
#pragma unroll
        for (int d = 0; d < F::NDIG; d++) acc[d] = (i4){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < F::KSTEPS; j++) {
            const i4 bs = b[j] ^ static_cast<int>(0x80808080u);
#pragma unroll
            for (int d = 0; d < F::NDIG; d++) acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j][d], bs, acc[d], 0, 0, 0);
        }
        // the slice's f32 MFMAs directly behind the front end's: the unpack / discriminator instructions below issue
        // between them (one MFMA occupies the matrix pipe for 8 issue slots)
        if constexpr (k < C::NPH && EARLY_SLICE) slice_k();
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int lo = acc[0][q];
            if (F::NDIG >= 2) lo += acc[1][q] * 256;
            const float flo = static_cast<float>(lo) * scale_lo;
            v[q] = F::NDIG >= 3 ? __builtin_fmaf(static_cast<float>(acc[2][q]), scale_hi, flo) : flo;
        }
        const long td = stamp(v[0] + v[1] + v[2] + v[3]);
        float pi = __shfl(v[2], src_lane, 64), pq = __shfl(v[3], src_lane, 64);
        pi = lane == 0 ? ci : pi;
        pq = lane == 0 ? cq : pq;
        ci = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[2]), 63));
        cq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[3]), 63));
        // 1048576: no division (timing only); 2097152: the general form with its tiny-denominator scaling (A/B)
        const float d0 = (DBG & 1048576) ? v[0] * pq - v[1] * pi : (DBG & 2097152) ? demod_fast(v[0], v[1], pi, pq) : demod_fast_bounded(v[0], v[1], pi, pq);
        const float d1 = (DBG & 1048576) ? v[2] * v[1] - v[3] * v[0] : (DBG & 2097152) ? demod_fast(v[2], v[3], v[0], v[1]) : demod_fast_bounded(v[2], v[3], v[0], v[1]);
        const int ol = F::COL_OUT * col + 2 * g;
        *reinterpret_cast<f2 *>(dring + dpos + ol) = (f2){d0, d1};
        if (dpos == 0 && ol < C::MIRROR) *reinterpret_cast<f2 *>(dring + C::DR + ol) = (f2){d0, d1};
        if constexpr (k < C::NPH && EARLY_SLICE && (DBG & 16384)) {
#pragma unroll
            for (int i = 0; i < F::KSTEPS * F::NDIG; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < C::KPT; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
        }
        const long te = stamp(d0 + d1);
        slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
        fill = fill + 1 == C::NSLOT ? 0 : fill + 1;
        dpos = dpos + C::TILE_OUT == C::DR ? 0 : dpos + C::TILE_OUT;
        if constexpr (k < C::NPH && !EARLY_SLICE) slice_k();
        if constexpr (k == C::NPH - 1) store_batch();
        if (DBG & 16) {
            const long tf = stamp(ci);   // issue time: the slice's MFMAs run on under the next tile
            ph[0] += tb_ - ta;
            ph[1] += tc - tb_;
            ph[2] += td - tc;
            ph[3] += te - td;
            ph[4] += tf - te;
            n_stamped++;
        }
    };
    for (int t = tb;;) {
        if (t >= t0 && (t < t_fast1 || (fast_last && t == t1 - C::TB)) && t % C::TB == 0 && (!pend || pend_ph == 0) && !(DBG & 1)) {
            if (!pend) {
                // nothing is pending in front of the run's first batch: a batch whose outputs are all out of range takes
                // the place (it multiplies whatever finite numbers the ring holds and stores nothing)
                begin_batch(t / C::TB);
                pend_a0 = n_audio;
            }
            for (; t < t_fast1; t += C::TB) {
                for_each_index(std::make_integer_sequence<int, C::TB>{},
                               [&](auto kc) { fast_tile(kc, std::false_type{}, t + decltype(kc)::value); });
                begin_batch(t / C::TB);
            }
            if (fast_last && t == t1 - C::TB) {
                for_each_index(std::make_integer_sequence<int, C::TB>{},
                               [&](auto kc) { fast_tile(kc, std::true_type{}, t + decltype(kc)::value); });
                begin_batch(t / C::TB);
                t += C::TB;
            }
        }
        const bool have_tile = t < t1;
        bool completed = false;
        if (have_tile) {
            const bool sl = pend;                                  // a slice of the pending batch rides along (wave-uniform)
            // keep P+1 tiles in flight behind this one: the slot of tile t-1 is free (its reads were waited for)
            const bool steady = t + C::P + 1 <= t1;
            const long ta = stamp(ci);
            if (steady) fu_dma_slot<C, F>(x, hist_end, n_bytes, t + C::P + 1, ring, fill, lane);
            // tile t's slot and the first piece of slot t+1 have landed (vmcnt counts in issue order; the
            // occasional output stores are not credited, which only waits longer)
            if (steady) wait_vmcnt<C::YOUNGER>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const long tb_ = stamp(ci);

            i4 b[F::KSTEPS];
            if (slot != C::NSLOT - 1) {                            // wave-uniform: the window cannot reach the ring's end
                const uint8_t *bsrc = ring + slot * C::TILE_BYTES + lane_off;
#pragma unroll
                for (int j = 0; j < F::KSTEPS; j++) b[j] = *reinterpret_cast<const i4 *>(bsrc + 64 * j);
            } else {
#pragma unroll
                for (int j = 0; j < F::KSTEPS; j++) b[j] = *reinterpret_cast<const i4 *>(ring + wrap_adr[j]);
            }
            if (sl) slice_load();
            __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): the slot may be refilled
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const long tc = stamp(__builtin_bit_cast(float, b[0][0]));

            i4 acc[F::NDIG];
#pragma unroll
            for (int d = 0; d < F::NDIG; d++) acc[d] = (i4){0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < ((DBG & 256) ? F::KSTEPS - 2 : F::KSTEPS); j++) {
                const i4 bs = (DBG & 4) ? b[j] : b[j] ^ static_cast<int>(0x80808080u);   // u8 ^ 0x80 = (u8 - 128) as int8
#pragma unroll
                for (int d = 0; d < ((DBG & 128) ? F::NDIG - 1 : F::NDIG); d++)
                    acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j][d], bs, acc[d], 0, 0, 0);
            }
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int lo = acc[0][k];
                if (F::NDIG >= 2) lo += acc[1][k] * 256;
                const float flo = static_cast<float>(lo) * scale_lo;
                v[k] = F::NDIG >= 3 ? __builtin_fmaf(static_cast<float>(acc[2][k]), scale_hi, flo) : flo;
            }
            const long td = stamp(v[0] + v[1] + v[2] + v[3]);
            float pi = __shfl(v[2], src_lane, 64), pq = __shfl(v[3], src_lane, 64);
            if (lane == 0) {
                pi = ci;
                pq = cq;
            }
            ci = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[2]), 63));
            cq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[3]), 63));
            const float d0 = demod_fast(v[0], v[1], pi, pq);
            const float d1 = demod_fast(v[2], v[3], v[0], v[1]);

            const int ol = F::COL_OUT * col + 2 * g;               // this lane's first output inside the tile
            const long o = static_cast<long>(t) * C::TILE_OUT + ol;
            *reinterpret_cast<f2 *>(dring + dpos + ol) = (f2){d0, d1};
            if (dpos == 0 && ol < C::MIRROR) *reinterpret_cast<f2 *>(dring + C::DR + ol) = (f2){d0, d1};
            if (t >= t0 && static_cast<long>(t + 1) * C::TILE_OUT > n_out - tail_keep &&   // wave-uniform: last tiles only
                o + 2 > n_out - tail_keep && o < n_out) {
                // state_mono and prev_i/prev_q for the next block (the discriminator's last samples)
                if (demod_tail && o >= n_out - tail_keep) demod_tail[o] = d0;
                if (demod_tail && o + 1 < n_out) demod_tail[o + 1] = d1;
                if (prev_out && o + 2 == n_out) *prev_out = make_float2(v[2], v[3]);
                if (prev_out && o + 1 == n_out) *prev_out = make_float2(v[0], v[1]);
            }
            const long te = stamp(d0 + d1);
            slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
            fill = fill + 1 == C::NSLOT ? 0 : fill + 1;
            dpos = dpos + C::TILE_OUT == C::DR ? 0 : dpos + C::TILE_OUT;
            // the slice's MFMAs go last: issued here they run under the next tile's waits and LDS reads, issued
            // right behind the front end's they would stand in front of its whole epilogue (a wave issues in order)
            if (sl) {
                slice_mfma();
                slice_done();
            }
            if (DBG & 16) {
                const long tf = stamp(ci);   // issue time: the slice's MFMAs run on under the next tile
                ph[0] += tb_ - ta;
                ph[1] += tc - tb_;
                ph[2] += td - tc;
                ph[3] += te - td;
                ph[4] += tf - te;
                n_stamped++;
            }
            completed = !(DBG & 1) && t >= t0 && ((t + 1) % C::TB == 0 || t + 1 == n_tiles);
            t++;
        }

        // ---- all that is left of the pending batch when the next one is already complete or the run
        //      is over ---------------------------------------------------------------------------------
        while (pend && (completed || !have_tile)) {
            slice_load();
            slice_mfma();
            slice_done();
        }
        if (completed) begin_batch((t - 1) / C::TB);
        if (!have_tile && !pend) break;
    }
    if (DBG & 1024) {   // keep the audio taps' registers occupied for the whole run (with DBG & 1: nothing else uses them)
        float keep = 0.0f;
#pragma unroll
        for (int j = 0; j < C::AK; j++) keep += au[j];
        if (keep == 12345.678f && pcm) pcm[0] = 1;
    }
    if ((DBG & 16) && audio && lane == 0) {
        long *rec = reinterpret_cast<long *>(audio) + 8L * wid;
        for (int k = 0; k < 5; k++) rec[k] = ph[k];
        rec[5] = stamp(0.0f) - run_start;
        rec[6] = n_stamped;
        long rt_end = 0;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_end)::"memory");
        rec[7] = rt_end - rt_start;                                // 100 MHz ticks: rec[5] / rec[7] = shader clock / 100 MHz
    }
}

template <int T, int D, int TA, int DA, int PF = 0, int DRF = 0, int DBG = 0, int KPTF = 0>
int launch_fused_mono(const FePlan &fe, const AudioPlan &au, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist,
                      const float *d_prev, const float *d_dhist_end, float *d_demod_tail, int tail_keep, float *d_prev_out,
                      float *d_audio, int16_t *d_pcm, int wrap, uint8_t *d_hist_next, const Options &o, hipStream_t stream)
{
    using C = FuCfg<T, D, TA, DA, PF, DRF, KPTF>;
    (void)o;
    if (C::F::FRONT > fe.hist_bytes) return fail(FMRX_EINVAL, "mono_fused: history too short");
    const long n_out = static_cast<long>(n_samples / D);
    const long n_tiles = (n_out + C::TILE_OUT - 1) / C::TILE_OUT;
    const long n_audio = n_out / DA;
    const long n_batches = (n_audio + C::AB_OUT - 1) / C::AB_OUT;
    long wgs_per_cu = (160 * 1024) / (4L * C::LDS_WAVE);
    if (wgs_per_cu > 2) wgs_per_cu = 2;
    // DBG 65536: one workgroup per CU (one wave per SIMD), forced by asking for more than half of the LDS
    const int lds_bytes = (DBG & 65536) ? (4 * C::LDS_WAVE > 84 * 1024 ? 4 * C::LDS_WAVE : 84 * 1024) : 4 * C::LDS_WAVE;
    if (DBG & 65536) wgs_per_cu = 1;
    // DBG 262144 / 524288: 2 x / 4 x as many, shorter runs (the workgroups then take turns on the CUs)
    const long max_waves = 256 * wgs_per_cu * 4 * ((DBG & 262144) ? 2 : 1) * ((DBG & 524288) ? 4 : 1);
    const long bpw = (n_batches + max_waves - 1) / max_waves;
    const long grid = ((n_batches + bpw - 1) / bpw + 3) / 4;
    if (lds_bytes > 64 * 1024) {   // more dynamic LDS than the default cap: opt in, once per device
        static bool raised[64] = {};
        int dev = 0;
        FMRX_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            FMRX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mono_fused_kernel<T, D, TA, DA, PF, DRF, DBG, KPTF>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
    }
    hipLaunchKernelGGL((mono_fused_kernel<T, D, TA, DA, PF, DRF, DBG, KPTF>), dim3(static_cast<unsigned>(grid)), dim3(256), lds_bytes, stream,
                       d_iq, d_hist + fe.hist_bytes, static_cast<long>(2 * n_samples), reinterpret_cast<const i4 *>(fe.a_img.p),
                       fe.scale_lo, au.mfma_table.p, reinterpret_cast<const float2 *>(d_prev), d_dhist_end, d_demod_tail,
                       tail_keep, reinterpret_cast<float2 *>(d_prev_out), d_audio, d_pcm, wrap, n_out, static_cast<int>(n_tiles),
                       n_audio, static_cast<int>(n_batches), static_cast<int>(bpw), d_hist_next, fe.hist_bytes);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(FMRX_EHIP, "launch mono_fused_kernel<%d,%d,%d,%d>: %s", T, D, TA, DA, hipGetErrorString(e));
    return FMRX_OK;
}

// the reference's two integer-decimation modes: mode 0 = (rf_decim 10, audio_decim 5), mode 1 = (5, 6)
#define FMRX_FUSED_CASES(X)                                                                         \
    X(101, 10, 101, 5) X(151, 10, 101, 5) X(13, 10, 101, 5) X(101, 10, 13, 5) X(151, 10, 13, 5) X(13, 10, 13, 5) \
    X(101, 5, 101, 6) X(151, 5, 101, 6) X(13, 5, 101, 6) X(101, 5, 13, 6) X(151, 5, 13, 6) X(13, 5, 13, 6)

#define FMRX_FE_MFMA_CASES(X) X(13, 10) X(101, 10) X(151, 10) X(13, 5) X(101, 5) X(151, 5) X(13, 3) X(101, 3) X(151, 3)

}  // namespace

int fe_mfma_plan_init(FePlan &pl, const float *h, int taps, int decim)
{
    pl.mfma = false;
    int s = 0;
    if (!fe_mfma_scale(h, taps, kFeMfmaDigits, &s)) return FMRX_OK;     // generic / vector-ALU kernels handle those taps
    std::vector<int8_t> img;
#define X(T_, D_)                                                                          \
    if (taps == T_ && decim == D_) {                                                       \
        static_assert(MfCfg<T_, D_>::NDIG == kFeMfmaDigits, "digit count");                \
        if (fe_mfma_shape(T_, D_).ksteps != MfCfg<T_, D_>::KSTEPS || fe_mfma_shape(T_, D_).front != MfCfg<T_, D_>::FRONT) \
            return fail(FMRX_EINVAL, "fe_mfma: host and device tile shapes disagree");     \
        fe_mfma_build_image(h, T_, D_, s, kFeMfmaDigits, img);                             \
        pl.mfma = true;                                                                    \
    }
    FMRX_FE_MFMA_CASES(X)
#undef X
    if (!pl.mfma) return FMRX_OK;
    pl.scale_lo = static_cast<float>(std::ldexp(1.0, -s - 7));          // 2^-s for the taps, /128 for the samples
    FMRX_TRY(pl.silence.alloc(pl.hist_bytes));
    FMRX_HIP(hipMemset(pl.silence.p, 128, pl.hist_bytes));
    FMRX_TRY(pl.a_img.alloc((img.size() + 3) / 4));
    FMRX_HIP(hipMemcpy(pl.a_img.p, img.data(), img.size(), hipMemcpyHostToDevice));
    return FMRX_OK;
}

bool fe_mfma_available(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist)
{
    return pl.mfma && d_hist && n_samples >= 8 && (reinterpret_cast<uintptr_t>(d_iq) % 16 == 0) &&
           ((2 * n_samples) % 16 == 0) && (reinterpret_cast<uintptr_t>(d_hist) % 16 == 0) && pl.hist_bytes % 16 == 0;
}

int fe_mfma_launch(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, const float *d_prev,
                   float *d_demod, float *d_if, float *d_prev_out, uint8_t *d_hist_next, const Options &o, hipStream_t stream,
                   const float *d_dhist_src, float *d_dhist_dst, int dhist_n)
{
    if (n_samples / pl.decim == 0) return FMRX_OK;
    if (!d_demod && !d_if) return fail(FMRX_EINVAL, "fe_mfma_launch: no output");
#ifdef FMRX_TUNING
    if (const int v = o.fe_mfma_tune) {   // "<workgroups per CU><tiles in flight>", (101,10) only
#define Y(B_, P_) \
    if (pl.taps == 101 && pl.decim == 10 && v == B_ * 10 + P_) \
        return launch_mfma<101, 10, B_, P_>(pl, d_iq, n_samples, d_hist, d_prev, d_demod, d_if, d_prev_out, d_hist_next, o, stream);
        Y(2, 2) Y(2, 3) Y(2, 4) Y(3, 3) Y(4, 3) Y(1, 3)
#undef Y
#define Y(G_) \
    if (pl.taps == 101 && pl.decim == 10 && v == G_ * 100 + 23) \
        return launch_mfma<101, 10, 2, 3, G_>(pl, d_iq, n_samples, d_hist, d_prev, d_demod, d_if, d_prev_out, d_hist_next, o, stream);
        Y(1) Y(2) Y(3) Y(8)
#undef Y
        // D = 3 / 5 with more workgroups per CU: 30000 + D*1000 + MINB*100 + P
#define Y(D_, B_, P_) \
    if (pl.taps == 101 && pl.decim == D_ && v == 30000 + D_ * 1000 + B_ * 100 + P_) \
        return launch_mfma<101, D_, B_, P_>(pl, d_iq, n_samples, d_hist, d_prev, d_demod, d_if, d_prev_out, d_hist_next, o, stream, \
                                            d_dhist_src, d_dhist_dst, dhist_n);
        Y(3, 2, 8) Y(3, 3, 8) Y(3, 4, 8) Y(3, 4, 4) Y(3, 3, 4) Y(3, 4, 6) Y(5, 2, 5) Y(5, 3, 5) Y(5, 3, 3) Y(5, 4, 3) Y(5, 4, 2)
#undef Y
        if (pl.taps == 101 && v == 1600) {   // general loop only (A/B of the straight-line interior loop), any decimation
#define Y(D_) \
    if (pl.decim == D_) return launch_mfma<101, D_, 2, 0, 16>(pl, d_iq, n_samples, d_hist, d_prev, d_demod, d_if, d_prev_out, d_hist_next, o, stream, \
                                                              d_dhist_src, d_dhist_dst, dhist_n);
            Y(10) Y(5) Y(3)
#undef Y
        }
    }
#endif
    // workgroups per CU / tiles in flight: the D = 3 / 5 kernels move few bytes per tile and gain from more waves per SIMD with a
    // shallower ring (tools/fe_mfma_tune_modes.py, whole steps: mode 3 0.0754 -> 0.0705 ms with <3, 4>, the decimate-by-5 step
    // 0.0382 -> 0.0356 ms with <4, 2>); longer filters keep two (their tap image alone is > 64 registers)
#define X(T_, D_) \
    if (pl.taps == T_ && pl.decim == D_) \
        return launch_mfma<T_, D_, (T_ <= 101 && D_ == 3) ? 3 : (T_ <= 101 && D_ == 5) ? 4 : 2, \
                           (T_ <= 101 && D_ == 3) ? 4 : (T_ <= 101 && D_ == 5) ? 2 : 0>( \
            pl, d_iq, n_samples, d_hist, d_prev, d_demod, d_if, d_prev_out, d_hist_next, o, stream, d_dhist_src, d_dhist_dst, dhist_n);
    FMRX_FE_MFMA_CASES(X)
#undef X
    return fail(FMRX_EINVAL, "fe_mfma_launch: no kernel for taps=%d decim=%d", pl.taps, pl.decim);
}

int fe_mfma_bank_lead(const FePlan &pl)
{
#define X(T_, D_) if (pl.taps == T_ && pl.decim == D_) return MfCfg<T_, D_>::LEAD;
    FMRX_FE_MFMA_CASES(X)
#undef X
    return -1;
}

// d_slots: n_channels rows of in_pitch bytes; the IF outputs [k_lo, k_hi) of every row's block (which starts block_off bytes into the
// row, with >= fe_mfma_bank_lead() bytes of the stream in front of it) -> d_demod + c * out_pitch + out_off + k.  total_bytes: size of
// the slots buffer (reads are clamped to it).  wgs_per_cu: cap on resident workgroups per CU (0 = as many as fit).
int fe_mfma_bank_launch(const FePlan &pl, const uint8_t *d_slots, long total_bytes, long in_pitch, long block_off, int n_channels,
                        long k_lo, long k_hi, float *d_demod, long out_pitch, long out_off, int wgs_per_cu_cap, hipStream_t stream)
{
    if (!pl.mfma) return fail(FMRX_EINVAL, "fe_mfma_bank: no matrix-core kernel for taps=%d decim=%d", pl.taps, pl.decim);
    if ((k_hi - k_lo) % 2 || (k_lo * pl.decim * 2) % 16 || in_pitch % 16 || block_off % 16 || out_pitch % 2 || (out_off + k_lo) % 2)
        return fail(FMRX_EINVAL, "fe_mfma_bank: misaligned geometry");
#define X(T_, D_)                                                                                                            if (pl.taps == T_ && pl.decim == D_) {                                                                                       using C = MfCfg<T_, D_>;                                                                                                 FeBankGeom bk;                                                                                                           bk.in_pitch = in_pitch;                                                                                                  bk.in_off = block_off + k_lo * D_ * 2;                                                                                   bk.out_pitch = out_pitch;                                                                                                bk.out_off = out_off + k_lo;                                                                                             bk.n_row = k_hi - k_lo;                                                                                                  bk.tiles_per_row = static_cast<int>((bk.n_row + C::TILE_OUT - 1) / C::TILE_OUT);                                          const long n_tiles = static_cast<long>(bk.tiles_per_row) * n_channels;                                                   if (n_tiles > 0x7fffffffL / 4) return fail(FMRX_EINVAL, "fe_mfma_bank: too many tiles");                                  long wgs_per_cu = (160 * 1024) / (4L * C::RING);                                                                         if (wgs_per_cu > 2) wgs_per_cu = 2;                                                                                      if (wgs_per_cu_cap >= 1 && wgs_per_cu_cap < wgs_per_cu) wgs_per_cu = wgs_per_cu_cap;                                     const long want = (n_tiles + 3) / 4;                                                                                     const long grid = want < 256 * wgs_per_cu ? want : 256 * wgs_per_cu;                                                     hipLaunchKernelGGL((fe_mfma_bank_kernel<T_, D_>), dim3(static_cast<unsigned>(grid)), dim3(256), 4 * C::RING, stream, d_slots,                            total_bytes, reinterpret_cast<const i4 *>(pl.a_img.p), pl.scale_lo, d_demod, bk, static_cast<int>(n_tiles));         hipError_t e = hipGetLastError();                                                                                        if (e != hipSuccess) return fail(FMRX_EHIP, "launch fe_mfma_bank_kernel<%d,%d>: %s", T_, D_, hipGetErrorString(e));         return FMRX_OK;                                                                                                      }
    FMRX_FE_MFMA_CASES(X)
#undef X
    return fail(FMRX_EINVAL, "fe_mfma_bank: no kernel for taps=%d decim=%d", pl.taps, pl.decim);
}

// front end + band-pass pair in one kernel (fe_bpf_bank_kernel): available for rf taps <= 101 and stereo taps <= 101 (register budget)
bool fe_bpf_bank_available(const FePlan &pl, int stereo_taps)
{
    return pl.mfma && (pl.taps == 101 || pl.taps == 13) && (pl.decim == 10 || pl.decim == 5) && (stereo_taps == 101 || stereo_taps == 13);
}

int fe_bpf_tables_init(DevBuf<float> &st_img, DevBuf<float> &car_img, const float *h_st, const float *h_car, int taps)
{
    std::vector<float> tab;
    audio_mfma_build_table(h_st, taps, 1, tab);
    FMRX_TRY(st_img.alloc(tab.size()));
    FMRX_HIP(hipMemcpy(st_img.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    audio_mfma_build_table(h_car, taps, 1, tab);
    FMRX_TRY(car_img.alloc(tab.size()));
    FMRX_HIP(hipMemcpy(car_img.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return FMRX_OK;
}

int fe_bpf_bank_launch(const FePlan &pl, int stereo_taps, const float *d_st_img, const float *d_car_img, const uint8_t *d_slots,
                       long total_bytes, long in_pitch, long block_off, int n_channels, long k_lo, long k_hi, float *d_demod,
                       long out_pitch, long out_off, float *d_bpf, long bpf_pitch, int8_t *d_car8, long car_pitch, int wgs_per_cu_cap,
                       hipStream_t stream)
{
    if (!fe_bpf_bank_available(pl, stereo_taps)) return fail(FMRX_EINVAL, "fe_bpf_bank: no fused kernel for these tap counts");
    if ((k_hi - k_lo) % 4 || k_lo % 4 || (k_lo * pl.decim * 2) % 16 || in_pitch % 16 || block_off % 16 || out_pitch % 4 || (out_off + k_lo) % 4 ||
        bpf_pitch % 4 || car_pitch % 4 || out_off < 128)
        return fail(FMRX_EINVAL, "fe_bpf_bank: misaligned geometry");
#define X(T_, D_, TS_)                                                                                                   \
    if (pl.taps == T_ && pl.decim == D_ && stereo_taps == TS_) {                                                         \
        using C = MfCfg<T_, D_>;                                                                                         \
        FeBankGeom bk;                                                                                                   \
        bk.in_pitch = in_pitch;                                                                                          \
        bk.in_off = block_off + k_lo * D_ * 2;                                                                           \
        bk.out_pitch = out_pitch;                                                                                        \
        bk.out_off = out_off + k_lo;                                                                                     \
        bk.n_row = k_hi - k_lo;                                                                                          \
        bk.tiles_per_row = static_cast<int>((bk.n_row + C::TILE_OUT - 1) / C::TILE_OUT);                                  \
        constexpr size_t lds_wave = C::RING + (128 + 2 * C::TILE_OUT + 32) * 4;                                          \
        long wgs_per_cu = (160 * 1024) / (4L * lds_wave);                                                                \
        if (wgs_per_cu > 2) wgs_per_cu = 2;                                                                              \
        if (wgs_per_cu_cap >= 1 && wgs_per_cu_cap < wgs_per_cu) wgs_per_cu = wgs_per_cu_cap;                             \
        const long want = (n_channels + 3) / 4;                                                                          \
        const long grid = want < 256 * wgs_per_cu ? want : 256 * wgs_per_cu;                                             \
        hipLaunchKernelGGL((fe_bpf_bank_kernel<T_, D_, TS_>), dim3(static_cast<unsigned>(grid)), dim3(256), 4 * lds_wave, stream,  \
                           d_slots, total_bytes, reinterpret_cast<const i4 *>(pl.a_img.p), pl.scale_lo, d_st_img, d_car_img, d_demod, d_bpf,  \
                           bpf_pitch, k_lo, d_car8, car_pitch, k_lo, bk, n_channels);                                    \
        hipError_t e = hipGetLastError();                                                                                \
        if (e != hipSuccess) return fail(FMRX_EHIP, "launch fe_bpf_bank_kernel<%d,%d,%d>: %s", T_, D_, TS_, hipGetErrorString(e)); \
        return FMRX_OK;                                                                                                  \
    }
    X(101, 10, 101) X(101, 5, 101) X(13, 10, 13) X(13, 5, 13) X(101, 10, 13) X(101, 5, 13) X(13, 10, 101) X(13, 5, 101)
#undef X
    return fail(FMRX_EINVAL, "fe_bpf_bank: no kernel for taps=%d decim=%d stereo taps=%d", pl.taps, pl.decim, stereo_taps);
}

// Toeplitz image of the audio taps in A-operand order of v_mfma_f32_16x16x4_f32: [kstep][lane], lane
// (row i = lane&15, k = lane>>4) holds the tap that output i of a column applies to window sample
// w = 16*(kstep/4) + 4*k + kstep%4 (so that a lane's B operands of 4 consecutive K-steps are 4
// consecutive samples), i.e. h[decim*i + taps-1 - w], or 0 outside the filter.
int audio_mfma_table_init(AudioPlan &pl, const float *h, int taps, int decim)
{
    std::vector<float> tab;
    audio_mfma_build_table(h, taps, decim, tab);
    FMRX_TRY(pl.mfma_table.alloc(tab.size()));
    FMRX_HIP(hipMemcpy(pl.mfma_table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return FMRX_OK;
}

bool mono_fused_available(const FePlan &fe, const AudioPlan &au, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist)
{
    if (!fe_mfma_available(fe, d_iq, n_samples, d_hist) || !au.mfma_table.p) return false;
    for (int k = 0; k < 1; k++) {
#define X(T_, D_, TA_, DA_) \
    if (fe.taps == T_ && fe.decim == D_ && au.taps == TA_ && au.decim == DA_) return true;
        FMRX_FUSED_CASES(X)
#undef X
    }
    return false;
}

int mono_fused_launch(const FePlan &fe, const AudioPlan &au, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist,
                      const float *d_prev, const float *d_dhist_end, float *d_demod_tail, int tail_keep, float *d_prev_out,
                      float *d_audio, int16_t *d_pcm, int wrap, uint8_t *d_hist_next, const Options &o, hipStream_t stream)
{
    if (!d_prev || !d_dhist_end) return fail(FMRX_EINVAL, "mono_fused_launch: null argument");
#ifdef FMRX_TUNING
    if (const int v = o.fused_tune) {   // ablation variants, (101,10,101,5) only
#define Y(ID_, P_, DR_, G_, K_)                                                                                            \
    if (fe.taps == 101 && fe.decim == 10 && au.taps == 101 && au.decim == 5 && v == ID_)                                    \
        return launch_fused_mono<101, 10, 101, 5, P_, DR_, G_, K_>(fe, au, d_iq, n_samples, d_hist, d_prev, d_dhist_end,    \
                                                                  d_demod_tail, tail_keep, d_prev_out, d_audio, d_pcm, wrap, \
                                                                  d_hist_next, o, stream);
        Y(2, 2, 0, 0, 0) Y(12, 2, 0, 1, 0) Y(1, 1, 0, 0, 0) Y(82, 2, 0, 8, 0) Y(102, 2, 0, 10, 0) Y(162, 2, 0, 16, 0) Y(224, 2, 0, 0, 24) Y(324, 3, 0, 0, 24) Y(216, 2, 0, 0, 16) Y(212, 2, 0, 0, 12) Y(20482, 2, 0, 2048, 0) Y(262, 2, 0, 26, 0) Y(172, 2, 0, 17, 0) Y(10252, 2, 0, 1025, 0) Y(5222, 2, 0, 522, 0) Y(5122, 2, 0, 512, 0) Y(22, 2, 0, 2, 0) Y(42, 2, 0, 4, 0) Y(1282, 2, 0, 128, 0) Y(2562, 2, 0, 256, 0) Y(3842, 2, 0, 384, 0) Y(3852, 2, 0, 385, 0) Y(40962, 2, 0, 4096, 0) Y(2621442, 2, 0, 262144, 0) Y(5242882, 2, 0, 524288, 0) Y(1310722, 2, 0, 131072, 0) Y(655362, 2, 0, 65536, 0) Y(655372, 2, 0, 65537, 0) Y(327682, 2, 0, 32768, 0) Y(81922, 2, 0, 8192, 0) Y(245762, 2, 0, 24576, 0) Y(3, 0, 0, 2097152, 0) Y(4, 0, 0, 0, 0)
#undef Y
        // mode 1's shape (101,5,101,6): 5000 + DBG
#define Y(G_)                                                                                                              \
    if (fe.taps == 101 && fe.decim == 5 && au.taps == 101 && au.decim == 6 && v == 5000 + G_)                                \
        return launch_fused_mono<101, 5, 101, 6, 0, 0, G_, 0>(fe, au, d_iq, n_samples, d_hist, d_prev, d_dhist_end,         \
                                                              d_demod_tail, tail_keep, d_prev_out, d_audio, d_pcm, wrap,   \
                                                              d_hist_next, o, stream);
        Y(0) Y(2097152) Y(1) Y(2) Y(4) Y(8) Y(16) Y(128) Y(256) Y(384) Y(388) Y(389) Y(400) Y(512) Y(514) Y(4096) Y(65536) Y(1048576) Y(1048960) Y(1049474)
#undef Y
    }
#endif
#define X(T_, D_, TA_, DA_)                                                                                          \
    if (fe.taps == T_ && fe.decim == D_ && au.taps == TA_ && au.decim == DA_)                                         \
        return launch_fused_mono<T_, D_, TA_, DA_>(fe, au, d_iq, n_samples, d_hist, d_prev, d_dhist_end, d_demod_tail, \
                                                   tail_keep, d_prev_out, d_audio, d_pcm, wrap, d_hist_next, o, stream);
    FMRX_FUSED_CASES(X)
#undef X
    return fail(FMRX_EINVAL, "mono_fused_launch: no kernel for this filter shape");
}

}  // namespace fmrx

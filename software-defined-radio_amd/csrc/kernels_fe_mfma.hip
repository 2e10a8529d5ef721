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
//   columns (N=16) = 16 consecutive groups of 8 outputs  ->  128 outputs per tile
//   K              = the raw interleaved byte stream of a column's window; a row's taps sit on
//                    every other byte (I rows on even bytes, Q rows on odd bytes), so the
//                    de-interleave costs nothing and the B operand is the input, untouched
//                    but for one XOR 0x80 per dword.
// A (the digit image of the taps, Toeplitz-shifted per row) lives in VGPRs for the whole kernel.
// Since every lane holds 16 consecutive K bytes of its row/column in both operands, the product
// does not depend on how the hardware numbers k inside a lane.
//
// Data movement.  Each wave owns a contiguous run of tiles and streams its bytes once:
// LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction, no VGPRs) fills a wave-private ring
// of P+1 tile slots, P tiles ahead of the MFMAs; the B fragments are ds_read_b128 at a 16*D-byte
// column stride.  No workgroup barrier, no de-interleave pass, every input byte read once
// (+ FRONT bytes per wave), every output written once.
#include "device_math.hpp"
#include "fmrx_internal.hpp"

#include <cmath>

namespace fmrx {
namespace {

using i4 = int __attribute__((ext_vector_type(4)));
using f2 = float __attribute__((ext_vector_type(2)));
using f4 = float __attribute__((ext_vector_type(4)));

constexpr int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int T, int D>
struct MfCfg {
    static constexpr int COL_OUT = 8;                              // outputs per column (x2 channels = 16 rows)
    static constexpr int COLS = 16;
    static constexpr int TILE_OUT = COL_OUT * COLS;                // 128
    static constexpr int COL_BYTES = 2 * D * COL_OUT;              // input bytes per column = column stride
    static constexpr int TILE_BYTES = COL_BYTES * COLS;            // 256*D
    static constexpr int FRONT = (2 * (T - 1) + 15) / 16 * 16;     // bytes of a column's window in front of its first output's sample
    static constexpr int WIN = FRONT + 2 * D * (COL_OUT - 1) + 2;  // bytes a column's rows touch
    static constexpr int KSTEPS = (WIN + 63) / 64;
    static constexpr int NDIG = kFeMfmaDigits;
    static constexpr int NPF = TILE_BYTES / 1024;                  // full 1 KiB DMA pieces per tile
    static constexpr int REM_LANES = (TILE_BYTES % 1024) / 16;     // lanes of the last, partial piece
    static constexpr int NP = NPF + (REM_LANES ? 1 : 0);
    static constexpr int P = iclamp(8192 / TILE_BYTES, 2, 10);     // tiles in flight ahead of the one being multiplied
    static constexpr int NSLOT = P + 1;
    static constexpr int RING = NSLOT * TILE_BYTES;                // bytes of LDS per wave
    static constexpr int PIECE0 = NPF ? 1024 : TILE_BYTES;
    // a tile reads its own slot and the head of the next one: that head must be one DMA piece
    static_assert(WIN - COL_BYTES <= PIECE0, "window overlap must fit the next slot's first piece");
    static_assert(64 * KSTEPS - COL_BYTES <= RING - TILE_BYTES, "K padding laps the ring");
    static_assert(TILE_BYTES % 16 == 0 && COL_BYTES % 16 == 0, "16-byte fragments");
    // vmcnt budget: DMA pieces younger than (next slot, piece 0) in steady state
    static constexpr int YOUNGER = (NP - 1) + NP * (P - 1);
    static_assert(YOUNGER + 2 * P <= 63, "vmcnt is 6 bits");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA of tile slot u (bytes [u*TILE - FRONT, (u+1)*TILE - FRONT) of the block) into ring slot rs.
// Bytes before the block come from the tail of the history; addresses past the block are clamped
// to its last 16 bytes (what lands there only ever meets zero taps or outputs that are not stored),
// so every instruction is issued with its full, compile-time set of lanes: the counted waits
// below depend on that.
template <class C>
__device__ __forceinline__ void mf_dma_slot(const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist_end, long n_bytes,
                                            int u, uint8_t *ring, int rs, int lane)
{
    const long s0 = static_cast<long>(u) * C::TILE_BYTES - C::FRONT;   // wave-uniform
    uint8_t *dst = ring + rs * C::TILE_BYTES;
    if (s0 >= 0 && s0 + C::NP * 1024L <= n_bytes) {
        const uint8_t *base = x + s0;   // scalar base, lane*16 + k*1024 offsets
#pragma unroll
        for (int k = 0; k < C::NP; k++)
            if (k < C::NPF || lane < C::REM_LANES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (lane * 16 + k * 1024)),
                                                 (__attribute__((address_space(3))) void *)(dst + k * 1024), 16, 0, 0);
    } else {
#pragma unroll
        for (int k = 0; k < C::NP; k++)
            if (k < C::NPF || lane < C::REM_LANES) {
                long off = s0 + k * 1024 + lane * 16;
                if (off > n_bytes - 16) off = n_bytes - 16;
                const uint8_t *src = off < 0 ? hist_end + off : x + off;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(dst + k * 1024), 16, 0, 0);
            }
    }
}

template <int T, int D, int MINB>
__global__ __launch_bounds__(256, MINB) void fe_mfma_kernel(
    const uint8_t *__restrict__ x, const uint8_t *__restrict__ hist_end, long n_bytes, const i4 *__restrict__ a_img,
    float scale_lo, const float2 *__restrict__ prev_in, float *__restrict__ demod, float *__restrict__ y_if,
    float2 *__restrict__ prev_out, long n_out, int n_tiles, int tiles_per_wave, uint8_t *__restrict__ hist_next,
    int hist_bytes)
{
    using C = MfCfg<T, D>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    uint8_t *ring = lds_raw + wave * C::RING;                 // this wave's private ring
    const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + wave);

    // the stream's last bytes become the next block's history (I_state/Q_state of the reference,
    // src/filter.cpp:182-187): one wave copies them; host guarantees n_bytes >= hist_bytes
    if (hist_next && wid == 0)
        for (int i = lane; i < hist_bytes; i += 64) hist_next[i] = x[n_bytes - hist_bytes + i];

    const int t0 = wid * tiles_per_wave;                      // this wave's tiles: [t0, t1)
    const int t1 = t0 + tiles_per_wave < n_tiles ? t0 + tiles_per_wave : n_tiles;
    if (t0 >= t1) return;
    // The discriminator needs the IF sample in front of the run.  A run that starts the block takes
    // it from the carried state (prev_i/prev_q, src/project.cpp:122-126); any other run first
    // multiplies the tile before its own ("dry": nothing stored).
    const int tb = t0 > 0 ? t0 - 1 : 0;

    // taps: KSTEPS x NDIG fragments, resident for the whole kernel
    i4 a[C::KSTEPS][C::NDIG];
#pragma unroll
    for (int j = 0; j < C::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) a[j][d] = a_img[(j * C::NDIG + d) * 64 + lane];
    float ci = 0.0f, cq = 0.0f;
    if (t0 == 0) {
        const float2 p = *prev_in;
        ci = p.x;
        cq = p.y;
    }
    // everything an ordinary load returns is in registers before the first DMA is issued (the
    // compiler drains vmcnt to 0 at the use of a plain load: keep that out of the streaming loop)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < C::KSTEPS; j++)
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) asm volatile("" : "+v"(a[j][d]));
    asm volatile("" : "+v"(ci), "+v"(cq));

    {
        int rs = 0;
        for (int u = tb; u <= tb + C::P && u <= t1; u++) {
            mf_dma_slot<C>(x, hist_end, n_bytes, u, ring, rs, lane);
            rs++;
        }
    }

    const int col = lane & 15, g = lane >> 4;
    const int lane_off = C::COL_BYTES * col + 16 * g;
    const float scale_hi = scale_lo * 65536.0f;
    const bool with_if = y_if != nullptr;
    const int src_lane = lane >= 16 ? lane - 16 : lane + 47;   // who holds the output in front of this lane's first
    int slot = 0;
    for (int t = tb; t < t1; t++) {
        // ---- tile t's slot and the first piece of slot t+1 have landed -------------------
        // vmcnt counts in issue order.  With every slot up to t+P issued, the operations younger
        // than (slot t+1, piece 0) are YOUNGER DMA pieces plus the stores of the P iterations in
        // between: exactly one demod store (+ one IF store) each once those were interior,
        // non-dry tiles.  Under-counting stores only waits longer; near the run's end wait for all.
        const bool steady = t + C::P <= t1;
        const bool warm = t - C::P >= t0;
        if (steady) {
            if (!warm) wait_vmcnt<C::YOUNGER>();
            else if (with_if) wait_vmcnt<C::YOUNGER + 2 * C::P>();
            else wait_vmcnt<C::YOUNGER + C::P>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- B fragments: 16 consecutive window bytes per lane and K-step ------------------
        const int base = slot * C::TILE_BYTES + lane_off;
        i4 b[C::KSTEPS];
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) {
            int adr = base + 64 * j;
            adr = adr >= C::RING ? adr - C::RING : adr;
            b[j] = *reinterpret_cast<const i4 *>(ring + adr);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): the slot may be refilled
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (t + C::P + 1 <= t1) mf_dma_slot<C>(x, hist_end, n_bytes, t + C::P + 1, ring, slot, lane);

        // ---- 3 exact int8 products per K-step ------------------------------------------------
        i4 acc[C::NDIG];
#pragma unroll
        for (int d = 0; d < C::NDIG; d++) acc[d] = (i4){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < C::KSTEPS; j++) {
            const i4 bs = b[j] ^ static_cast<int>(0x80808080u);   // u8 ^ 0x80 = (u8 - 128) as int8
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
        if (lane == 0) {
            pi = ci;
            pq = cq;
        }
        ci = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[2]), 63));
        cq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[3]), 63));
        const float d0 = demod_fast(v[0], v[1], pi, pq);
        const float d1 = demod_fast(v[2], v[3], v[0], v[1]);

        const long o = static_cast<long>(t) * C::TILE_OUT + C::COL_OUT * col + 2 * g;
        if (t >= t0 && o < n_out) {
            if (o + 1 < n_out) {
                *reinterpret_cast<f2 *>(demod + o) = (f2){d0, d1};
                if (with_if) *reinterpret_cast<f4 *>(y_if + 2 * o) = (f4){v[0], v[1], v[2], v[3]};
                if (prev_out && o + 2 == n_out) *prev_out = make_float2(v[2], v[3]);
            } else {
                demod[o] = d0;
                if (with_if) *reinterpret_cast<f2 *>(y_if + 2 * o) = (f2){v[0], v[1]};
                if (prev_out) *prev_out = make_float2(v[0], v[1]);
            }
        }
        slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
    }
}

template <int T, int D>
int launch_mfma(const FePlan &pl, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, const float *d_prev,
                float *d_demod, float *d_if, float *d_prev_out, uint8_t *d_hist_next, hipStream_t stream)
{
    using C = MfCfg<T, D>;
    constexpr int MINB = 2;                                    // workgroups per CU the register budget is set for
    if (C::FRONT > pl.hist_bytes) return fail(FMRX_EINVAL, "fe_mfma: history too short");
    const long n_out = static_cast<long>(n_samples / D);
    const long n_tiles = (n_out + C::TILE_OUT - 1) / C::TILE_OUT;
    long wgs_per_cu = (160 * 1024) / (4L * C::RING);
    if (wgs_per_cu > MINB) wgs_per_cu = MINB;
    if (const char *e = std::getenv("FMRX_FE_WGS_PER_CU")) {   // tuning knob
        const long v = std::atol(e);
        if (v >= 1 && v < wgs_per_cu) wgs_per_cu = v;
    }
    // a run shorter than ~8 tiles spends too much on its dry tile: fewer, longer runs on small blocks
    const long max_waves = 256 * wgs_per_cu * 4;
    long waves = (n_tiles + 7) / 8;
    if (waves > max_waves) waves = max_waves;
    if (waves < 1) waves = 1;
    const long tpw = (n_tiles + waves - 1) / waves;
    const long grid = ((n_tiles + tpw - 1) / tpw + 3) / 4;
    hipLaunchKernelGGL((fe_mfma_kernel<T, D, MINB>), dim3(static_cast<unsigned>(grid)), dim3(256), 4 * C::RING, stream, d_iq,
                       d_hist + pl.hist_bytes, static_cast<long>(2 * n_samples), reinterpret_cast<const i4 *>(pl.a_img.p),
                       pl.scale_lo, reinterpret_cast<const float2 *>(d_prev), d_demod, d_if,
                       reinterpret_cast<float2 *>(d_prev_out), n_out, static_cast<int>(n_tiles), static_cast<int>(tpw),
                       d_hist_next, pl.hist_bytes);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch fe_mfma_kernel<%d,%d>: %s", T, D, hipGetErrorString(e));
    return FMRX_OK;
}

// Digit image of the taps in A-operand order: [kstep][digit][lane][16 bytes].  Lane (row m = lane&15,
// quarter g = lane>>4) holds the coefficients that row applies to window bytes 64*kstep + 16*g + 0..15.
// Row m -> output r = 2*(m/4) + (m%4)/2 of the column, channel c = m%2: the C layout
// (row = 4*(lane>>4) + reg) then hands lane (col, g) the pair of outputs 2g, 2g+1 as (I,Q,I,Q).
template <int T, int D>
void build_image(const float *h, int s, std::vector<int8_t> &img)
{
    using C = MfCfg<T, D>;
    img.assign(static_cast<size_t>(C::KSTEPS) * C::NDIG * 64 * 16, 0);
    for (int j = 0; j < C::KSTEPS; j++)
        for (int lane = 0; lane < 64; lane++)
            for (int b = 0; b < 16; b++) {
                const int m = lane & 15, g = lane >> 4;
                const int r = 2 * (m / 4) + (m % 4) / 2, c = m % 2;
                const int p = 64 * j + 16 * g + b;                      // window byte
                const int e = C::FRONT + 2 * D * r + c - p;             // = 2k for the tap that meets it
                if (e < 0 || (e & 1) || e / 2 > T - 1) continue;
                long q = std::llround(std::ldexp(static_cast<double>(h[e / 2]), s));
                for (int d = 0; d < C::NDIG; d++) {
                    const long dig = ((q + 128) & 255) - 128;           // balanced digit in [-128, 127]
                    img[((static_cast<size_t>(j) * C::NDIG + d) * 64 + lane) * 16 + b] = static_cast<int8_t>(dig);
                    q = (q - dig) / 256;
                }
            }
}

#define FMRX_FE_MFMA_CASES(X) X(13, 10) X(101, 10) X(151, 10) X(13, 5) X(101, 5) X(151, 5) X(13, 3) X(101, 3) X(151, 3)

}  // namespace

int fe_mfma_plan_init(FePlan &pl, const float *h, int taps, int decim)
{
    pl.mfma = false;
    double maxabs = 0.0;
    for (int k = 0; k < taps; k++) {
        if (!std::isfinite(h[k])) return FMRX_OK;                       // generic / VALU kernels handle those
        maxabs = std::fmax(maxabs, std::fabs(static_cast<double>(h[k])));
    }
    if (maxabs == 0.0 || maxabs < 1e-30 || maxabs > 1e30) return FMRX_OK;
    // largest s with max|round(h*2^s)| <= 127*256^(NDIG-1): every balanced digit then fits int8
    const double limit = 127.0 * std::pow(256.0, kFeMfmaDigits - 1);
    int s = static_cast<int>(std::floor(std::log2(limit / maxabs)));
    while (std::ldexp(maxabs, s) > limit) s--;
    std::vector<int8_t> img;
#define X(T_, D_)                       \
    if (taps == T_ && decim == D_) {    \
        build_image<T_, D_>(h, s, img); \
        pl.mfma = true;                 \
    }
    FMRX_FE_MFMA_CASES(X)
#undef X
    if (!pl.mfma) return FMRX_OK;
    pl.scale_lo = static_cast<float>(std::ldexp(1.0, -s - 7));          // 2^-s for the taps, /128 for the samples
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
                   float *d_demod, float *d_if, float *d_prev_out, uint8_t *d_hist_next, hipStream_t stream)
{
    if (n_samples / pl.decim == 0) return FMRX_OK;
    if (!d_prev || !d_demod) return fail(FMRX_EINVAL, "fe_mfma_launch: null argument");
#define X(T_, D_) \
    if (pl.taps == T_ && pl.decim == D_) \
        return launch_mfma<T_, D_>(pl, d_iq, n_samples, d_hist, d_prev, d_demod, d_if, d_prev_out, d_hist_next, stream);
    FMRX_FE_MFMA_CASES(X)
#undef X
    return fail(FMRX_EINVAL, "fe_mfma_launch: no kernel for taps=%d decim=%d", pl.taps, pl.decim);
}

}  // namespace fmrx

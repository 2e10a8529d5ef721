// kernels_resample.hip -- polyphase rational resampler (expander + FIR +
// decimator collapsed), the audio stage of modes 2 and 3.
//
// Replaces convolveBlockResampleFIR (src/filter.cpp:191-223; called at
// src/project.cpp:353, 227, 264) in its stream form
//     y[k] = (1+U) * sum_j h[ph + j*U] * x[floor(k*D/U) - j],   ph = (k*D) mod U.
//
// The reference (and the generic kernel) walk h with stride U -- 588 B (mode 2)
// or 1764 B (mode 3) between consecutive taps of one output, which is what made
// this stage 6.6x slower than the plain audio FIR on the CPU (report Table 3).
// Here the taps are re-laid out once, polyphase-major: row ph holds
// h[ph], h[ph+U], h[ph+2U], ... contiguously (padded to 16 bytes), so a thread
// streams its row with 16-byte loads from L2 (the table is 59 KB / 178 KB) while
// the input window of the workgroup's 256 consecutive outputs (~7-9 KB) is staged
// once in LDS by coalesced loads.
//
// Arithmetic is the reference's, operation for operation: products and sums
// separately rounded, j ascending, then y += y*U -- the stage stays bit-exact.
//
// resample_lds_kernel goes one step further: 256 consecutive outputs hit every one of the U rows,
// so the kernel above pulls ~100 KB of taps per workgroup from L2 and runs at L2 bandwidth.  The
// whole table (59 KB) or half of it (mode 3: 2 x 92 KB) fits LDS: persistent workgroups of 1024
// threads load their slice of W taps x all phases once and stream tiles of 1024 outputs past it.
// A second pass continues every output's sum from where the first left it (the partial sum goes
// through y as a float, so the rounding sequence -- j ascending -- is unchanged: still bit-exact).
//
// resample_mfma_kernel (pipeline path of modes 2/3, not the bit-exact primitive): the pattern of phases repeats
// every U outputs = D inputs, so 16 consecutive outputs (rows) x 16 consecutive periods (columns) share one
// banded tap matrix: a 16x16 output tile is  A[16 x K] * X[K x 16]  on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32), K = the inputs the 16 rows touch (15*D/U + J, 186 / 212 for modes 2 / 3), A resident
// in registers, X = the periods' input windows staged in LDS, block of 16 periods after block (a workgroup walks several:
// taps loaded once, the next block fetched under the current block's products).  Per output ~190 MACs are issued
// instead of 101, but as 1/64 of a matrix instruction instead of a dependent chain with two LDS gathers per MAC.
// The sum is an fma chain over the window (newest sample first, like the reference's j ascending) instead of
// separately rounded products and sums: equal to float32 rounding (1e-7), not bit for bit.
#include "device_math.hpp"
#include "fmrx_internal.hpp"

#include <algorithm>

#pragma clang fp contract(off)

namespace fmrx {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kNT = 256;

__global__ __launch_bounds__(kNT) void resample_poly_kernel(const float *__restrict__ x, long n_in, long n_out,
                                                            const float *__restrict__ table, int J, int JP, int decim,
                                                            int upsamp, int span, float *__restrict__ y)
{
    extern __shared__ float xs[];
    const int t = threadIdx.x;
    const long k0 = static_cast<long>(blockIdx.x) * kNT;
    const long b0 = (k0 * decim) / upsamp;          // newest input of the first output
    const long lo = b0 - (J - 1);                   // oldest input any output of this tile touches
    for (int i = t; i < span; i += kNT) {
        const long g = lo + i;                      // negative -> carried history in front of the block
        xs[i] = g < n_in ? x[g] : 0.0f;
    }
    __syncthreads();
    const long k = k0 + t;
    if (k >= n_out) return;
    const long m = k * decim;
    const int ph = static_cast<int>(m % upsamp);
    const int b = static_cast<int>(m / upsamp - lo);   // index of x[floor(kD/U)] in xs
    const f4 *row = reinterpret_cast<const f4 *>(table + static_cast<long>(ph) * JP);
    float acc = 0.0f;
    for (int j4 = 0; j4 < JP / 4; j4++) {
        const f4 h = row[j4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = 4 * j4 + e;
            if (j < J) {
                const float prod = h[e] * xs[b - j];
                acc = acc + prod;
            }
        }
    }
    const float g = acc * static_cast<float>(upsamp);
    y[k] = acc + g;
}

constexpr int kLT = 1024;    // threads of the LDS-table kernel
constexpr int kLR = 1;       // outputs per thread and tile (2 was measured: the kernel is bound by LDS bandwidth and bank conflicts, not by the chain latency: no faster)
constexpr int kRowPad = 4;   // floats: rows then start on all 16 bank groups, not 8

__global__ __launch_bounds__(kLT) void resample_lds_kernel(const float *__restrict__ x, long n_in, long n_out,
                                                            const float *__restrict__ table, int J, int JP, int decim,
                                                            int upsamp, int span, int j0, int W, int first, int last,
                                                            long n_tiles, float *__restrict__ y)
{
    extern __shared__ float lds[];
    const int WP = W + kRowPad;
    float *tab = lds;                       // [upsamp][WP]: taps j0 .. j0+W-1 of every phase
    float *xs = lds + upsamp * WP;          // [span]
    const int t = threadIdx.x;
    const int w4 = W / 4;
    constexpr int TILE = kLT * kLR;         // outputs per tile
    for (int i = t; i < upsamp * w4; i += kLT) {
        const int row = i / w4, c = i - row * w4;
        f4 v = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        if (j0 + 4 * c < JP) v = *reinterpret_cast<const f4 *>(table + static_cast<long>(row) * JP + j0 + 4 * c);
        *reinterpret_cast<f4 *>(tab + row * WP + 4 * c) = v;
    }
    // the next tile's inputs are fetched into registers while this tile is multiplied (one workgroup
    // per CU: nothing else would hide the fetch)
    constexpr int kNL = 16;                             // >= span / kLT (host checks)
    float xn[kNL];
    auto fetch = [&](long tile) {
        const long lo = (tile * TILE * decim) / upsamp - (J - 1);   // oldest input any output of the tile touches
#pragma unroll
        for (int q = 0; q < kNL; q++) {
            const int i = t + q * kLT;
            const long g = lo + i;                      // negative -> carried history in front of the block
            xn[q] = (i < span && g < n_in) ? x[g] : 0.0f;
        }
    };
    if (static_cast<long>(blockIdx.x) < n_tiles) fetch(blockIdx.x);
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long k0 = tile * TILE;
        const long lo = (k0 * decim) / upsamp - (J - 1);
        __syncthreads();                                // the previous tile's reads of xs (and the table fill)
#pragma unroll
        for (int q = 0; q < kNL; q++)
            if (t + q * kLT < span) xs[t + q * kLT] = xn[q];
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
        // kLR outputs per thread, kLT apart (coalesced stores): every output keeps the reference's own chain -- products
        // and sums separately rounded, j ascending -- so the result is bit-exact; the chains of a thread are independent
        // and fill each other's latency
        const f4 *row[kLR];
        const float *xp[kLR];
        float acc[kLR];
        long kk[kLR];
#pragma unroll
        for (int r = 0; r < kLR; r++) {
            const long k = k0 + t + static_cast<long>(r) * kLT;
            kk[r] = k < n_out ? k : -1;
            const long m = (k < n_out ? k : k0) * decim;
            const int ph = static_cast<int>(m % upsamp);
            const int b = static_cast<int>(m / upsamp - lo) - j0;   // index in xs of the sample tap j0 meets
            row[r] = reinterpret_cast<const f4 *>(tab + ph * WP);
            xp[r] = xs + b;
            acc[r] = (first || kk[r] < 0) ? 0.0f : y[kk[r]];
        }
        const int nfull = (J - j0) / 4 < w4 ? (J - j0) / 4 : w4;   // groups of 4 taps that exist entirely
        for (int j4 = 0; j4 < nfull; j4++) {
            f4 h[kLR];
#pragma unroll
            for (int r = 0; r < kLR; r++) h[r] = row[r][j4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int r = 0; r < kLR; r++) {
                    const float prod = h[r][e] * xp[r][-(4 * j4 + e)];
                    acc[r] = acc[r] + prod;
                }
            }
        }
        if (nfull < w4) {                                          // the group the filter ends in
#pragma unroll
            for (int r = 0; r < kLR; r++) {
                const f4 h = row[r][nfull];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (j0 + 4 * nfull + e < J) {
                        const float prod = h[e] * xp[r][-(4 * nfull + e)];
                        acc[r] = acc[r] + prod;
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < kLR; r++) {
            if (kk[r] < 0) continue;
            if (last) {
                const float g = acc[r] * static_cast<float>(upsamp);
                y[kk[r]] = acc[r] + g;
            } else {
                y[kk[r]] = acc[r];
            }
        }
    }
}


// ---- matrix-core resampler -----------------------------------------------------------------------------------
constexpr int kRsTiles = 4;                    // output tiles per workgroup: one per wave
constexpr int kRsPieces = 160;                 // 16-byte pieces per staged row at most (16 rows x 161 x 16 B = 41 KB)

// workgroup (chain, grp): output tiles [m0, m1) of group grp (wave w = tile m0 + w, its taps resident in registers) x the period
// blocks pb = 8 chain + xcd, + 8 n_chain, ...: 16 periods each.  groups[4 grp ..] = m0, m1, lo, pieces: the 16 LDS rows hold
// inputs [lo, lo + 4 pieces) of each period, row stride pieces | 1 (odd: rows then start on all 16 bank groups), and one more
// piece per row behind them takes what the staging threads read past a row's end.  Tile m: K index w <-> input offset
// tile_top[m] - w inside the period.  The next block's inputs are fetched into registers before the current block's matrix
// products and go to LDS behind them: one buffer, two barriers per block.
// ELEM = false: x is 16-byte aligned and the caller vouches for finite, readable samples kRsFront in front of the history and
// kRsBack behind the block (the pipeline's discriminator buffer has them): only taps that are zero ever meet those, so every
// block is staged the same way, with one address register and instruction offsets.
// ELEM = true: any x: element by element, outside [g_min, n_in) = a word that holds 0.0f (the tap image's first: K index 0 of
// tile 0 lies past its band).
template <int KS4, int NL, bool ELEM>
__global__ __launch_bounds__(256, ELEM ? 2 : KS4 >= 14 ? 3 : 4) void resample_mfma_kernel(const float *__restrict__ x, long n_in, long g_min, long n_per,
                                                             const float *__restrict__ a_tab, const int *__restrict__ tile_top,
                                                             const int *__restrict__ groups, int n_groups, int n_chain, int decim,
                                                             int upsamp, float *__restrict__ y, int16_t *__restrict__ pcm, int wrap)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    // workgroup id -> (chain, group): workgroups go round the 8 XCDs by id, so id % 8 is kept the same for all groups of a
    // period block and they are adjacent in dispatch order: the overlap of their windows is served by one L2
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int g_idx = slot % n_groups;
    const long npb = (n_per + 15) >> 4;
    const long pb_step = 8L * n_chain;
    long pb = static_cast<long>(slot / n_groups) * 8 + xcd;
    if (pb >= npb) return;
    const int *grp = groups + 4 * g_idx;
    const int m0 = grp[0], m1 = grp[1], lo = grp[2], pieces = grp[3];
    const int stride = 4 * (pieces | 1);
    const int m = m0 + wave;
    const bool have = m < m1;                                  // wave-uniform
    // this wave's taps first: their L2 round trip runs under the staging of the rows
    f4 a[KS4];
    {
        const f4 *ap = reinterpret_cast<const f4 *>(a_tab) + (static_cast<long>(have ? m : m0) * 64 + lane) * KS4;
#pragma unroll
        for (int jj = 0; jj < KS4; jj++) a[jj] = ap[jj];
    }
    const int top = tile_top[have ? m : m0];
    // staging: 16 threads per row (period), thread c the pieces c, c + 16, ... (NL of them)
    const int row = t >> 4, c = t & 15;
    f4 v[NL];
    auto fetch = [&](long b) {
        const long q0 = b * 16;
        if constexpr (!ELEM) {
            // rows past the last period repeat it (never stored)
            const int last_row = static_cast<int>(min(15L, n_per - 1 - q0));
            const unsigned voff = 4u * (static_cast<unsigned>(min(row, last_row)) * decim + 4 * c);   // bytes from the block's first piece
            const char *base = reinterpret_cast<const char *>(x + (q0 * decim + lo));                 // uniform
#pragma unroll
            for (int u = 0; u < NL; u++) v[u] = *reinterpret_cast<const f4 *>(base + (voff + 256u * u));
        } else {
            long q = q0 + row;
            q = q < n_per ? q : n_per - 1;
            const long gr = q * decim + lo;
#pragma unroll
            for (int u = 0; u < NL; u++) {
                const int pc = 4 * min(c + 16 * u, pieces - 1);          // past the row's end: its last piece again
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const long g = gr + pc + e;
                    v[u][e] = *((g >= g_min && g < n_in) ? x + g : a_tab);
                }
            }
        }
    };
    auto put = [&]() {
        int cz = c;
        asm volatile("" : "+v"(cz));                           // the LDS addresses are recomputed here, not kept across the loop
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int pc = cz + 16 * u;
            const int sp = pieces | 1;                         // row stride in pieces
            const int at = ELEM ? row * sp + min(pc, pieces - 1) : pc < pieces ? row * sp + pc : 16 * sp + row;
            reinterpret_cast<f4 *>(rows)[at] = v[u];
        }
    };
    fetch(pb);
    put();
#pragma unroll
    for (int jj = 0; jj < KS4; jj++) asm volatile("" : "+v"(a[jj]));   // the taps stay in registers (not re-loaded at their use)
    __syncthreads();
    const int n = lane & 15, kq = lane >> 4;
    // this lane's operands of K-steps 4jj..4jj+3: inputs top - 16jj - 4kq - {0,1,2,3} of period n
    const float *bp = rows + n * stride + (top - 3 - lo) - 4 * kq;
    const int r = 16 * m + 4 * kq;
    const bool whole = 16 * m + 16 <= upsamp;                  // wave-uniform: every row of the tile is an output
    const float fu = static_cast<float>(upsamp);
    for (;;) {
        const long nxt = pb + pb_step;
        const bool more = nxt < npb;                           // uniform over the workgroup
        if (more) fetch(nxt);
        if (have) {
            f4 acc = (f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int jj = 0; jj < KS4; jj++) {
                const f4 xv = *reinterpret_cast<const f4 *>(bp - 16 * jj);
#pragma unroll
                for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj][e], xv[3 - e], acc, 0, 0, 0);
            }
            const long q = pb * 16 + n;
            f4 out;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float gq = acc[e] * fu;                  // y += y*U (src/filter.cpp:221), separately rounded
                out[e] = acc[e] + gq;
            }
            if (q < n_per) {
                const long o = q * upsamp + r;
                if (whole && !y) {                             // the pipeline's case: s16 only (src/threadMonoOnly.cpp:185-191)
                    struct __attribute__((packed)) S4 { int16_t s[4]; } pk;
#pragma unroll
                    for (int e = 0; e < 4; e++) pk.s[e] = pcm_pack_flat(out[e], wrap);
                    __builtin_memcpy(pcm + o, &pk, sizeof pk);   // one 8-byte store, 2-byte aligned
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (r + e < upsamp) {
                            if (y) y[o + e] = out[e];
                            if (pcm) pcm[o + e] = pcm_pack(out[e], wrap);
                        }
                }
            }
        }
        if (!more) break;
        __syncthreads();                                       // every wave has read this block's rows
        put();
        __syncthreads();
        pb = nxt;
    }
}

}  // namespace

// host side of the matrix-core resampler: tile geometry, tap image, tile groups
static int resample_mfma_plan(ResamplePlan &pl, const float *h)
{
    pl.mfma = false;
    const int U = pl.upsamp, D = pl.decim, J = pl.J;
    if (D % 4 != 0 || U < 16) return FMRX_OK;
    const int ntiles = (U + 15) / 16;
    std::vector<int> top(ntiles), b0(ntiles);
    int K = 0;
    for (int m = 0; m < ntiles; m++) {
        const int r_last = std::min(16 * m + 15, U - 1);
        const int bmax = static_cast<int>(static_cast<long>(r_last) * D / U);
        b0[m] = static_cast<int>(static_cast<long>(16 * m) * D / U);
        top[m] = (bmax + 1 + 3) / 4 * 4 - 1;
        K = std::max(K, top[m] - b0[m] + J);
    }
    const int KS4 = std::max(8, (K + 31) / 32 * 2);             // K-steps in sixteens: an even count from 8 to 16 (kernel instances)
    if (KS4 > 16) return FMRX_OK;
    // tile groups of kRsTiles consecutive tiles (one per wave): a period's staged window = what the group's tiles read
    std::vector<int> grp;
    int max_pieces = 0;
    for (int m = 0; m < ntiles; m += kRsTiles) {
        const int m1 = std::min(m + kRsTiles, ntiles);
        const int lo = top[m] - 16 * KS4 + 1;                      // oldest input tile m reads; top % 4 == 3 -> a multiple of 4
        const int pieces = (top[m1 - 1] - lo + 1) / 4;
        if (pieces > kRsPieces) return FMRX_OK;
        max_pieces = std::max(max_pieces, pieces);
        grp.insert(grp.end(), {m, m1, lo, pieces});
    }
    // tap image [tile][lane][K-step]: lane (row i = lane & 15, kq = lane >> 4), K-step ks <-> K index w = 16 (ks/4) + 4 kq + ks%4
    std::vector<float> img(static_cast<size_t>(ntiles) * 64 * 4 * KS4, 0.0f);
    for (int m = 0; m < ntiles; m++)
        for (int lane = 0; lane < 64; lane++) {
            const int i = lane & 15, kq = lane >> 4, r = 16 * m + i;
            if (r >= U) continue;
            const long rd = static_cast<long>(r) * D;
            const int ph = static_cast<int>(rd % U), bi = static_cast<int>(rd / U);
            for (int ks = 0; ks < 4 * KS4; ks++) {
                const int w = 16 * (ks / 4) + 4 * kq + ks % 4;
                const int j = bi - (top[m] - w);
                if (j >= 0 && j < J && ph + static_cast<long>(j) * U < pl.taps)
                    img[(static_cast<size_t>(m) * 64 + lane) * 4 * KS4 + ks] = h[ph + j * U];
            }
        }
    FMRX_TRY(pl.mfma_img.alloc(img.size()));
    FMRX_HIP(hipMemcpy(pl.mfma_img.p, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
    FMRX_TRY(pl.mfma_top.alloc(top.size()));
    FMRX_HIP(hipMemcpy(pl.mfma_top.p, top.data(), top.size() * sizeof(int), hipMemcpyHostToDevice));
    FMRX_TRY(pl.mfma_groups.alloc(grp.size()));
    FMRX_HIP(hipMemcpy(pl.mfma_groups.p, grp.data(), grp.size() * sizeof(int), hipMemcpyHostToDevice));
    pl.mfma_ks4 = KS4;
    pl.mfma_ngroups = static_cast<int>(grp.size() / 4);
    pl.mfma_pieces = max_pieces;
    // what piece staging reads: from the first period's lowest window start to the last period's highest one + 16 NL pieces
    {
        const int nl = std::max(4, (max_pieces + 15) / 16);
        int lo_min = 0, lo_max = 0;
        for (size_t g = 0; g < grp.size(); g += 4) {
            lo_min = std::min(lo_min, grp[g + 2]);
            lo_max = std::max(lo_max, grp[g + 2]);
        }
        pl.mfma_reach_ok = -lo_min <= kResampleFront && lo_max + 64 * nl - D <= kResampleBack;
    }
    pl.mfma = true;
    return FMRX_OK;
}

template <int KS4, int NL, bool ELEM>
static int resample_mfma_launch_ks(const ResamplePlan &pl, const float *x, size_t n_in, float *d_y, int16_t *d_pcm, int wrap,
                                   hipStream_t stream, int chains)
{
    const long n_per = static_cast<long>(n_in / pl.decim);
    const long npb8 = (n_per + 127) / 128;                               // period blocks per XCD
    const size_t lds = (static_cast<size_t>(16) * (pl.mfma_pieces | 1) + 16) * 16;
    int &wgs = pl.mfma_wgs_per_cu[ELEM ? 1 : 0];
    if (wgs == 0) {
        FMRX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs, resample_mfma_kernel<KS4, NL, ELEM>, 256, lds));
        if (wgs < 1) wgs = 1;
    }
    // chains per XCD: as many workgroups as are resident at once (32 CUs per XCD), every chain the same number of blocks (+-1)
    long cmax = std::max<long>(1, 32L * wgs / pl.mfma_ngroups);
    if (chains > 0) cmax = chains;                                      // option resample_chains: A/B only
    const long iters = (npb8 + cmax - 1) / cmax;
    const long n_chain = (npb8 + iters - 1) / iters;
    hipLaunchKernelGGL((resample_mfma_kernel<KS4, NL, ELEM>), dim3(static_cast<unsigned>(n_chain * 8 * pl.mfma_ngroups)), dim3(256),
                       lds, stream, x, static_cast<long>(n_in), -static_cast<long>(pl.J - 1), n_per, pl.mfma_img.p,
                       pl.mfma_top.p, pl.mfma_groups.p, pl.mfma_ngroups, static_cast<int>(n_chain), pl.decim, pl.upsamp, d_y,
                       d_pcm, wrap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch resample_mfma_kernel: %s", hipGetErrorString(e));
    return FMRX_OK;
}

// loads per staging thread: an instance per count from 4 to 10 (piece staging), per even count (element staging)
template <int KS4>
static int resample_mfma_launch_nl(const ResamplePlan &pl, const float *x, size_t n_in, float *d_y, int16_t *d_pcm, int wrap,
                                   hipStream_t stream, bool pieces16, int chains)
{
    const int nl = std::max(4, (pl.mfma_pieces + 15) / 16);
    if (!pieces16) {
        switch ((nl + 1) / 2 * 2) {
#define X(N_) case N_: return resample_mfma_launch_ks<KS4, N_, true>(pl, x, n_in, d_y, d_pcm, wrap, stream, chains);
            X(4) X(6) X(8) X(10)
#undef X
        }
    } else {
        switch (nl) {
#define X(N_) case N_: return resample_mfma_launch_ks<KS4, N_, false>(pl, x, n_in, d_y, d_pcm, wrap, stream, chains);
            X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#undef X
        }
    }
    return fail(FMRX_EINVAL, "resample_mfma_launch: no kernel for %d pieces per row", pl.mfma_pieces);
}

static int resample_mfma_launch(const ResamplePlan &pl, const float *x, size_t n_in, float *d_y, int16_t *d_pcm, int wrap,
                                hipStream_t stream, bool margins, int chains)
{
    // 16-byte pieces, every block staged alike: x aligned (it is, unless the caller's all-pass delay shifted it by a count that
    // is not a multiple of 4) and samples to read on either side of the block
    const bool pieces16 = margins && pl.mfma_reach_ok && reinterpret_cast<uintptr_t>(x) % 16 == 0;
    switch (pl.mfma_ks4) {
#define X(K_) case K_: return resample_mfma_launch_nl<K_>(pl, x, n_in, d_y, d_pcm, wrap, stream, pieces16, chains);
        X(8) X(10) X(12) X(14) X(16)
#undef X
    }
    return fail(FMRX_EINVAL, "resample_mfma_launch: no kernel for %d K-steps", 4 * pl.mfma_ks4);
}

// whole periods, enough of them to fill the chip: the matrix-core kernel may run
bool resample_mfma_available(const ResamplePlan &pl, const float *d_x, size_t n_in, int delay, const Options &o)
{
    (void)d_x;
    (void)delay;   // any alignment: the staging falls back to 4-byte loads
    return pl.fast && pl.mfma && !o.resample_exact && n_in % pl.decim == 0 && n_in / pl.decim >= 64;
}

int resample_plan_init(ResamplePlan &pl, const float *h, int taps, int decim, int upsamp)
{
    pl.taps = taps;
    pl.decim = decim;
    pl.upsamp = upsamp;
    pl.J = (taps + upsamp - 1) / upsamp;
    pl.JP = (pl.J + 3) / 4 * 4;
    std::vector<float> tab(static_cast<size_t>(upsamp) * pl.JP, 0.0f);
    for (int ph = 0; ph < upsamp; ph++)
        for (int j = 0; ph + j * upsamp < taps; j++) tab[static_cast<size_t>(ph) * pl.JP + j] = h[ph + j * upsamp];
    FMRX_TRY(pl.table.alloc(tab.size()));
    FMRX_HIP(hipMemcpy(pl.table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    FMRX_TRY(pl.h.alloc(taps));
    FMRX_HIP(hipMemcpy(pl.h.p, h, taps * sizeof(float), hipMemcpyHostToDevice));
    // inputs spanned by 256 consecutive outputs, plus the J-1 older ones
    pl.span = static_cast<int>((static_cast<long>(kNT) * decim + upsamp - 1) / upsamp) + pl.J + 1;
    pl.fast = pl.span * sizeof(float) <= 60 * 1024;
    // LDS-table kernel: as few passes as make (all phases x W taps) + one tile's inputs fit 144 KB of LDS
    pl.span_l = static_cast<int>((static_cast<long>(kLT) * kLR * decim + upsamp - 1) / upsamp) + pl.J + 1;
    pl.npass = 0;
    for (int np = 1; np <= 4; np++) {
        const int W = (pl.JP / 4 + np - 1) / np * 4;
        if (pl.span_l <= 16 * kLT &&   // the kernel prefetches a tile's inputs in 16 registers per thread
            (static_cast<long>(upsamp) * (W + kRowPad) + pl.span_l) * sizeof(float) <= 160 * 1024) {
            pl.npass = np;
            pl.W = W;
            break;
        }
    }
    return resample_mfma_plan(pl, h);
}

// x points at the block start; x[-(J-1+delay) .. -1] must be readable history
int resample_launch(const ResamplePlan &pl, const float *d_x, size_t n_in, int delay, float *d_y, const Options &o,
                    hipStream_t stream, bool force_generic, bool exact, int16_t *d_pcm, int wrap, bool margins)
{
    const size_t n_out = (n_in * static_cast<size_t>(pl.upsamp)) / pl.decim;
    if (n_out == 0) return FMRX_OK;
    // the matrix-core kernel (float32-rounding-equal, not bit-exact: never for the primitive); it packs the PCM itself
    if (!force_generic && !exact && resample_mfma_available(pl, d_x, n_in, delay, o))
        return resample_mfma_launch(pl, d_x - delay, n_in, d_y, d_pcm, wrap, stream, margins, o.resample_chains);
    if (!d_y) return fail(FMRX_EINVAL, "resample_launch: this path needs the f32 output buffer");
    if (d_pcm) {   // every other kernel writes f32 only: pack behind it
        FMRX_TRY(resample_launch(pl, d_x, n_in, delay, d_y, o, stream, force_generic, exact, nullptr, 0, margins));
        return k_pcm16(d_y, n_out, d_pcm, wrap, stream);
    }
    if (!pl.fast || force_generic)
        return k_resample_generic(d_x - delay, n_in, pl.h.p, pl.taps, pl.decim, pl.upsamp, d_y, stream);
    if (pl.npass > 0 && n_out >= 64 * kLT && !o.resample_l2) {
        const size_t lds_bytes = (static_cast<size_t>(pl.upsamp) * (pl.W + kRowPad) + pl.span_l) * sizeof(float);
        if (lds_bytes > 64 * 1024) {   // more dynamic LDS than the default cap: opt in, once per device
            static bool raised[64] = {};
            int dev = 0;
            FMRX_HIP(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !raised[dev]) {
                FMRX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&resample_lds_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                if (dev >= 0 && dev < 64) raised[dev] = true;
            }
        }
        const long n_tiles = static_cast<long>((n_out + kLT * kLR - 1) / (kLT * kLR));
        const unsigned grid = static_cast<unsigned>(n_tiles < 256 ? n_tiles : 256);
        for (int pass = 0; pass < pl.npass; pass++) {
            hipLaunchKernelGGL(resample_lds_kernel, dim3(grid), dim3(kLT), lds_bytes, stream, d_x - delay,
                               static_cast<long>(n_in), static_cast<long>(n_out), pl.table.p, pl.J, pl.JP, pl.decim,
                               pl.upsamp, pl.span_l, pass * pl.W, pl.W, pass == 0, pass == pl.npass - 1, n_tiles, d_y);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(FMRX_EHIP, "launch resample_lds_kernel: %s", hipGetErrorString(e));
        }
        return FMRX_OK;
    }
    const unsigned grid = static_cast<unsigned>((n_out + kNT - 1) / kNT);
    hipLaunchKernelGGL(resample_poly_kernel, dim3(grid), dim3(kNT), pl.span * sizeof(float), stream, d_x - delay,
                       static_cast<long>(n_in), static_cast<long>(n_out), pl.table.p, pl.J, pl.JP, pl.decim, pl.upsamp,
                       pl.span, d_y);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch resample_poly_kernel: %s", hipGetErrorString(e));
    return FMRX_OK;
}

}  // namespace fmrx

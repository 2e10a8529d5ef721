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
#include "fmrx_internal.hpp"

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

}  // namespace

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
    return FMRX_OK;
}

// x points at the block start; x[-(J-1+delay) .. -1] must be readable history
int resample_launch(const ResamplePlan &pl, const float *d_x, size_t n_in, int delay, float *d_y, const Options &o,
                    hipStream_t stream, bool force_generic)
{
    const size_t n_out = (n_in * static_cast<size_t>(pl.upsamp)) / pl.decim;
    if (n_out == 0) return FMRX_OK;
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

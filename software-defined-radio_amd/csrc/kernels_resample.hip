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
    return FMRX_OK;
}

// x points at the block start; x[-(J-1+delay) .. -1] must be readable history
int resample_launch(const ResamplePlan &pl, const float *d_x, size_t n_in, int delay, float *d_y, hipStream_t stream,
                    bool force_generic)
{
    const size_t n_out = (n_in * static_cast<size_t>(pl.upsamp)) / pl.decim;
    if (n_out == 0) return FMRX_OK;
    if (!pl.fast || force_generic)
        return k_resample_generic(d_x - delay, n_in, pl.h.p, pl.taps, pl.decim, pl.upsamp, d_y, stream);
    const unsigned grid = static_cast<unsigned>((n_out + kNT - 1) / kNT);
    hipLaunchKernelGGL(resample_poly_kernel, dim3(grid), dim3(kNT), pl.span * sizeof(float), stream, d_x - delay,
                       static_cast<long>(n_in), static_cast<long>(n_out), pl.table.p, pl.J, pl.JP, pl.decim, pl.upsamp,
                       pl.span, d_y);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch resample_poly_kernel: %s", hipGetErrorString(e));
    return FMRX_OK;
}

}  // namespace fmrx

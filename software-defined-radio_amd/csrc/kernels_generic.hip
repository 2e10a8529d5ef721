// kernels_generic.hip -- parameter-generic HIP kernels for every primitive of
// the path (any tap count, any decimation, any alignment).
//
// These are the kernels behind the stage-level C ABI (one call per reference
// primitive) and the fall-back of the pipeline when no specialised kernel
// exists for a (taps, decim) pair.  They keep the reference's float32
// evaluation ORDER (separate multiply and add, taps ascending, no FMA
// contraction), so given identical inputs they reproduce the reference's
// results bit for bit -- except fmPLL, whose sinf/cosf/atan2f come from the
// device math library.  The throughput path is kernels_fe.hip /
// kernels_audio.hip.
//
// Reference semantics: src/filter.cpp (line ranges cited per kernel).
#include "device_math.hpp"
#include "fmrx_internal.hpp"

// No implicit a*b+c -> fma anywhere in this file: the reference is built for
// baseline x86-64 (no FMA), SURVEY A.2.
#pragma clang fp contract(off)

namespace fmrx {

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(size_t n, int per_block = kBlock)
{
    size_t g = (n + per_block - 1) / per_block;
    return static_cast<unsigned>(g ? g : 1);
}

#define FMRX_LAUNCH_CHECK(name)                                                                   \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) return fail(FMRX_EHIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- FIR, decimating, float in (src/filter.cpp:133-154, 158-188) ------------
// One output per thread; x[-(taps-1)] .. x[-1] is the carried history.
__global__ void fir_generic_kernel(const float *__restrict__ x, size_t n_out, const float *__restrict__ h, int taps,
                                   int decim, float *__restrict__ y)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    const float *xs = x + static_cast<ptrdiff_t>(k) * decim;
    float acc = 0.0f;
    for (int n = 0; n < taps; n++) {
        const float prod = h[n] * xs[-n];
        acc = acc + prod;
    }
    y[k] = acc;
}

// ---- front end on raw u8 I/Q (src/iofunc.cpp:133, project.cpp:98-121) ---------
__device__ inline float u8_norm(uint8_t u)
{
    // (u-128)/128.0 in double then float: exact, so a float divide by 128 is identical
    return static_cast<float>(static_cast<int>(u) - 128) * 0.0078125f;
}

__global__ void fe_generic_kernel(const uint8_t *__restrict__ iq, const uint8_t *__restrict__ hist, int hist_bytes,
                                  size_t n_out, const float *__restrict__ h, int taps, int decim,
                                  float2 *__restrict__ y)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    const ptrdiff_t s0 = static_cast<ptrdiff_t>(k) * decim;  // complex-sample index of tap 0
    float ai = 0.0f, aq = 0.0f;
    for (int n = 0; n < taps; n++) {
        const ptrdiff_t s = s0 - n;
        uint8_t ui, uq;
        if (s >= 0) {
            ui = iq[2 * s];
            uq = iq[2 * s + 1];
        } else if (hist) {
            ui = hist[hist_bytes + 2 * s];
            uq = hist[hist_bytes + 2 * s + 1];
        } else {
            ui = uq = 128;
        }
        const float pi = h[n] * u8_norm(ui);
        const float pq = h[n] * u8_norm(uq);
        ai = ai + pi;
        aq = aq + pq;
    }
    y[k] = make_float2(ai, aq);
}

// ---- polyphase rational resampler, stream form (src/filter.cpp:191-223) ---------
// y[k] = (1+U) * sum_j h[ph + j*U] * x[(k*D - ph)/U - j],  ph = (k*D) % U
__global__ void resample_generic_kernel(const float *__restrict__ x, size_t n_out, const float *__restrict__ h,
                                        int taps, int decim, int upsamp, float *__restrict__ y)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    const long long m = static_cast<long long>(k) * decim;
    const int ph = static_cast<int>(m % upsamp);
    const float *xs = x + (m - ph) / upsamp;
    float acc = 0.0f;
    int j = 0;
    for (int n = ph; n < taps; n += upsamp, j++) {
        const float prod = h[n] * xs[-j];
        acc = acc + prod;
    }
    const float g = acc * static_cast<float>(upsamp);
    y[k] = acc + g;
}

// ---- FM discriminator (src/filter.cpp:248-266) ------------------------------------
__device__ inline float demod_one(float i, float q, float pi, float pq) { return demod_exact(i, q, pi, pq); }

template <bool FAST>
__global__ void demod_if_kernel(const float2 *__restrict__ z, size_t n, const float2 *__restrict__ prev,
                                float2 *__restrict__ prev_out, float *__restrict__ out)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float2 c = z[k];
    const float2 p = k ? z[k - 1] : *prev;
    out[k] = FAST ? demod_fast(c.x, c.y, p.x, p.y) : demod_exact(c.x, c.y, p.x, p.y);
    if (prev_out && k == n - 1) *prev_out = c;
}

__global__ void demod_planar_kernel(const float *__restrict__ I, const float *__restrict__ Q, size_t n, float prev_i,
                                    float prev_q, float *__restrict__ out)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float pi = k ? I[k - 1] : prev_i, pq = k ? Q[k - 1] : prev_q;
    out[k] = demod_one(I[k], Q[k], pi, pq);
}

// ---- arctangent demodulator of the reference's Python model (model/fmSupportLib.py:502-531), float64 --------------
// The model walks the samples: phase = atan2(Q, I); [prev, phase] = np.unwrap([prev, phase]); out = phase - prev; prev = phase.
// np.unwrap of a pair adds  ddmod - dd  to the second element,  dd = phase - prev,  ddmod = mod(dd + pi, 2 pi) - pi  (pi
// instead of -pi when dd > 0), nothing where |dd| < pi: out[k] is the phase step wrapped into (-pi, pi].  The model's running
// phase is the UNWRAPPED one (it grows with the stream); a step computed from the wrapped phases of the two samples is the
// same real number, and differs from the model's float64 result by the rounding of that growing phase (1e-12 after 1e4 rad):
// every output is independent, one thread each.  (C++ reference: the discriminator fmDemod; this variant is model-only.)
__device__ inline double unwrap_step(double dd)
{
    const double pi = 3.141592653589793, two_pi = 6.283185307179586;
    double r = fmod(dd + pi, two_pi);                 // np.mod: the result has the sign of the divisor
    if (r < 0.0) r += two_pi;
    double ddmod = r - pi;
    if (ddmod == -pi && dd > 0.0) ddmod = pi;
    return fabs(dd) < pi ? dd : ddmod;
}

__global__ void demod_arctan_planar_kernel(const double *__restrict__ I, const double *__restrict__ Q, size_t n, double prev_phase,
                                           double *__restrict__ out)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double ph = atan2(Q[k], I[k]);
    const double pv = k ? atan2(Q[k - 1], I[k - 1]) : prev_phase;
    out[k] = unwrap_step(ph - pv);
}

__global__ void demod_arctan_if_kernel(const float2 *__restrict__ z, size_t n, const float2 *__restrict__ prev, float *__restrict__ out)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float2 c = z[k];
    const float2 p = k ? z[k - 1] : *prev;
    const double ph = atan2(static_cast<double>(c.y), static_cast<double>(c.x));
    const double pv = atan2(static_cast<double>(p.y), static_cast<double>(p.x));
    out[k] = static_cast<float>(unwrap_step(ph - pv));
}

// ---- element-wise helpers ------------------------------------------------------------
__global__ void u8_to_f32_kernel(const uint8_t *__restrict__ raw, size_t n, float *__restrict__ out)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) out[k] = u8_norm(raw[k]);
}

__global__ void deinterleave_kernel(const float2 *__restrict__ iq, size_t n, float *__restrict__ I, float *__restrict__ Q)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) {
        const float2 v = iq[k];
        I[k] = v.x;
        Q[k] = v.y;
    }
}

// src/threadMonoOnly.cpp:185-191
__device__ inline int16_t pcm_one(float a, int wrap) { return pcm_pack(a, wrap); }

__global__ void pcm16_kernel(const float *__restrict__ a, size_t n, int16_t *__restrict__ out, int wrap)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) out[k] = pcm_one(a[k], wrap);
}

// interleaved L,R: the layout of the writer at src/project.cpp:292-302
__global__ void pcm16_stereo_kernel(const float *__restrict__ l, const float *__restrict__ r, size_t n,
                                    int16_t *__restrict__ out, int wrap)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) {
        out[2 * k] = pcm_one(l[k], wrap);
        out[2 * k + 1] = pcm_one(r[k], wrap);
    }
}

// src/filter.cpp:14-29: out = [state, in[0 .. n-ns)]
__global__ void all_pass_kernel(const float *__restrict__ in, size_t n, const float *__restrict__ state, size_t ns,
                                float *__restrict__ out)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) out[k] = k < ns ? state[k] : in[k - ns];
}

// src/project.cpp:246-248: mixer = bpf * pll * 2
__global__ void mix_kernel(const float *__restrict__ bpf, const float *__restrict__ pll, size_t n, float *__restrict__ out)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) {
        const float t = bpf[k] * pll[k];
        out[k] = t * 2;
    }
}

// src/project.cpp:277-280
__global__ void combine_kernel(const float *__restrict__ st, const float *__restrict__ mono, size_t n,
                               float *__restrict__ l, float *__restrict__ r)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) {
        l[k] = st[k] + mono[k];
        r[k] = mono[k] - st[k];
    }
}

// src/filter.cpp:227-245
__global__ void upsample_kernel(const float *__restrict__ x, size_t n_out, float *__restrict__ xu, int up)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n_out) xu[k] = (k % up) == 0 ? x[k / up] : 0.0f;
}

__global__ void downsample_kernel(const float *__restrict__ in, size_t n_out, float *__restrict__ out, int ds)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n_out) out[k] = in[k * ds];
}

__global__ void fill_u8_kernel(uint8_t *d, size_t n, uint8_t v)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) d[k] = v;
}

}  // namespace

int k_fir_generic(const float *d_x, size_t n_out, const float *d_h, int taps, int decim, float *d_y, hipStream_t s)
{
    if (n_out == 0) return FMRX_OK;
    hipLaunchKernelGGL(fir_generic_kernel, dim3(grid_for(n_out)), dim3(kBlock), 0, s, d_x, n_out, d_h, taps, decim, d_y);
    FMRX_LAUNCH_CHECK("fir_generic");
    return FMRX_OK;
}

int k_fe_generic(const uint8_t *d_iq, const uint8_t *d_hist, int hist_bytes, size_t n_samples, const float *d_h,
                 int taps, int decim, float *d_if, hipStream_t s)
{
    const size_t n_out = n_samples / decim;
    if (n_out == 0) return FMRX_OK;
    hipLaunchKernelGGL(fe_generic_kernel, dim3(grid_for(n_out)), dim3(kBlock), 0, s, d_iq, d_hist, hist_bytes, n_out,
                       d_h, taps, decim, reinterpret_cast<float2 *>(d_if));
    FMRX_LAUNCH_CHECK("fe_generic");
    return FMRX_OK;
}

int k_resample_generic(const float *d_x, size_t n_in, const float *d_h, int taps, int decim, int upsamp, float *d_y,
                       hipStream_t s)
{
    const size_t n_out = (n_in * static_cast<size_t>(upsamp)) / decim;
    if (n_out == 0) return FMRX_OK;
    hipLaunchKernelGGL(resample_generic_kernel, dim3(grid_for(n_out)), dim3(kBlock), 0, s, d_x, n_out, d_h, taps, decim,
                       upsamp, d_y);
    FMRX_LAUNCH_CHECK("resample_generic");
    return FMRX_OK;
}

int k_fm_demod_if(const float *d_if, size_t n, const float *d_prev, float *d_prev_out, float *d_demod, int fast,
                  hipStream_t s)
{
    if (n == 0) return FMRX_OK;
    if (fast)
        hipLaunchKernelGGL(demod_if_kernel<true>, dim3(grid_for(n)), dim3(kBlock), 0, s,
                           reinterpret_cast<const float2 *>(d_if), n, reinterpret_cast<const float2 *>(d_prev),
                           reinterpret_cast<float2 *>(d_prev_out), d_demod);
    else
        hipLaunchKernelGGL(demod_if_kernel<false>, dim3(grid_for(n)), dim3(kBlock), 0, s,
                           reinterpret_cast<const float2 *>(d_if), n, reinterpret_cast<const float2 *>(d_prev),
                           reinterpret_cast<float2 *>(d_prev_out), d_demod);
    FMRX_LAUNCH_CHECK("demod_if");
    return FMRX_OK;
}

int k_fm_demod_arctan_planar(const double *d_i, const double *d_q, size_t n, double prev_phase, double *d_out, hipStream_t s)
{
    if (n == 0) return FMRX_OK;
    hipLaunchKernelGGL(demod_arctan_planar_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, d_i, d_q, n, prev_phase, d_out);
    FMRX_LAUNCH_CHECK("demod_arctan_planar");
    return FMRX_OK;
}

int k_fm_demod_arctan_if(const float *d_if, size_t n, const float *d_prev, float *d_demod, hipStream_t s)
{
    if (n == 0) return FMRX_OK;
    hipLaunchKernelGGL(demod_arctan_if_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, reinterpret_cast<const float2 *>(d_if), n,
                       reinterpret_cast<const float2 *>(d_prev), d_demod);
    FMRX_LAUNCH_CHECK("demod_arctan_if");
    return FMRX_OK;
}

int k_fm_demod_planar(const float *d_i, const float *d_q, size_t n, float prev_i, float prev_q, float *d_demod,
                      hipStream_t s)
{
    if (n == 0) return FMRX_OK;
    hipLaunchKernelGGL(demod_planar_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, d_i, d_q, n, prev_i, prev_q, d_demod);
    FMRX_LAUNCH_CHECK("demod_planar");
    return FMRX_OK;
}

#define FMRX_ELEMENTWISE(fn, kern, n, ...)                                                     \
    int fn                                                                                     \
    {                                                                                          \
        if ((n) == 0) return FMRX_OK;                                                          \
        hipLaunchKernelGGL(kern, dim3(grid_for(n)), dim3(kBlock), 0, s, __VA_ARGS__);          \
        FMRX_LAUNCH_CHECK(#kern);                                                              \
        return FMRX_OK;                                                                        \
    }

FMRX_ELEMENTWISE(k_u8_to_f32(const uint8_t *d_raw, size_t n, float *d_out, hipStream_t s), u8_to_f32_kernel, n, d_raw, n, d_out)
FMRX_ELEMENTWISE(k_deinterleave(const float *d_iq, size_t n, float *d_i, float *d_q, hipStream_t s), deinterleave_kernel, n,
                 reinterpret_cast<const float2 *>(d_iq), n, d_i, d_q)
FMRX_ELEMENTWISE(k_pcm16(const float *d_a, size_t n, int16_t *d_out, int wrap, hipStream_t s), pcm16_kernel, n, d_a, n, d_out, wrap)
FMRX_ELEMENTWISE(k_pcm16_stereo(const float *d_l, const float *d_r, size_t n, int16_t *d_out, int wrap, hipStream_t s),
                 pcm16_stereo_kernel, n, d_l, d_r, n, d_out, wrap)
FMRX_ELEMENTWISE(k_all_pass(const float *d_in, size_t n, const float *d_state, size_t nstate, float *d_out, hipStream_t s),
                 all_pass_kernel, n, d_in, n, d_state, nstate, d_out)
FMRX_ELEMENTWISE(k_mix(const float *d_bpf, const float *d_pll, size_t n, float *d_mix, hipStream_t s), mix_kernel, n, d_bpf, d_pll, n, d_mix)
FMRX_ELEMENTWISE(k_combine(const float *d_st, const float *d_mono, size_t n, float *d_l, float *d_r, hipStream_t s),
                 combine_kernel, n, d_st, d_mono, n, d_l, d_r)
FMRX_ELEMENTWISE(k_downsample(const float *d_in, size_t n, float *d_out, int ds, hipStream_t s), downsample_kernel, n, d_in, n, d_out, ds)
FMRX_ELEMENTWISE(k_fill_u8(uint8_t *d, size_t n, uint8_t v, hipStream_t s), fill_u8_kernel, n, d, n, v)

int k_split_if(const float *d_if, size_t n, float *d_i, float *d_q, hipStream_t s) { return k_deinterleave(d_if, n, d_i, d_q, s); }

int k_upsample(const float *d_x, size_t n, float *d_xu, int up, hipStream_t s)
{
    const size_t n_out = n * static_cast<size_t>(up);
    if (n_out == 0) return FMRX_OK;
    hipLaunchKernelGGL(upsample_kernel, dim3(grid_for(n_out)), dim3(kBlock), 0, s, d_x, n_out, d_xu, up);
    FMRX_LAUNCH_CHECK("upsample");
    return FMRX_OK;
}

}  // namespace fmrx

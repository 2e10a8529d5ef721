// capi.hip -- the extern "C" layer of libfmrx.so (include/fmrx.h): library
// functions and the host-buffer stage API, one entry point per reference
// primitive (include/filter.h:18-43, include/iofunc.h:36 under /root/reference).
//
// Every stage function: validate (the reference's unchecked preconditions
// become FMRX_EINVAL) -> H2D into per-thread scratch -> HIP kernel(s) -> D2H.
// No stage has a CPU implementation: without a device they return FMRX_ENODEV.
#include "fmrx_internal.hpp"
#include "build_id.hpp"   // FMRX_SRC_HASH: written by the Makefile (SHA-256 over the library's sources)

namespace fmrx {

// ---- error plumbing ----------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(FMRX_ENODEV, "no usable HIP device (%s); libfmrx has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    return FMRX_OK;
}

// ---- run-time options -----------------------------------------------------------
bool option_ref(Options &o, const char *name, long **as_long, int **as_int)
{
    *as_long = nullptr;
    *as_int = nullptr;
    const std::string n = name ? name : "";
    if (n == "fused_min_audio") *as_long = &o.fused_min_audio;
    else if (n == "fe_variant") *as_int = &o.fe_variant;
    else if (n == "resample_l2") *as_int = &o.resample_l2;
    else if (n == "resample_exact") *as_int = &o.resample_exact;
    else if (n == "fe_wgs_per_cu") *as_int = &o.fe_wgs_per_cu;
    else if (n == "pll_warmup") *as_int = &o.pll_warmup;
    else if (n == "pll_segment") *as_int = &o.pll_segment;
    else if (n == "pll_head") *as_int = &o.pll_head;
    else if (n == "pll_start") *as_int = &o.pll_start;
    else if (n == "pll_align") *as_int = &o.pll_align;
    else if (n == "pll_mode") *as_int = &o.pll_mode;
    else if (n == "demod") *as_int = &o.demod;
    else if (n == "bank_streams") *as_int = &o.bank_streams;
    else if (n == "bank_fe_wgs") *as_int = &o.bank_fe_wgs;
    else if (n == "bank_fused") *as_int = &o.bank_fused;
    else if (n == "bank_fe_wgs_fused") *as_int = &o.bank_fe_wgs_fused;
    else if (n == "resample_chains") *as_int = &o.resample_chains;
    else if (n == "overlap_calls") *as_int = &o.overlap_calls;
    else if (n == "fused_tune") *as_int = &o.fused_tune;
    else if (n == "fe_mfma_tune") *as_int = &o.fe_mfma_tune;
    else return false;
    return true;
}

// built-in values, overridden once by the environment (first use; thread-safe static initialisation)
Options &default_options()
{
    static Options o = [] {
        Options d;
        if (const char *e = std::getenv("FMRX_FE_VARIANT")) d.fe_variant = std::strcmp(e, "valu") == 0 ? 1 : 0;
        if (const char *e = std::getenv("FMRX_FUSED_MIN_AUDIO")) d.fused_min_audio = std::atol(e);
        if (std::getenv("FMRX_RESAMPLE_L2")) d.resample_l2 = 1;
        if (const char *e = std::getenv("FMRX_RESAMPLE_EXACT")) d.resample_exact = std::atoi(e);
        if (const char *e = std::getenv("FMRX_RESAMPLE_CHAINS")) d.resample_chains = std::atoi(e);
        if (const char *e = std::getenv("FMRX_OVERLAP_CALLS")) d.overlap_calls = std::atoi(e);
        if (const char *e = std::getenv("FMRX_FE_WGS_PER_CU")) d.fe_wgs_per_cu = std::atoi(e);
        if (const char *e = std::getenv("FMRX_PLL_WARMUP")) d.pll_warmup = std::atoi(e);
        if (const char *e = std::getenv("FMRX_PLL_SEGMENT")) d.pll_segment = std::atoi(e);
        if (const char *e = std::getenv("FMRX_PLL_HEAD")) d.pll_head = std::atoi(e);
        if (const char *e = std::getenv("FMRX_PLL_START")) d.pll_start = std::atoi(e);
        if (const char *e = std::getenv("FMRX_PLL_ALIGN")) d.pll_align = std::atoi(e);
        if (const char *e = std::getenv("FMRX_PLL_MODE")) d.pll_mode = std::atoi(e);
        if (const char *e = std::getenv("FMRX_BANK_STREAMS")) d.bank_streams = std::atoi(e);
        if (const char *e = std::getenv("FMRX_BANK_FE_WGS")) d.bank_fe_wgs = std::atoi(e);
        if (const char *e = std::getenv("FMRX_BANK_FUSED")) d.bank_fused = std::atoi(e);
        if (const char *e = std::getenv("FMRX_DEMOD")) d.demod = std::strcmp(e, "arctan") == 0 ? 1 : std::atoi(e);
#ifdef FMRX_TUNING
        if (const char *e = std::getenv("FMRX_FUSED_TUNE")) d.fused_tune = std::atoi(e);
        if (const char *e = std::getenv("FMRX_FE_MFMA_TUNE")) d.fe_mfma_tune = std::atoi(e);
#endif
        return d;
    }();
    return o;
}

std::mutex &options_mutex()
{
    static std::mutex m;
    return m;
}

Options options_snapshot()
{
    std::lock_guard<std::mutex> lock(options_mutex());
    return default_options();
}

int set_option_in(Options &o, const char *name, long value)
{
    long *pl = nullptr;
    int *pi = nullptr;
    if (!option_ref(o, name, &pl, &pi)) return fail(FMRX_EINVAL, "unknown option '%s'", name ? name : "(null)");
#ifndef FMRX_TUNING
    if ((pi == &o.fused_tune || pi == &o.fe_mfma_tune) && value != 0)
        return fail(FMRX_EINVAL, "option '%s': ablation kernels exist only in a -DFMRX_TUNING build of libfmrx", name);
#endif
    if (pi == &o.fe_variant && value != 0 && value != 1) return fail(FMRX_EINVAL, "option fe_variant: 0 (mfma) or 1 (valu)");
    if (pi == &o.pll_mode && (value < 0 || value > 2)) return fail(FMRX_EINVAL, "option pll_mode: 0, 1 or 2");
    if (pi == &o.demod && value != 0 && value != 1) return fail(FMRX_EINVAL, "option demod: 0 (the C++ reference's discriminator) or 1 (arctan)");
    if (pl) *pl = value;
    else *pi = static_cast<int>(value);
    return FMRX_OK;
}

namespace {

// per-thread device scratch for the host-buffer stage functions
struct Scratch {
    DevBuf<float> a, b, c, d, h;
    DevBuf<uint8_t> u, uh;
    DevBuf<int16_t> s16;
};
Scratch &scratch()
{
    static thread_local Scratch s;
    return s;
}

inline int h2d(void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return FMRX_OK;
    FMRX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return FMRX_OK;
}
inline int d2h(void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return FMRX_OK;
    FMRX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return FMRX_OK;
}
inline int sync0()
{
    FMRX_HIP(hipStreamSynchronize(nullptr));
    return FMRX_OK;
}

}  // namespace
}  // namespace fmrx

using namespace fmrx;

extern "C" {

const char *fmrx_version(void)
{
#ifdef FMRX_TUNING
    return "fmrx 0.3 (gfx950, TUNING build: ablation kernels included) src:" FMRX_SRC_HASH;
#else
    return "fmrx 0.3 (gfx950) src:" FMRX_SRC_HASH;
#endif
}

// The process-wide defaults may be changed by one thread while another creates a handle (which copies them): writers and the
// copy go through one mutex (options_snapshot); per-block paths only ever read a handle's own copy.
int fmrx_set_option(const char *name, long value)
{
    std::lock_guard<std::mutex> lock(options_mutex());
    return set_option_in(default_options(), name, value);
}

int fmrx_get_option(const char *name, long *value)
{
    std::lock_guard<std::mutex> lock(options_mutex());
    long *pl = nullptr;
    int *pi = nullptr;
    if (!value || !option_ref(default_options(), name, &pl, &pi)) return fail(FMRX_EINVAL, "unknown option '%s'", name ? name : "(null)");
    *value = pl ? *pl : *pi;
    return FMRX_OK;
}
const char *fmrx_last_error(void) { return g_err; }

int fmrx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int fmrx_set_device(int device)
{
    FMRX_TRY(require_device());
    FMRX_HIP(hipSetDevice(device));
    return FMRX_OK;
}

int fmrx_host_alloc(void **out, size_t bytes)
{
    if (!out || bytes == 0) return fail(FMRX_EINVAL, "host_alloc: bad arguments");
    FMRX_TRY(require_device());
    FMRX_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return FMRX_OK;
}

int fmrx_host_free(void *p)
{
    if (p) FMRX_HIP(hipHostFree(p));
    return FMRX_OK;
}

// ---- element-wise stages ---------------------------------------------------------
int fmrx_u8_to_f32(const uint8_t *raw, size_t n, float *out)
{
    if ((!raw || !out) && n) return fail(FMRX_EINVAL, "u8_to_f32: null buffer");
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.u.ensure(n));
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(h2d(s.u.p, raw, n));
    FMRX_TRY(k_u8_to_f32(s.u.p, n, s.a.p, nullptr));
    return d2h(out, s.a.p, n * sizeof(float));
}

int fmrx_deinterleave(const float *iq, size_t n_pairs, float *I, float *Q)
{
    if ((!iq || !I || !Q) && n_pairs) return fail(FMRX_EINVAL, "deinterleave: null buffer");
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(2 * n_pairs));
    FMRX_TRY(s.b.ensure(n_pairs));
    FMRX_TRY(s.c.ensure(n_pairs));
    FMRX_TRY(h2d(s.a.p, iq, 2 * n_pairs * sizeof(float)));
    FMRX_TRY(k_deinterleave(s.a.p, n_pairs, s.b.p, s.c.p, nullptr));
    FMRX_TRY(d2h(I, s.b.p, n_pairs * sizeof(float)));
    return d2h(Q, s.c.p, n_pairs * sizeof(float));
}

int fmrx_pcm16(const float *audio, size_t n, int16_t *out, int wrap)
{
    if ((!audio || !out) && n) return fail(FMRX_EINVAL, "pcm16: null buffer");
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(s.s16.ensure(n));
    FMRX_TRY(h2d(s.a.p, audio, n * sizeof(float)));
    FMRX_TRY(k_pcm16(s.a.p, n, s.s16.p, wrap, nullptr));
    return d2h(out, s.s16.p, n * sizeof(int16_t));
}

// ---- FIR family -----------------------------------------------------------------------
// device layout for a block with carried history: [history | block], kernels get
// a pointer to the block and read history at negative indices.
static int fir_common(float *y, size_t n_out, const float *x, size_t n, const float *h, size_t taps,
                      const float *state, size_t n_hist, size_t n_tail_zero, unsigned decim)
{
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n_hist + n + n_tail_zero));
    FMRX_TRY(s.h.ensure(taps));
    FMRX_TRY(s.b.ensure(n_out));
    if (state) FMRX_TRY(h2d(s.a.p, state, n_hist * sizeof(float)));
    else FMRX_HIP(hipMemset(s.a.p, 0, n_hist * sizeof(float)));
    FMRX_TRY(h2d(s.a.p + n_hist, x, n * sizeof(float)));
    if (n_tail_zero) FMRX_HIP(hipMemset(s.a.p + n_hist + n, 0, n_tail_zero * sizeof(float)));
    FMRX_TRY(h2d(s.h.p, h, taps * sizeof(float)));
    FMRX_TRY(k_fir_generic(s.a.p + n_hist, n_out, s.h.p, static_cast<int>(taps), static_cast<int>(decim), s.b.p, nullptr));
    return d2h(y, s.b.p, n_out * sizeof(float));
}

int fmrx_convolve_fir(float *y, const float *x, size_t n, const float *h, size_t taps)
{
    if (!y || !x || !h || taps == 0 || n == 0) return fail(FMRX_EINVAL, "convolve_fir: bad arguments");
    if (taps > 65535) return fail(FMRX_EINVAL, "convolve_fir: taps %zu > 65535", taps);
    FMRX_TRY(require_device());
    // full convolution = block FIR over [0^(taps-1) | x | 0^(taps-1)]
    return fir_common(y, n + taps - 1, x, n, h, taps, nullptr, taps - 1, taps - 1, 1);
}

static int block_fir(const char *name, float *y, const float *x, size_t n, const float *h, size_t taps, float *state,
                     unsigned decim)
{
    if (!y || !x || !h || !state || taps == 0) return fail(FMRX_EINVAL, "%s: null buffer", name);
    if (decim == 0) return fail(FMRX_EINVAL, "%s: decim must be >= 1", name);
    if (taps > 65535) return fail(FMRX_EINVAL, "%s: taps %zu > 65535 (unsigned short in the reference)", name, taps);
    if (n < taps - 1)
        return fail(FMRX_EINVAL, "%s: block of %zu samples is shorter than taps-1 = %zu (state refresh would read before the block)",
                    name, n, taps - 1);
    FMRX_TRY(require_device());
    FMRX_TRY(fir_common(y, n / decim, x, n, h, taps, state, taps - 1, 0, decim));
    // state <- last taps-1 samples of x (src/filter.cpp:148-153, 182-187): a host copy in the reference too
    std::memcpy(state, x + n - (taps - 1), (taps - 1) * sizeof(float));
    return FMRX_OK;
}

int fmrx_convolve_block_fir(float *y, const float *x, size_t n, const float *h, size_t taps, float *state)
{
    return block_fir("convolve_block_fir", y, x, n, h, taps, state, 1);
}

int fmrx_convolve_block_fast_fir(float *y, const float *x, size_t n, const float *h, size_t taps, float *state,
                                 unsigned decim)
{
    return block_fir("convolve_block_fast_fir", y, x, n, h, taps, state, decim);
}

int fmrx_convolve_block_resample_fir(float *y, const float *x, size_t n, const float *h, size_t taps, float *state,
                                     unsigned decim, unsigned upsamp)
{
    if (!y || !x || !h || !state || taps == 0) return fail(FMRX_EINVAL, "convolve_block_resample_fir: null buffer");
    if (decim == 0 || upsamp == 0) return fail(FMRX_EINVAL, "convolve_block_resample_fir: decim and upsamp must be >= 1");
    if (taps > 65535) return fail(FMRX_EINVAL, "convolve_block_resample_fir: taps %zu > 65535", taps);
    if (n * upsamp < taps - 1)
        return fail(FMRX_EINVAL, "convolve_block_resample_fir: n*upsamp = %zu < taps-1 = %zu", n * upsamp, taps - 1);
    FMRX_TRY(require_device());
    // The reference keeps its state in the UPSAMPLED index space: stream sample
    // x[-d] (d = 1..H) lives in state[taps-1 - d*upsamp]  (src/filter.cpp:207, 218-222).
    const size_t H = (taps - 1) / upsamp;
    std::vector<float> hist(H ? H : 1, 0.0f);
    for (size_t d = 1; d <= H; d++) hist[H - d] = state[taps - 1 - d * upsamp];
    Scratch &s = scratch();
    const size_t n_out = (n * upsamp) / decim;
    const size_t Hp = (H + 3) / 4 * 4;   // keeps the block 16-byte aligned behind its history
    FMRX_TRY(s.a.ensure(Hp + n));
    FMRX_TRY(s.b.ensure(n_out));
    FMRX_TRY(h2d(s.a.p + (Hp - H), hist.data(), H * sizeof(float)));
    FMRX_TRY(h2d(s.a.p + Hp, x, n * sizeof(float)));
    // polyphase-table kernels (bit-exact: the reference's operations in its order, kernels_resample.hip);
    // the LDS-resident-table form from 65 536 outputs per call unless the option resample_l2 is set
    // the plan (polyphase table, tap images: several device allocations and copies) is kept per thread and rebuilt only when
    // the filter changes: a block-streaming caller passes the same taps every call (src/project.cpp:353)
    struct PlanCache {
        ResamplePlan plan;
        std::vector<float> h;
        unsigned decim = 0, upsamp = 0;
    };
    static thread_local PlanCache pc;
    if (pc.decim != decim || pc.upsamp != upsamp || pc.h.size() != taps || std::memcmp(pc.h.data(), h, taps * sizeof(float)) != 0) {
        pc.decim = pc.upsamp = 0;                              // invalid until the new plan is complete
        FMRX_TRY(resample_plan_init(pc.plan, h, static_cast<int>(taps), static_cast<int>(decim), static_cast<int>(upsamp)));
        pc.h.assign(h, h + taps);
        pc.decim = decim;
        pc.upsamp = upsamp;
    }
    FMRX_TRY(resample_launch(pc.plan, s.a.p + Hp, n, 0, s.b.p, options_snapshot(), nullptr, false, /*exact=*/true));
    FMRX_TRY(sync0());
    FMRX_TRY(d2h(y, s.b.p, n_out * sizeof(float)));
    // state refresh exactly as src/filter.cpp:218-222 (host copy): k = U-1; for
    // i = U*n-(taps-1); i < U*n-U; i += U: state[k] = x[i/U + 1]; k += U
    {
        const long U = upsamp, ns = static_cast<long>(taps) - 1, N = static_cast<long>(n);
        long k = U - 1;
        for (long i = U * N - ns; i < U * N - U; i += U) {
            state[k] = x[(i / U) + 1];
            k += U;
        }
    }
    return FMRX_OK;
}

int fmrx_upsample(const float *x, size_t n, float *xu, int up)
{
    if ((!x || !xu) && n) return fail(FMRX_EINVAL, "upsample: null buffer");
    if (up < 1) return fail(FMRX_EINVAL, "upsample: rate must be >= 1");
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(s.b.ensure(n * up));
    FMRX_TRY(h2d(s.a.p, x, n * sizeof(float)));
    FMRX_TRY(k_upsample(s.a.p, n, s.b.p, up, nullptr));
    return d2h(xu, s.b.p, n * up * sizeof(float));
}

int fmrx_downsample(float *out, size_t *n_out, const float *in, size_t n, unsigned short ds)
{
    if (!out || !in || !n_out) return fail(FMRX_EINVAL, "downsample: null buffer");
    if (ds == 0) return fail(FMRX_EINVAL, "downsample: factor must be >= 1");
    FMRX_TRY(require_device());
    // size rule of src/filter.cpp:240: ceil(n / (float)ds), evaluated in float
    const size_t ny = static_cast<size_t>(ceilf(static_cast<float>(n) / static_cast<float>(ds)));
    *n_out = ny;
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n + ds));
    FMRX_TRY(s.b.ensure(ny));
    FMRX_TRY(h2d(s.a.p, in, n * sizeof(float)));
    FMRX_TRY(k_downsample(s.a.p, ny, s.b.p, ds, nullptr));
    return d2h(out, s.b.p, ny * sizeof(float));
}

// ---- demod / stereo helpers -----------------------------------------------------------
int fmrx_fm_demod(float *out, const float *I, const float *Q, size_t n, float *prev_i, float *prev_q)
{
    if (!out || !I || !Q || !prev_i || !prev_q) return fail(FMRX_EINVAL, "fm_demod: null buffer");
    if (n == 0) return FMRX_OK;
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(s.b.ensure(n));
    FMRX_TRY(s.c.ensure(n));
    FMRX_TRY(h2d(s.a.p, I, n * sizeof(float)));
    FMRX_TRY(h2d(s.b.p, Q, n * sizeof(float)));
    FMRX_TRY(k_fm_demod_planar(s.a.p, s.b.p, n, *prev_i, *prev_q, s.c.p, nullptr));
    FMRX_TRY(d2h(out, s.c.p, n * sizeof(float)));
    *prev_i = I[n - 1];
    *prev_q = Q[n - 1];
    return FMRX_OK;
}

int fmrx_fm_demod_arctan(double *out, const double *I, const double *Q, size_t n, double *prev_phase)
{
    if (!out || !I || !Q || !prev_phase) return fail(FMRX_EINVAL, "fm_demod_arctan: null buffer");
    if (n == 0) return FMRX_OK;
    FMRX_TRY(require_device());
    static thread_local DevBuf<double> di, dq, dout;
    FMRX_TRY(di.ensure(n));
    FMRX_TRY(dq.ensure(n));
    FMRX_TRY(dout.ensure(n));
    FMRX_TRY(h2d(di.p, I, n * sizeof(double)));
    FMRX_TRY(h2d(dq.p, Q, n * sizeof(double)));
    // the phase in front of the block only matters modulo 2 pi (np.unwrap's mod takes care of the turns it has accumulated)
    FMRX_TRY(k_fm_demod_arctan_planar(di.p, dq.p, n, *prev_phase, dout.p, nullptr));
    FMRX_TRY(d2h(out, dout.p, n * sizeof(double)));
    double ph = *prev_phase;                                // the model's running (unwrapped) phase: prev + the steps, in order
    for (size_t k = 0; k < n; k++) ph += out[k];
    *prev_phase = ph;
    return FMRX_OK;
}

int fmrx_all_pass(const float *in, size_t n, float *state, size_t nstate, float *out)
{
    if (!in || !state || !out) return fail(FMRX_EINVAL, "all_pass: null buffer");
    if (n < nstate) return fail(FMRX_EINVAL, "all_pass: block of %zu samples shorter than the delay %zu", n, nstate);
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(s.b.ensure(nstate));
    FMRX_TRY(s.c.ensure(n));
    FMRX_TRY(h2d(s.a.p, in, n * sizeof(float)));
    FMRX_TRY(h2d(s.b.p, state, nstate * sizeof(float)));
    FMRX_TRY(k_all_pass(s.a.p, n, s.b.p, nstate, s.c.p, nullptr));
    FMRX_TRY(d2h(out, s.c.p, n * sizeof(float)));
    std::memcpy(state, in + n - nstate, nstate * sizeof(float));
    return FMRX_OK;
}

int fmrx_fm_pll(const float *in, size_t n, float *nco_out, float *state, float freq, float Fs, float ncoScale,
                float phaseAdjust, float normBandwidth)
{
    if (!in || !nco_out || !state) return fail(FMRX_EINVAL, "fm_pll: null buffer");
    if (!(Fs > 0)) return fail(FMRX_EINVAL, "fm_pll: Fs must be positive");
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(s.b.ensure(n + 1));
    FMRX_TRY(s.c.ensure(6));
    FMRX_TRY(h2d(s.a.p, in, n * sizeof(float)));
    FMRX_TRY(h2d(s.c.p, state, 6 * sizeof(float)));
    FMRX_TRY(k_fm_pll(s.a.p, n, s.b.p, s.c.p, freq, Fs, ncoScale, phaseAdjust, normBandwidth, 0, nullptr));
    FMRX_TRY(d2h(nco_out, s.b.p, (n + 1) * sizeof(float)));
    return d2h(state, s.c.p, 6 * sizeof(float));
}

int fmrx_stereo_mix(const float *stereo_filt, const float *pll, size_t n, float *mixer)
{
    if ((!stereo_filt || !pll || !mixer) && n) return fail(FMRX_EINVAL, "stereo_mix: null buffer");
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(s.b.ensure(n));
    FMRX_TRY(s.c.ensure(n));
    FMRX_TRY(h2d(s.a.p, stereo_filt, n * sizeof(float)));
    FMRX_TRY(h2d(s.b.p, pll, n * sizeof(float)));
    FMRX_TRY(k_mix(s.a.p, s.b.p, n, s.c.p, nullptr));
    return d2h(mixer, s.c.p, n * sizeof(float));
}

int fmrx_stereo_combine(const float *stereo_final, const float *mono, size_t n, float *left, float *right)
{
    if ((!stereo_final || !mono || !left || !right) && n) return fail(FMRX_EINVAL, "stereo_combine: null buffer");
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(s.b.ensure(n));
    FMRX_TRY(s.c.ensure(n));
    FMRX_TRY(s.d.ensure(n));
    FMRX_TRY(h2d(s.a.p, stereo_final, n * sizeof(float)));
    FMRX_TRY(h2d(s.b.p, mono, n * sizeof(float)));
    FMRX_TRY(k_combine(s.a.p, s.b.p, n, s.c.p, s.d.p, nullptr));
    FMRX_TRY(d2h(left, s.c.p, n * sizeof(float)));
    return d2h(right, s.d.p, n * sizeof(float));
}

// ---- diagnostics ---------------------------------------------------------------------------
int fmrx_diag_libm(int fn, const float *a, const float *b, size_t n, float *out)
{
    if (fn < 0 || fn > 5) return fail(FMRX_EINVAL, "diag_libm: fn must be 0 (sinf), 1 (cosf), 2 (atan2f) or 3..5 (their branch-free forms)");
    if ((!a || !out || (fn % 3 == 2 && !b)) && n) return fail(FMRX_EINVAL, "diag_libm: null buffer");
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(s.b.ensure(n));
    FMRX_TRY(s.c.ensure(n));
    FMRX_TRY(h2d(s.a.p, a, n * sizeof(float)));
    if (fn % 3 == 2) FMRX_TRY(h2d(s.b.p, b, n * sizeof(float)));
    FMRX_TRY(k_libm_eval(fn, s.a.p, s.b.p, n, s.c.p, nullptr));
    return d2h(out, s.c.p, n * sizeof(float));
}

int fmrx_diag_stream_read_dev(const void *d_buf, size_t bytes, int method, void *stream)
{
    if (!d_buf || method < 0 || method > 4096) return fail(FMRX_EINVAL, "diag_stream_read_dev: bad arguments");
    FMRX_TRY(require_device());
    static thread_local DevBuf<unsigned> sink;
    FMRX_TRY(sink.ensure(4));
    return k_stream_read(d_buf, bytes, method, sink.p, static_cast<hipStream_t>(stream));
}

int fmrx_estimate_psd(float *freq, float *psd, const float *samples, size_t n, float Fs, int nfft)
{
    if (!freq || !psd || !samples) return fail(FMRX_EINVAL, "estimate_psd: null buffer");
    if (nfft < 2 || nfft % 2 || nfft > 65536) return fail(FMRX_EINVAL, "estimate_psd: nfft must be even and in 2..65536");
    if (n < static_cast<size_t>(nfft)) return fail(FMRX_EINVAL, "estimate_psd: %zu samples < nfft %d", n, nfft);
    if (!(Fs > 0)) return fail(FMRX_EINVAL, "estimate_psd: Fs must be positive");
    FMRX_TRY(require_device());
    Scratch &s = scratch();
    const size_t nseg = n / nfft, half = nfft / 2;
    FMRX_TRY(s.a.ensure(n));
    FMRX_TRY(s.b.ensure(nseg * half));
    FMRX_TRY(s.c.ensure(half));
    FMRX_TRY(s.d.ensure(half));
    FMRX_TRY(h2d(s.a.p, samples, nseg * nfft * sizeof(float)));
    FMRX_TRY(k_estimate_psd(s.a.p, n, Fs, nfft, s.b.p, s.c.p, s.d.p, nullptr));
    FMRX_TRY(d2h(freq, s.c.p, half * sizeof(float)));
    return d2h(psd, s.d.p, half * sizeof(float));
}

// ---- fused front end as a stage ----------------------------------------------------------
struct fmrx_fe_plan {
    FePlan plan;
};

int fmrx_fe_plan_create(fmrx_fe_plan **out, const float *h, size_t taps, unsigned decim)
{
    if (!out || !h) return fail(FMRX_EINVAL, "fe_plan_create: null argument");
    if (taps < 2 || taps > 65535) return fail(FMRX_EINVAL, "fe_plan_create: taps %zu not in 2..65535", taps);
    if (decim == 0) return fail(FMRX_EINVAL, "fe_plan_create: decim must be >= 1");
    FMRX_TRY(require_device());
    fmrx_fe_plan *p = new fmrx_fe_plan;
    int rc = fe_plan_init(p->plan, h, static_cast<int>(taps), static_cast<int>(decim));
    if (rc != FMRX_OK) {
        delete p;
        return rc;
    }
    *out = p;
    return FMRX_OK;
}

int fmrx_fe_plan_destroy(fmrx_fe_plan *plan)
{
    delete plan;
    return FMRX_OK;
}

int fmrx_fe_plan_is_specialised(const fmrx_fe_plan *plan) { return plan && plan->plan.fast ? 1 : 0; }
size_t fmrx_fe_plan_history_bytes(const fmrx_fe_plan *plan) { return plan ? plan->plan.hist_bytes : 0; }

int fmrx_fe_run_dev(const fmrx_fe_plan *plan, const uint8_t *d_iq, size_t n_samples, const uint8_t *d_hist, float *d_if,
                    int force_generic, void *stream)
{
    if (!plan || !d_iq || !d_if) return fail(FMRX_EINVAL, "fe_run_dev: null argument");
    return fe_launch(plan->plan, d_iq, n_samples, d_hist, d_if, default_options(), static_cast<hipStream_t>(stream),
                     force_generic != 0);
}

int fmrx_fe_fir_decim_u8(const uint8_t *iq, size_t n_samples, const float *h, size_t taps, unsigned decim, uint8_t *hist,
                         float *if_i, float *if_q, int force_generic)
{
    if (!iq || !h) return fail(FMRX_EINVAL, "fe_fir_decim_u8: null buffer");
    if (taps < 2 || taps > 65535) return fail(FMRX_EINVAL, "fe_fir_decim_u8: taps %zu not in 2..65535", taps);
    if (decim == 0) return fail(FMRX_EINVAL, "fe_fir_decim_u8: decim must be >= 1");
    if (hist && n_samples < taps - 1)
        return fail(FMRX_EINVAL, "fe_fir_decim_u8: block of %zu samples shorter than taps-1 = %zu", n_samples, taps - 1);
    FMRX_TRY(require_device());
    FePlan plan;
    FMRX_TRY(fe_plan_init(plan, h, static_cast<int>(taps), static_cast<int>(decim)));
    Scratch &s = scratch();
    const size_t n_out = n_samples / decim;
    const size_t hb = plan.hist_bytes, live = 2 * (taps - 1);
    FMRX_TRY(s.u.ensure(2 * n_samples));
    FMRX_TRY(s.uh.ensure(hb));
    FMRX_TRY(s.a.ensure(2 * n_out));
    FMRX_TRY(s.b.ensure(n_out));
    FMRX_TRY(s.c.ensure(n_out));
    FMRX_TRY(h2d(s.u.p, iq, 2 * n_samples));
    if (hist) {
        FMRX_TRY(k_fill_u8(s.uh.p, hb, 128, nullptr));
        FMRX_TRY(h2d(s.uh.p + (hb - live), hist, live));
    }
    FMRX_TRY(fe_launch(plan, s.u.p, n_samples, hist ? s.uh.p : nullptr, s.a.p, default_options(), nullptr, force_generic != 0));
    FMRX_TRY(k_split_if(s.a.p, n_out, s.b.p, s.c.p, nullptr));
    if (if_i) FMRX_TRY(d2h(if_i, s.b.p, n_out * sizeof(float)));
    if (if_q) FMRX_TRY(d2h(if_q, s.c.p, n_out * sizeof(float)));
    FMRX_TRY(sync0());
    if (hist) std::memcpy(hist, iq + 2 * n_samples - live, live);  // carry: the last taps-1 samples, as bytes
    return FMRX_OK;
}

}  // extern "C"

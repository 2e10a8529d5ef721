// rds.hip -- the RDS path: include/fmrx.h fmrx_rds_*.
//
// The reference has this path only as a Python / NumPy model in float64 (model/fmMonoBlock.py:238-296 on top of
// model/fmSupportLib.py; it never reached the C++, report p.8), so float64 is the arithmetic to match and the MI355X's
// full-rate FP64 vector ALUs run it.  Per block of discriminator output (fm_demod, what the front-end kernels produce):
//   rds_channel = band-pass 54-60 kHz (151 taps)                       fmMonoBlock.py:241   rds_fir_kernel
//   rds_allpass = delay by 75 samples                                  :245                 an index offset
//   rds_carrier = band-pass 113.5-114.5 kHz of rds_channel^2           :248-251             rds_fir_kernel (squares on the fly)
//   PLL at 114 kHz, ncoScale 0.5, phaseAdjust 3pi/8, bandwidth 0.002   :254                 rds_pll_kernel (serial) + rds_nco_kernel
//   mixer I / Q = NCO * allpass * 2                                    :259, :270           rds_mix_kernel
//   rational resampler U/D (247/960 in mode 0), 101*U taps, 3 kHz      :262, :271           rds_resample_kernel (gain U, as the model)
//   root-raised-cosine matched filter (101 taps)                       :266, :273           rds_fir_kernel
// and on the host, as in the model, clock and data recovery, Manchester and differential decoding, frame synchronisation
// (fmSupportLib.py:103-249, 30-100) on the 61 750 Hz output.  Histories are carried as raw samples in front of each buffer.
#include "fmrx_internal.hpp"

#include <cmath>

#pragma clang fp contract(off)

using namespace fmrx;

namespace {

constexpr double kPi = 3.141592653589793;   // math.pi / np.pi

__global__ void rds_cvt_kernel(const float *__restrict__ x, long n, double *__restrict__ y)
{
    const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) y[i] = static_cast<double>(x[i]);
}

// y[k] = sum_j h[j] * f(x[k-j]), f = identity or square; x[-(taps-1)..-1] readable
__global__ void rds_fir_kernel(const double *__restrict__ x, long n, const double *__restrict__ h, int taps, double *__restrict__ y, int square)
{
    const long k = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double acc = 0.0;
    for (int j = taps - 1; j >= 0; j--) {              // from the last tap inwards, as lfilter's transposed form adds them up
        const double v = x[k - j];
        acc = h[j] * (square ? v * v : v) + acc;
    }
    y[k] = acc;
}

// the recurrence of fmPll (fmSupportLib.py:297-353), float64; arg[k] = trigArg of step k (the NCO pair is applied by rds_nco_kernel)
__global__ void rds_pll_kernel(const double *__restrict__ x, long n, double *__restrict__ arg, double *__restrict__ state, double freq,
                               double Fs, double normBandwidth)
{
    if (blockIdx.x || threadIdx.x) return;
    const double Kp = normBandwidth * 2.666, Ki = normBandwidth * normBandwidth * 3.555;
    double integ = state[0], phase = state[1], fI = state[2], fQ = state[3], off = state[5];
    const double w = 2 * kPi * (freq / Fs);
    double last = 0.0;
    for (long k = 0; k < n; k++) {
        const double eD = atan2(x[k] * (-fQ), x[k] * (+fI));
        integ = integ + Ki * eD;
        phase = phase + Kp * eD + integ;
        off += 1;
        last = w * off + phase;
        sincos(last, &fQ, &fI);
        arg[k] = last;
    }
    state[0] = integ; state[1] = phase; state[2] = fI; state[3] = fQ; state[5] = off;
    state[7] = last;                                       // raw: rds_nco_kernel finishes state[4], state[6]
}

__global__ void rds_nco_kernel(const double *__restrict__ arg, long n, double ncoScale, double phaseAdjust, double *__restrict__ out_i,
                               double *__restrict__ out_q, double *__restrict__ state)
{
    const long k = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k == 0) {                                          // out[0] = the previous block's last value
        out_i[0] = state[4];
        out_q[0] = state[6];
    }
    if (k >= n) return;
    double s, c;
    sincos(arg[k] * ncoScale + phaseAdjust, &s, &c);
    out_i[k + 1] = c;
    out_q[k + 1] = s;
}
__global__ void rds_nco_state_kernel(long n, const double *__restrict__ out_i, const double *__restrict__ out_q, double *__restrict__ state)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state[4] = out_i[n];
        state[6] = out_q[n];
    }
}

// mixer = NCO[:-1] * allpass * 2, allpass = channel delayed by `delay`
__global__ void rds_mix_kernel(const double *__restrict__ nco_i, const double *__restrict__ nco_q, const double *__restrict__ ch, int delay,
                               long n, double *__restrict__ mi, double *__restrict__ mq)
{
    const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ap = ch[i - delay];
    mi[i] = nco_i[i] * ap * 2;
    mq[i] = nco_q[i] * ap * 2;
}

// convolveBlockResampleFIR of the Python model (fmSupportLib.py:388-407) in stream form, gain U
__global__ void rds_resample_kernel(const double *__restrict__ x, long n_out, const double *__restrict__ h, int taps, int decim, int upsamp,
                                    double *__restrict__ y)
{
    const long k = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    const long m = k * decim;
    const int ph = static_cast<int>(m % upsamp);
    const long b = m / upsamp;
    double acc = 0.0;
    long j = 0;
    for (int t = ph; t < taps; t += upsamp, j++) acc = acc + h[t] * x[b - j];
    y[k] = acc * upsamp;
}

__global__ void rds_tail_kernel(double *__restrict__ buf, long n, int hist)
{
    // buf = [hist | n]: history <- the last hist samples (n >= hist: no overlap hazard)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hist) buf[i] = buf[n + i];
}

unsigned g1(long n) { return static_cast<unsigned>((n + 255) / 256); }

// ---- host: coefficient design, float64 (fmSupportLib.py:358-385, 251-287) ----------------------------------------
void design_lpf64(int n, double Fs, double Fc, double *h)
{
    const double norm = Fc / (Fs / 2), c = (n - 1) / 2.0;
    for (int i = 0; i < n; i++) {
        double v;
        if (i == c) v = norm;
        else {
            const double a = kPi * norm * (i - c);
            v = norm * (std::sin(a) / a);
        }
        const double w = std::sin(i * kPi / n);
        h[i] = v * (w * w);
    }
}
void design_bpf64(int n, double Fs, double Fb, double Fe, double *h)
{
    const double center = ((Fe + Fb) / 2) / (Fs / 2), width = (Fe - Fb) / (Fs / 2), c = (n - 1) / 2.0;
    for (int i = 0; i < n; i++) {
        double v;
        if (i == c) v = width;
        else {
            const double a = kPi * width / 2 * (i - c);
            v = width * (std::sin(a) / a);
        }
        v = v * std::cos(i * kPi * center);
        const double w = std::sin(i * kPi / n);
        h[i] = v * (w * w);
    }
}
void design_rrc64(double Fs, int n, double *h)
{
    const double T = 1 / 2375.0, beta = 0.90;
    for (int k = 0; k < n; k++) {
        const double t = (k - n / 2.0) / Fs;
        if (t == 0.0) h[k] = 1.0 + beta * ((4 / kPi) - 1);
        else if (t == -T / (4 * beta) || t == T / (4 * beta))
            h[k] = (beta / std::sqrt(2.0)) * (((1 + 2 / kPi) * (std::sin(kPi / (4 * beta)))) + ((1 - 2 / kPi) * (std::cos(kPi / (4 * beta)))));
        else
            h[k] = (std::sin(kPi * t * (1 - beta) / T) + 4 * beta * (t / T) * std::cos(kPi * t * (1 + beta) / T)) /
                   (kPi * t * (1 - (4 * beta * t / T) * (4 * beta * t / T)) / T);
    }
}

// ---- host: bit recovery (fmSupportLib.py:103-249, 30-100) -----------------------------------------------------------
int symbol_to_bit(const double *pair) { return pair[0] > 0 ? 1 : 0; }

// state = {pair0, pair1, start, prev_size}; bits: room for n/sps + 2
size_t cdr(const double *x, size_t n, int sps, int block_count, double *state, uint8_t *bits)
{
    double pair[2] = {state[0], state[1]};
    const long start0 = static_cast<long>(state[2]), prev_size = static_cast<long>(state[3]);
    long start = start0;
    std::vector<uint8_t> head;
    std::vector<double> pts(n, 0.0), samples;
    long size = 0;
    for (;;) {
        std::fill(pts.begin(), pts.end(), 0.0);
        size = 0;
        for (long i = start; i < static_cast<long>(n); i += sps) {
            if (i == start && start == start0 && prev_size % 2 == 1) {   // the point that completes the previous block's pair
                pair[1] = x[i];
                head.push_back(static_cast<uint8_t>(symbol_to_bit(pair)));
                pair[0] = pair[1];
                start += sps;                                            // (the scan goes on from where it is)
                continue;
            }
            const bool far = i >= start + 2L * sps;
            const double a = far ? pts[i - 2L * sps] : 0.0, b = far ? pts[i - sps] : 0.0;
            if (far && a > 0 && b > 0 && x[i] > 0) pts[i] = -x[i];       // the third of three high / low points is flipped
            else if (far && a < 0 && b < 0 && x[i] < 0) pts[i] = -x[i];
            else pts[i] = x[i];
            size++;
        }
        samples.assign(static_cast<size_t>(size), 0.0);
        for (long i = start; i < static_cast<long>(n); i += sps) samples[(i - start) / sps] = pts[i];
        bool again = false;
        for (size_t i = 0; i + 1 < samples.size(); i += 2) {
            if ((samples[i] < 0 && samples[i + 1] < 0) || (samples[i] > 0 && samples[i + 1] > 0)) {
                if (std::fabs(samples[i]) < 0.3 || std::fabs(samples[i + 1]) < 0.3) {
                    if (std::fabs(samples[i]) < 0.3) samples[i] = -samples[i];
                    else samples[i + 1] = -samples[i + 1];
                } else {                                                 // cannot be mended: re-start one symbol later
                    start += sps;
                    if (block_count != 0) {
                        pair[1] = samples[0];
                        head.push_back(static_cast<uint8_t>(symbol_to_bit(pair)));
                        pair[0] = pair[1];
                    }
                    again = true;
                    break;
                }
            }
        }
        if (!again) break;
    }
    pair[0] = samples.empty() ? pair[0] : samples.back();
    const long last_index = (size - 1) * sps + start;
    state[0] = pair[0];
    state[1] = pair[1];
    state[2] = static_cast<double>(sps - (static_cast<long>(n) - last_index));
    state[3] = static_cast<double>(size);
    size_t nb = 0;
    for (uint8_t b : head) bits[nb++] = b;
    for (size_t i = 0; i + 1 < samples.size(); i += 2) bits[nb++] = (samples[i] > 0 && samples[i + 1] < 0) ? 1 : 0;   // manchestering
    return nb;
}

const char *frame_sync(const uint8_t *bits, size_t n, size_t *next_index)
{
    static const unsigned parity[26] = {0x200, 0x100, 0x080, 0x040, 0x020, 0x010, 0x008, 0x004, 0x002, 0x001, 0x2DC, 0x16E, 0x0B7,
                                        0x287, 0x39F, 0x313, 0x355, 0x376, 0x1BB, 0x201, 0x3DC, 0x1EE, 0x0F7, 0x2A7, 0x38F, 0x31B};
    const char *off = " ";
    size_t i = 0;
    while (i + 26 < n) {
        unsigned s = 0;
        for (int k = 0; k < 26; k++)
            if (bits[i + k] == 1) s ^= parity[k];
        const char *hit = s == 0x3D8 ? "A" : s == 0x3D4 ? "B" : s == 0x25C ? "C" : s == 0x3CC ? "C_apos" : s == 0x258 ? "D" : nullptr;
        if (hit) {
            off = hit;
            if (n - (i + 26) < 26) break;
            i += 26;
        } else {
            i += 1;
        }
    }
    *next_index = off[0] == ' ' ? i : i + 26;
    return off;
}

}  // namespace

struct fmrx_rds {
    fmrx_rds_params p{};
    int device = 0;
    size_t max_n = 0;
    int Hx = 0, Hc = 0, Hm = 0, Hr = 0, delay = 0;
    long block = 0;
    hipStream_t stream = nullptr;
    DevBuf<float> in;
    DevBuf<double> h_ch, h_car, h_rs, h_rrc, xd, ch, car, arg, pll_i, pll_q, mi, mq, ri, rq, yi, yq, state;
    std::vector<uint8_t> decoded;
    size_t last_n = 0, last_out = 0;
};

extern "C" {

int fmrx_rds_band_pass(int taps, double Fs, double Fb, double Fe, double *h)
{
    if (!h || taps < 2) return fail(FMRX_EINVAL, "rds_band_pass: bad arguments");
    design_bpf64(taps, Fs, Fb, Fe, h);
    return FMRX_OK;
}
int fmrx_rds_imp_response(int taps, double Fs, double Fc, double *h)
{
    if (!h || taps < 2) return fail(FMRX_EINVAL, "rds_imp_response: bad arguments");
    design_lpf64(taps, Fs, Fc, h);
    return FMRX_OK;
}
int fmrx_rds_rrc(double Fs, int taps, double *h)
{
    if (!h || taps < 2) return fail(FMRX_EINVAL, "rds_rrc: bad arguments");
    design_rrc64(Fs, taps, h);
    return FMRX_OK;
}
int fmrx_rds_cdr(const double *x, size_t n, int sps, int block_count, double *state4, uint8_t *bits, size_t *n_bits)
{
    if (!x || !state4 || !bits || !n_bits || sps < 1) return fail(FMRX_EINVAL, "rds_cdr: bad arguments");
    *n_bits = cdr(x, n, sps, block_count, state4, bits);
    return FMRX_OK;
}
int fmrx_rds_diff_decode(const uint8_t *in, size_t n, uint8_t *out)
{
    if ((!in || !out) && n) return fail(FMRX_EINVAL, "rds_diff_decode: null buffer");
    for (size_t i = 0; i < n; i++) out[i] = i == 0 ? in[0] : (in[i] != in[i - 1]);
    return FMRX_OK;
}
int fmrx_rds_frame_sync(const uint8_t *bits, size_t n, char *offset_type, size_t *next_index)
{
    if ((!bits && n) || !offset_type || !next_index) return fail(FMRX_EINVAL, "rds_frame_sync: null argument");
    std::strcpy(offset_type, frame_sync(bits, n, next_index));
    return FMRX_OK;
}

int fmrx_rds_mode_params(int mode, fmrx_rds_params *p)
{
    if (!p) return fail(FMRX_EINVAL, "rds_mode_params: null");
    // model/fmMonoBlock.py:72-90: the model defines the RDS rates for modes 0 and 2 only
    if (mode == 0) *p = fmrx_rds_params{240000, 151, 247, 960, 26, 101};
    else if (mode == 2) *p = fmrx_rds_params{240000, 151, 817, 1920, 43, 101};
    else return fail(FMRX_EINVAL, "rds_mode_params: the reference's model defines RDS parameters for modes 0 and 2 only");
    return FMRX_OK;
}

int fmrx_rds_create(fmrx_rds **out, const fmrx_rds_params *p, size_t max_block, int device)
{
    if (!out || !p) return fail(FMRX_EINVAL, "rds_create: null argument");
    if (p->taps < 3 || p->taps > 65535 || p->upsamp < 1 || p->decim < 1 || p->sps < 1 || p->rrc_taps < 2 || p->if_Fs <= 0)
        return fail(FMRX_EINVAL, "rds_create: bad parameters");
    FMRX_TRY(require_device());
    FMRX_HIP(hipSetDevice(device));
    fmrx_rds *r = new fmrx_rds;
    r->p = *p;
    r->device = device;
    r->max_n = max_block;
    r->delay = (p->taps - 1) / 2;
    r->Hx = p->taps - 1;
    r->Hm = (101 * p->upsamp - 1) / p->upsamp;
    r->Hc = std::max(p->taps - 1, r->delay + 1);
    r->Hr = p->rrc_taps - 1;
    auto body = [&]() -> int {
        const int rs_taps = 101 * p->upsamp;
        std::vector<double> h(std::max(rs_taps, p->taps));
        auto up = [&](DevBuf<double> &d, int n) -> int {
            FMRX_TRY(d.alloc(n));
            FMRX_HIP(hipMemcpy(d.p, h.data(), n * sizeof(double), hipMemcpyHostToDevice));
            return FMRX_OK;
        };
        design_bpf64(p->taps, p->if_Fs, 54e3, 60e3, h.data());            // fmMonoBlock.py:138
        FMRX_TRY(up(r->h_ch, p->taps));
        design_bpf64(p->taps, p->if_Fs, 113.5e3, 114.5e3, h.data());      // :139
        FMRX_TRY(up(r->h_car, p->taps));
        design_lpf64(rs_taps, static_cast<double>(p->if_Fs) * p->upsamp, 3e3, h.data());   // :140
        FMRX_TRY(up(r->h_rs, rs_taps));
        design_rrc64(2375.0 * p->sps, p->rrc_taps, h.data());             // :141
        FMRX_TRY(up(r->h_rrc, p->rrc_taps));
        const size_t n = max_block, no = n * p->upsamp / p->decim + 1;
        FMRX_TRY(r->in.alloc(n));
        FMRX_TRY(r->xd.alloc(r->Hx + n));
        FMRX_TRY(r->ch.alloc(r->Hc + n));
        FMRX_TRY(r->car.alloc(n));
        FMRX_TRY(r->arg.alloc(n));
        FMRX_TRY(r->pll_i.alloc(n + 1));
        FMRX_TRY(r->pll_q.alloc(n + 1));
        FMRX_TRY(r->mi.alloc(r->Hm + n));
        FMRX_TRY(r->mq.alloc(r->Hm + n));
        FMRX_TRY(r->ri.alloc(r->Hr + no));
        FMRX_TRY(r->rq.alloc(r->Hr + no));
        FMRX_TRY(r->yi.alloc(no));
        FMRX_TRY(r->yq.alloc(no));
        FMRX_TRY(r->state.alloc(8));
        FMRX_HIP(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
        return fmrx_rds_reset(r);
    };
    const int rc = body();
    if (rc != FMRX_OK) {
        fmrx_rds_destroy(r);
        return rc;
    }
    *out = r;
    return FMRX_OK;
}

int fmrx_rds_reset(fmrx_rds *r)
{
    if (!r) return fail(FMRX_EINVAL, "rds_reset: null handle");
    FMRX_HIP(hipSetDevice(r->device));
    FMRX_HIP(hipDeviceSynchronize());
    FMRX_HIP(hipMemset(r->xd.p, 0, r->Hx * sizeof(double)));
    FMRX_HIP(hipMemset(r->ch.p, 0, r->Hc * sizeof(double)));
    FMRX_HIP(hipMemset(r->mi.p, 0, r->Hm * sizeof(double)));
    FMRX_HIP(hipMemset(r->mq.p, 0, r->Hm * sizeof(double)));
    FMRX_HIP(hipMemset(r->ri.p, 0, r->Hr * sizeof(double)));
    FMRX_HIP(hipMemset(r->rq.p, 0, r->Hr * sizeof(double)));
    const double init[8] = {0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0};      // fmMonoBlock.py:186
    FMRX_HIP(hipMemcpy(r->state.p, init, sizeof(init), hipMemcpyHostToDevice));
    r->decoded.clear();
    r->block = 0;
    return FMRX_OK;
}

int fmrx_rds_destroy(fmrx_rds *r)
{
    if (!r) return FMRX_OK;
    (void)hipSetDevice(r->device);
    if (r->stream) {
        (void)hipStreamSynchronize(r->stream);
        (void)hipStreamDestroy(r->stream);
    }
    delete r;
    return FMRX_OK;
}

size_t fmrx_rds_n_out(const fmrx_rds *r, size_t n) { return r ? n * r->p.upsamp / r->p.decim : 0; }

int fmrx_rds_process_dev(fmrx_rds *r, const float *d_demod, size_t n, void *stream)
{
    if (!r || !d_demod) return fail(FMRX_EINVAL, "rds_process_dev: null argument");
    if (n == 0 || n > r->max_n) return fail(FMRX_EINVAL, "rds_process_dev: block of %zu samples (max %zu)", n, r->max_n);
    if ((n * r->p.upsamp) % r->p.decim) return fail(FMRX_EINVAL, "rds_process_dev: n*upsamp = %zu not a multiple of decim %d", n * r->p.upsamp, r->p.decim);
    // every carried history is refreshed by a copy of its buffer's tail to its front (rds_tail_kernel): the block must be at
    // least as long as each of them, or source and destination of that copy overlap
    if (n < static_cast<size_t>(r->Hc) || n < static_cast<size_t>(r->Hx) || n < static_cast<size_t>(r->Hm) ||
        n * r->p.upsamp / r->p.decim < static_cast<size_t>(r->Hr))
        return fail(FMRX_EINVAL, "rds_process_dev: block of %zu samples is shorter than a filter history (%d / %d / %d input samples, %d resampled)",
                    n, r->Hx, r->Hc, r->Hm, r->Hr);
    FMRX_HIP(hipSetDevice(r->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const fmrx_rds_params &p = r->p;
    const long N = static_cast<long>(n), NO = static_cast<long>(n * p.upsamp / p.decim);
    double *x = r->xd.p + r->Hx, *ch = r->ch.p + r->Hc, *mi = r->mi.p + r->Hm, *mq = r->mq.p + r->Hm, *ri = r->ri.p + r->Hr, *rq = r->rq.p + r->Hr;
    hipLaunchKernelGGL(rds_cvt_kernel, dim3(g1(N)), dim3(256), 0, s, d_demod, N, x);
    hipLaunchKernelGGL(rds_fir_kernel, dim3(g1(N)), dim3(256), 0, s, x, N, r->h_ch.p, p.taps, ch, 0);
    hipLaunchKernelGGL(rds_fir_kernel, dim3(g1(N)), dim3(256), 0, s, ch, N, r->h_car.p, p.taps, r->car.p, 1);
    hipLaunchKernelGGL(rds_pll_kernel, dim3(1), dim3(64), 0, s, r->car.p, N, r->arg.p, r->state.p, 114e3, static_cast<double>(p.if_Fs), 0.002);
    hipLaunchKernelGGL(rds_nco_kernel, dim3(g1(N)), dim3(256), 0, s, r->arg.p, N, 0.5, 3 * kPi / 8, r->pll_i.p, r->pll_q.p, r->state.p);
    hipLaunchKernelGGL(rds_nco_state_kernel, dim3(1), dim3(64), 0, s, N, r->pll_i.p, r->pll_q.p, r->state.p);
    hipLaunchKernelGGL(rds_mix_kernel, dim3(g1(N)), dim3(256), 0, s, r->pll_i.p, r->pll_q.p, ch, r->delay, N, mi, mq);
    hipLaunchKernelGGL(rds_resample_kernel, dim3(g1(NO)), dim3(256), 0, s, mi, NO, r->h_rs.p, 101 * p.upsamp, p.decim, p.upsamp, ri);
    hipLaunchKernelGGL(rds_resample_kernel, dim3(g1(NO)), dim3(256), 0, s, mq, NO, r->h_rs.p, 101 * p.upsamp, p.decim, p.upsamp, rq);
    hipLaunchKernelGGL(rds_fir_kernel, dim3(g1(NO)), dim3(256), 0, s, ri, NO, r->h_rrc.p, p.rrc_taps, r->yi.p, 0);
    hipLaunchKernelGGL(rds_fir_kernel, dim3(g1(NO)), dim3(256), 0, s, rq, NO, r->h_rrc.p, p.rrc_taps, r->yq.p, 0);
    // histories for the next block (raw samples in front of each buffer)
    hipLaunchKernelGGL(rds_tail_kernel, dim3(g1(r->Hx)), dim3(256), 0, s, r->xd.p, N, r->Hx);
    hipLaunchKernelGGL(rds_tail_kernel, dim3(g1(r->Hc)), dim3(256), 0, s, r->ch.p, N, r->Hc);
    hipLaunchKernelGGL(rds_tail_kernel, dim3(g1(r->Hm)), dim3(256), 0, s, r->mi.p, N, r->Hm);
    hipLaunchKernelGGL(rds_tail_kernel, dim3(g1(r->Hm)), dim3(256), 0, s, r->mq.p, N, r->Hm);
    hipLaunchKernelGGL(rds_tail_kernel, dim3(g1(r->Hr)), dim3(256), 0, s, r->ri.p, NO, r->Hr);
    hipLaunchKernelGGL(rds_tail_kernel, dim3(g1(r->Hr)), dim3(256), 0, s, r->rq.p, NO, r->Hr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "rds kernels: %s", hipGetErrorString(e));
    r->last_n = n;
    r->last_out = static_cast<size_t>(NO);
    return FMRX_OK;
}

int fmrx_rds_process(fmrx_rds *r, const float *fm_demod, size_t n, double *rrc_i, double *rrc_q, uint8_t *bits, size_t *n_bits,
                     char *offset_type)
{
    if (!r || !fm_demod) return fail(FMRX_EINVAL, "rds_process: null argument");
    if (n > r->max_n) return fail(FMRX_EINVAL, "rds_process: block of %zu samples (max %zu)", n, r->max_n);
    FMRX_HIP(hipSetDevice(r->device));
    hipStream_t s = r->stream;
    FMRX_HIP(hipMemcpyAsync(r->in.p, fm_demod, n * sizeof(float), hipMemcpyHostToDevice, s));
    FMRX_TRY(fmrx_rds_process_dev(r, r->in.p, n, s));
    const size_t no = r->last_out;
    std::vector<double> yi(no);
    FMRX_HIP(hipMemcpyAsync(yi.data(), r->yi.p, no * sizeof(double), hipMemcpyDeviceToHost, s));
    if (rrc_q) FMRX_HIP(hipMemcpyAsync(rrc_q, r->yq.p, no * sizeof(double), hipMemcpyDeviceToHost, s));
    FMRX_HIP(hipStreamSynchronize(s));
    if (rrc_i) std::memcpy(rrc_i, yi.data(), no * sizeof(double));
    // fmMonoBlock.py:276-297: clock and data recovery (its state is re-made every block there), differential decoding,
    // frame synchronisation over the bits kept so far
    double st[4] = {0.0, 0.0, 158.0, 0.0};
    std::vector<uint8_t> b(no / r->p.sps + 4), d;
    const size_t nb = cdr(yi.data(), no, r->p.sps, static_cast<int>(r->block), st, b.data());
    d.resize(nb);
    for (size_t i = 0; i < nb; i++) d[i] = i == 0 ? b[0] : (b[i] != b[i - 1]);
    if (bits) std::memcpy(bits, d.data(), nb);
    if (n_bits) *n_bits = nb;
    r->decoded.insert(r->decoded.end(), d.begin(), d.end());
    size_t next = 0;
    const char *off = frame_sync(r->decoded.data(), r->decoded.size(), &next);
    r->decoded.erase(r->decoded.begin(), r->decoded.begin() + static_cast<long>(std::min(next, r->decoded.size())));
    if (offset_type) std::strcpy(offset_type, off);
    r->block++;
    return FMRX_OK;
}

int fmrx_rds_read_tap(fmrx_rds *r, int which, double *out, size_t *n)
{
    if (!r || !n) return fail(FMRX_EINVAL, "rds_read_tap: null argument");
    FMRX_HIP(hipSetDevice(r->device));
    FMRX_HIP(hipDeviceSynchronize());
    const double *src = nullptr;
    size_t cnt = 0;
    switch (which) {
    case FMRX_RDS_TAP_CHANNEL: src = r->ch.p + r->Hc; cnt = r->last_n; break;   // after the tail copy the block region is intact
    case FMRX_RDS_TAP_CARRIER: src = r->car.p; cnt = r->last_n; break;
    case FMRX_RDS_TAP_PLL_I: src = r->pll_i.p; cnt = r->last_n + 1; break;
    case FMRX_RDS_TAP_PLL_Q: src = r->pll_q.p; cnt = r->last_n + 1; break;
    case FMRX_RDS_TAP_RESAMPLED_I: src = r->ri.p + r->Hr; cnt = r->last_out; break;
    case FMRX_RDS_TAP_RRC_I: src = r->yi.p; cnt = r->last_out; break;
    case FMRX_RDS_TAP_RRC_Q: src = r->yq.p; cnt = r->last_out; break;
    case FMRX_RDS_TAP_PLL_STATE: src = r->state.p; cnt = 7; break;
    default: return fail(FMRX_EINVAL, "rds_read_tap: unknown tap %d", which);
    }
    *n = cnt;
    if (!out || cnt == 0) return FMRX_OK;
    FMRX_HIP(hipMemcpy(out, src, cnt * sizeof(double), hipMemcpyDeviceToHost));
    return FMRX_OK;
}

}  // extern "C"

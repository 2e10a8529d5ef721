// pipeline.hip -- the receiver pipeline handle of include/fmrx.h.
//
// Replaces main()'s setup and the three thread bodies of the reference:
//   RF_FrontEnd  src/project.cpp:40-152   (u8 -> IF I/Q -> FM demod)
//   RF_MONO      src/project.cpp:311-382  (audio FIR + decimate | resample)
//   RF_STEREO    src/project.cpp:154-309  (all-pass, pilot/stereo BPFs, PLL,
//                                          mixer, second audio FIR, L/R)
// with every intermediate resident in HBM and the carried state
// (src/project.cpp:61-65, 446-458) owned by the handle.  The producer/consumer
// queue of the reference (project.cpp:470-496) has no counterpart here: stages
// are kernels on one HIP stream, so the hand-off is stream order.
//
// Device layout of one block (n = complex input samples, n_if = n/rf_decim):
//   in      u8   [2n]                         raw I/Q (host-buffer entry point only; process_dev reads the caller's)
//   fe_hist u8   [hist_bytes] x2              the stream's last bytes (= I_state/Q_state), ping-pong
//   ifb     f32  [2 n_if]                     interleaved IF I,Q -- only with set_keep_intermediates / generic path
//   demod   f32  [Hd | n_if] x2               discriminator output; the previous block's tail is the history
//   mono    f32  [n_audio]                    (mono modes 0/1 write straight into the caller's buffers)
//   stereo: carrier, bpf [n_if]; pll [n_if+1]; mixer [Hm | n_if]; final, L, R [n_audio]
// The all-pass delay of the stereo path (project.cpp:194, filter.cpp:14-29) is
// not a kernel: the mono audio FIR simply reads demod `delay` samples earlier.
#include "fmrx_internal.hpp"

using namespace fmrx;

struct fmrx_pipeline {
    fmrx_params p{};
    int channels = 1;
    int device = 0;
    size_t max_bytes = 0;
    bool resample = false;
    bool force_generic = false;
    bool profiling = false;
    int prof_every = 1;           // HIP events around every prof_every-th call
    unsigned long seq = 0;        // calls since profiling was enabled
    int Ha = 0;     // audio-stage history, in its input samples
    int delay = 0;  // all-pass delay (stereo), samples
    int Hd = 0;     // history kept in front of demod (>= what the stages need, multiple of 4: keeps demod[0] 16-byte aligned)
    int Hm = 0;     // history kept in front of the stereo mixer output (same rule)
    int St = 0;     // stereo taps
    bool keep_if = false;        // materialise the IF I/Q stream (diagnostics / read_tap)
    bool prev_override = false;  // the next block takes IF[-1] from prev_iq (after set_state)
    bool if_valid = false;       // ifb holds the last block's IF samples
    bool demod_valid = true;     // the demod buffer holds the whole last block (not just its tail: fused mono kernel)
    bool mixer_valid = false;    // the mixer output of the last block was stored (the fused stereo kernel keeps it on chip)
    bool pll_warm = false;       // the PLL has seen a block since reset / set_state (its state is a locked one)
    double pll_off = 0.0;        // host copy of the PLL's trigOffset (IF samples since the stream began, as the reference counts them)
    Options opt;                 // copied from the process defaults at creation; fmrx_pipeline_set_option

    hipStream_t stream = nullptr;  // used by the host-buffer entry point
    FePlan fe;
    AudioPlan audio;
    ResamplePlan rs;           // resampler (modes 2, 3)
    BpfPairPlan bpf_plan;      // stereo + pilot band-pass filters

    DevBuf<uint8_t> in;
    DevBuf<uint8_t> fe_hist[2];
    int fe_cur = 0;
    DevBuf<float> prev_iq[2];
    int prev_cur = 0;
    DevBuf<float> ifb, mono, tmp_hist;
    // discriminator output, two buffers [Hd | n_if] used alternately: the history of a block is the
    // tail of the previous block's buffer, so nothing has to be copied between blocks unless a kernel
    // needs it contiguous in front of its input (materialise_history)
    DevBuf<float> demod_buf[2];
    int demod_last = 0;          // buffer that holds the last processed block
    size_t demod_n_last = 0;     // its length; its history front is valid iff demod_front[demod_last]
    bool demod_front[2] = {true, true};
    DevBuf<float> carrier, bpf, pll, pll_state, pll_scratch, mixer, st_final, left, right;
    // option overlap_calls (stereo, modes 0/1, parallel PLL): the three stages of a call run on internal streams, a call apart
    // each -- front (front end, band-pass pair, the PLL's chunk records), PLL (lanes + repair), output stage -- on the buffer
    // set of the call's parity (set 0 = the buffers above); the caller's stream waits for the output stage's event
    DevBuf<float> carrier1, bpf1, pll1, lti_rec[2];
    hipStream_t ov_stream[3] = {};   // front, PLL, output
    hipEvent_t ov_done[3][2] = {};   // [stage][parity]: the stage of the last call of that parity has finished
    hipEvent_t ov_entry = nullptr;   // the caller's stream at the call: what it did with the previous output is over
    bool ov_ready = false;           // overlap_setup has completed
    int ov_active = 0;               // the regime (option value) of the last call
    int last_set = 0;                // buffer set the last call used (read_tap)
    // state_stereofilt: the mixer output's last Hm samples (index Hm+g holds sample g < 0), written by one call
    // and read by the next: two buffers used alternately, mix_cur = the one the next call reads
    DevBuf<float> mix_tail[2];
    int mix_cur = 0;
    DevBuf<float> out_f32;
    DevBuf<int16_t> out_pcm;
    // asynchronous host-buffer calls (fmrx_pipeline_submit / _wait): two slots, each with its own stream and device staging
    // buffers (slot 0 = `stream`, `in`, `out_f32`, `out_pcm` above): block i+1's copy to the device runs under block i's kernels
    // and block i-1's copy back.  Only the kernels of consecutive blocks are ordered (they carry the state): ev_kern.
    hipStream_t stream2 = nullptr;
    DevBuf<uint8_t> in2;
    DevBuf<float> out_f32_2;
    DevBuf<int16_t> out_pcm_2;
    hipEvent_t ev_kern[2] = {}, ev_done[2] = {};
    bool slot_busy[2] = {false, false};
    int next_slot = 0, oldest_slot = 0, last_kern_slot = -1;
    bool async_ready = false;

    size_t last_n_if = 0, last_n_audio = 0;
    const float *last_mono = nullptr;   // where the last block's mono audio was written
    // profiling: a ring of per-call event quadruples {start, after FE kernel,
    // after audio stage, end}, recorded on the caller's stream
    static constexpr int kRing = 128;
    hipEvent_t ev[kRing][4] = {};
    unsigned long calls = 0;      // profiled calls since profiling was enabled
};

namespace {

__global__ void hist_update_kernel(const uint8_t *__restrict__ old_hist, const uint8_t *__restrict__ x, long n_bytes,
                                   int hb, uint8_t *__restrict__ new_hist)
{
    // new history = last hb bytes of the stream [old_hist | x]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hb) return;
    const long src = static_cast<long>(i) + n_bytes - hb;  // offset into x; negative -> old history
    new_hist[i] = src >= 0 ? x[src] : old_hist[hb + src];
}

// The fused mono kernel gives each wave whole batches of 256 audio samples; below opt.fused_min_audio
// (default 65 536) audio samples per call there are too few batches to fill the chip and the two-kernel
// path is used (0 = always fuse when possible, a huge value = never).

int n_if_of(const fmrx_pipeline *pl, size_t n_bytes) { return static_cast<int>((n_bytes / 2) / pl->p.rf_decim); }

size_t n_audio_of(const fmrx_pipeline *pl, size_t n_bytes)
{
    const size_t n_if = (n_bytes / 2) / pl->p.rf_decim;
    if (pl->resample) return (n_if * static_cast<size_t>(pl->p.audio_upsamp)) / pl->p.audio_decim;
    return n_if / pl->p.audio_decim;
}

// resample_launch's margins hold for [d_x - delay, + n_in) iff it lies that far inside one of the discriminator buffers
// ([Hd | n_if | margin], finite throughout); checked against the allocations, not assumed
bool resample_margins(const fmrx_pipeline *pl, const float *d_x, size_t n_in, int delay)
{
    for (int i = 0; i < 2; i++) {
        const float *b = pl->demod_buf[i].p, *e = b + pl->demod_buf[i].n;
        if (b && d_x - delay - kResampleFront >= b && d_x - delay + n_in + kResampleBack <= e) return true;
    }
    return false;
}

int audio_stage(fmrx_pipeline *pl, const float *d_x, size_t n_in, int delay, float *d_y, hipStream_t s)
{
    if (pl->resample)
        return resample_launch(pl->rs, d_x, n_in, delay, d_y, pl->opt, s, pl->force_generic, false, nullptr, 0,
                               resample_margins(pl, d_x, n_in, delay));
    return audio_fir_launch(pl->audio, d_x, nullptr, n_in, delay, d_y, nullptr, 0, s, pl->force_generic);
}


int reset_state(fmrx_pipeline *pl)
{
    pl->slot_busy[0] = pl->slot_busy[1] = false;   // the device-wide wait below retires whatever was submitted
    pl->next_slot = pl->oldest_slot = 0;
    pl->last_kern_slot = -1;
    // process_dev runs on the caller's stream: whatever is still in flight there must not see the
    // state change under it (reset is rare; a device-wide wait is the simple, safe order)
    FMRX_HIP(hipDeviceSynchronize());
    hipStream_t s = pl->stream;
    for (int i = 0; i < 2; i++) {
        FMRX_TRY(k_fill_u8(pl->fe_hist[i].p, pl->fe.hist_bytes, 128, s));
        FMRX_HIP(hipMemsetAsync(pl->prev_iq[i].p, 0, 2 * sizeof(float), s));
    }
    for (int i = 0; i < 2; i++) FMRX_HIP(hipMemsetAsync(pl->demod_buf[i].p, 0, pl->Hd * sizeof(float), s));
    pl->demod_last = 0;
    pl->demod_n_last = 0;
    pl->demod_front[0] = pl->demod_front[1] = true;
    if (pl->channels == 2) {
        for (int i = 0; i < 2; i++) FMRX_HIP(hipMemsetAsync(pl->mix_tail[i].p, 0, pl->Hm * sizeof(float), s));
        const float init[6] = {0.0f, 0.0f, 1.0f, 0.0f, 1.0f, 0.0f};  // src/project.cpp:458
        FMRX_HIP(hipMemcpyAsync(pl->pll_state.p, init, sizeof(init), hipMemcpyHostToDevice, s));
        FMRX_HIP(hipMemsetAsync(pl->pll_scratch.p + 5, 0, 3 * sizeof(float), s));   // no phase-slope history yet
    }
    FMRX_HIP(hipStreamSynchronize(s));
    pl->fe_cur = pl->prev_cur = pl->mix_cur = 0;
    pl->prev_override = false;
    pl->if_valid = false;
    pl->pll_warm = false;
    pl->pll_off = 0.0;
    return FMRX_OK;
}

int check_block(const fmrx_pipeline *pl, size_t n_bytes)
{
    const fmrx_params &p = pl->p;
    if (n_bytes == 0 || n_bytes % 2) return fail(FMRX_EINVAL, "process: n_bytes %zu must be a positive even number", n_bytes);
    if (n_bytes > pl->max_bytes) return fail(FMRX_EINVAL, "process: block of %zu bytes exceeds max_block_bytes %zu", n_bytes, pl->max_bytes);
    const size_t n = n_bytes / 2;
    if (n % p.rf_decim) return fail(FMRX_EINVAL, "process: %zu samples not a multiple of rf_decim %d", n, p.rf_decim);
    if (n < static_cast<size_t>(p.rf_taps - 1)) return fail(FMRX_EINVAL, "process: %zu samples < rf_taps-1", n);
    const size_t n_if = n / p.rf_decim;
    if (pl->resample) {
        if ((n_if * p.audio_upsamp) % p.audio_decim)
            return fail(FMRX_EINVAL, "process: n_if*upsamp = %zu not a multiple of audio_decim %d", n_if * p.audio_upsamp, p.audio_decim);
    } else if (n_if % p.audio_decim) {
        return fail(FMRX_EINVAL, "process: %zu IF samples not a multiple of audio_decim %d", n_if, p.audio_decim);
    }
    if (n_if < static_cast<size_t>(pl->Ha)) return fail(FMRX_EINVAL, "process: %zu IF samples < audio history %d", n_if, pl->Ha);
    if (pl->channels == 2 && n_if < static_cast<size_t>(pl->St - 1))
        return fail(FMRX_EINVAL, "process: %zu IF samples < stereo_taps-1", n_if);
    return FMRX_OK;
}

}  // namespace

extern "C" {

int fmrx_pipeline_create(fmrx_pipeline **out, const fmrx_params *p, int channels, size_t max_block_bytes, int device)
{
    if (!out || !p) return fail(FMRX_EINVAL, "pipeline_create: null argument");
    if (channels != 1 && channels != 2) return fail(FMRX_EINVAL, "pipeline_create: channels must be 1 or 2");
    if (p->rf_taps < 2 || p->rf_taps > 65535 || p->audio_taps < 2 || p->audio_taps > 65535 || p->rf_decim < 1 ||
        p->audio_decim < 1 || p->audio_upsamp < 0)
        return fail(FMRX_EINVAL, "pipeline_create: bad parameters");
    if (channels == 2 && (p->stereo_taps < 2 || p->stereo_taps > 65535))
        return fail(FMRX_EINVAL, "pipeline_create: bad stereo_taps");
    if (max_block_bytes < 2) return fail(FMRX_EINVAL, "pipeline_create: max_block_bytes too small");
    FMRX_TRY(require_device());
    FMRX_HIP(hipSetDevice(device));

    fmrx_pipeline *pl = new fmrx_pipeline;
    pl->opt = options_snapshot();
    pl->p = *p;
    pl->channels = channels;
    pl->device = device;
    pl->max_bytes = max_block_bytes;
    pl->resample = p->audio_upsamp > 0;
    pl->Ha = pl->resample ? (p->audio_taps - 1) / p->audio_upsamp : p->audio_taps - 1;
    pl->St = channels == 2 ? p->stereo_taps : 0;
    pl->delay = channels == 2 ? (p->stereo_taps - 1) / 2 : 0;
    pl->Hd = pl->Ha + pl->delay;
    if (channels == 2 && pl->St - 1 + 3 > pl->Hd) pl->Hd = pl->St - 1 + 3;   // the band-pass pair kernel reads 16-byte chunks
    if (pl->resample) pl->Hd += kResampleFront;   // the matrix-core resampler stages whole 16-byte pieces around its windows
    pl->Hd = (pl->Hd + 3) / 4 * 4 + 4;   // the specialised audio kernel loads aligned 16-byte chunks
    pl->Hm = (pl->Ha + 3) / 4 * 4 + 4;

    int rc = FMRX_OK;
    auto body = [&]() -> int {
        FMRX_HIP(hipStreamCreateWithFlags(&pl->stream, hipStreamNonBlocking));
        for (auto &q : pl->ev)
            for (auto &e : q) FMRX_HIP(hipEventCreate(&e));
        // coefficients: project.cpp:50 (rf), :321-323 (audio), :172-173 (stereo)
        std::vector<float> h(p->rf_taps);
        design_lpf(static_cast<float>(p->rf_Fs), static_cast<float>(100000), p->rf_taps, h.data());
        FMRX_TRY(fe_plan_init(pl->fe, h.data(), p->rf_taps, p->rf_decim));
        h.resize(p->audio_taps);
        const int design_fs = pl->resample ? p->if_Fs * p->audio_upsamp : p->if_Fs;
        design_lpf(static_cast<float>(design_fs), static_cast<float>(16000), p->audio_taps, h.data());
        if (pl->resample) {
            FMRX_TRY(resample_plan_init(pl->rs, h.data(), p->audio_taps, p->audio_decim, p->audio_upsamp));
        } else {
            FMRX_TRY(audio_plan_init(pl->audio, h.data(), p->audio_taps, p->audio_decim));
        }
        const size_t n_max = max_block_bytes / 2;
        const size_t n_if = n_max / p->rf_decim;
        const size_t n_au = n_audio_of(pl, max_block_bytes) + 1;
        FMRX_TRY(pl->in.alloc(pl->fe.hist_bytes + max_block_bytes + 16));
        for (int i = 0; i < 2; i++) {
            FMRX_TRY(pl->fe_hist[i].alloc(pl->fe.hist_bytes));
            FMRX_TRY(pl->prev_iq[i].alloc(2));
        }
        FMRX_TRY(pl->ifb.alloc(2 * n_if + 16));
        for (int i = 0; i < 2; i++) {   // [Hd | n_if | margin]: all of it finite from the start (resample_launch's margins)
            FMRX_TRY(pl->demod_buf[i].alloc(pl->Hd + n_if + 16 + kResampleBack));
            FMRX_HIP(hipMemset(pl->demod_buf[i].p, 0, (pl->Hd + n_if + 16 + kResampleBack) * sizeof(float)));
        }
        FMRX_TRY(pl->tmp_hist.alloc(pl->Hd + pl->Hm + 16));
        FMRX_TRY(pl->mono.alloc(n_au));
        if (channels == 2) {
            std::vector<float> hc(p->stereo_taps), hs(p->stereo_taps);
            design_bpf(static_cast<float>(p->if_Fs), 18.5e3f, 19.5e3f, p->stereo_taps, hc.data());
            design_bpf(static_cast<float>(p->if_Fs), 22e3f, 54e3f, p->stereo_taps, hs.data());
            FMRX_TRY(bpf_pair_plan_init(pl->bpf_plan, hs.data(), hc.data(), p->stereo_taps));
            FMRX_TRY(pl->carrier.alloc(n_if + 16));
            FMRX_TRY(pl->bpf.alloc(n_if + 16));
            FMRX_TRY(pl->pll.alloc(n_if + 17));
            FMRX_TRY(pl->pll_state.alloc(8));
            FMRX_TRY(pl->pll_scratch.alloc(pll_parallel_scratch_floats(n_if)));
            FMRX_HIP(hipMemset(pl->pll_scratch.p, 0, 8 * sizeof(float)));
            FMRX_TRY(pl->mixer.alloc(pl->Hm + n_if + 16));
            for (int i = 0; i < 2; i++) FMRX_TRY(pl->mix_tail[i].alloc(pl->Hm + 4));
            FMRX_TRY(pl->st_final.alloc(n_au));
            FMRX_TRY(pl->left.alloc(n_au));
            FMRX_TRY(pl->right.alloc(n_au));
        }
        FMRX_TRY(pl->out_f32.alloc(2 * n_au));
        FMRX_TRY(pl->out_pcm.alloc(2 * n_au));
        return reset_state(pl);
    };
    rc = body();
    if (rc != FMRX_OK) {
        fmrx_pipeline_destroy(pl);
        return rc;
    }
    *out = pl;
    return FMRX_OK;
}

int fmrx_pipeline_destroy(fmrx_pipeline *pl)
{
    if (!pl) return FMRX_OK;
    (void)hipSetDevice(pl->device);
    if (pl->stream) {
        (void)hipStreamSynchronize(pl->stream);
        (void)hipStreamDestroy(pl->stream);
    }
    for (auto &q : pl->ev)
        for (auto &e : q)
            if (e) (void)hipEventDestroy(e);
    if (pl->stream2) {
        (void)hipStreamSynchronize(pl->stream2);
        (void)hipStreamDestroy(pl->stream2);
    }
    for (auto &e : pl->ev_kern)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : pl->ev_done)
        if (e) (void)hipEventDestroy(e);
    for (auto &st : pl->ov_stream)
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (auto &q : pl->ov_done)
        for (auto &e : q)
            if (e) (void)hipEventDestroy(e);
    if (pl->ov_entry) (void)hipEventDestroy(pl->ov_entry);
    delete pl;
    return FMRX_OK;
}

int fmrx_pipeline_reset(fmrx_pipeline *pl)
{
    if (!pl) return fail(FMRX_EINVAL, "pipeline_reset: null handle");
    FMRX_HIP(hipSetDevice(pl->device));
    return reset_state(pl);
}

size_t fmrx_pipeline_n_if(const fmrx_pipeline *pl, size_t n_bytes) { return pl ? n_if_of(pl, n_bytes) : 0; }
size_t fmrx_pipeline_n_audio(const fmrx_pipeline *pl, size_t n_bytes) { return pl ? n_audio_of(pl, n_bytes) : 0; }

int fmrx_pipeline_set_profiling(fmrx_pipeline *pl, int on)
{
    if (!pl) return fail(FMRX_EINVAL, "null handle");
    pl->profiling = on != 0;
    pl->prof_every = on > 1 ? on : 1;
    pl->seq = 0;
    pl->calls = 0;
    return FMRX_OK;
}

int fmrx_pipeline_pll_diagnostics(fmrx_pipeline *pl, unsigned *repaired_segments, float *max_dphase, float *max_dinteg)
{
    if (!pl) return fail(FMRX_EINVAL, "null handle");
    if (pl->channels != 2) return fail(FMRX_EINVAL, "pll_diagnostics: not a stereo pipeline");
    FMRX_HIP(hipSetDevice(pl->device));
    FMRX_HIP(hipDeviceSynchronize());
    unsigned hdr[8];
    FMRX_HIP(hipMemcpy(hdr, pl->pll_scratch.p, sizeof(hdr), hipMemcpyDeviceToHost));
    if (repaired_segments) *repaired_segments = hdr[2];
    if (max_dphase) std::memcpy(max_dphase, &hdr[3], sizeof(float));
    if (max_dinteg) std::memcpy(max_dinteg, &hdr[4], sizeof(float));
    return FMRX_OK;
}

int fmrx_pipeline_set_option(fmrx_pipeline *pl, const char *name, long value)
{
    if (!pl) return fail(FMRX_EINVAL, "null handle");
    return set_option_in(pl->opt, name, value);
}

int fmrx_pipeline_set_keep_intermediates(fmrx_pipeline *pl, int on)
{
    if (!pl) return fail(FMRX_EINVAL, "null handle");
    pl->keep_if = on != 0;
    return FMRX_OK;
}

int fmrx_pipeline_set_force_generic(fmrx_pipeline *pl, int on)
{
    if (!pl) return fail(FMRX_EINVAL, "null handle");
    pl->force_generic = on != 0;
    return FMRX_OK;
}

// internal streams, events and the second buffer set of option overlap_calls, on first use
static int overlap_setup(fmrx_pipeline *pl)
{
    if (pl->ov_ready) return FMRX_OK;
    // every handle is created only where it is still missing: a call that failed half way (out of memory, say) is simply
    // continued by the next one, nothing is created twice and nothing leaks (destroy frees whatever exists)
    const size_t n_if = (pl->max_bytes / 2) / pl->p.rf_decim;
    FMRX_TRY(pl->carrier1.ensure(pl->carrier.n));
    FMRX_TRY(pl->bpf1.ensure(pl->bpf.n));
    FMRX_TRY(pl->pll1.ensure(pl->pll.n));
    for (auto &b : pl->lti_rec) FMRX_TRY(b.ensure(pll_parallel_lti_floats(n_if) + 2));
    for (auto &q : pl->ov_done)
        for (auto &e : q)
            if (!e) FMRX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!pl->ov_entry) FMRX_HIP(hipEventCreateWithFlags(&pl->ov_entry, hipEventDisableTiming));
    for (auto &st : pl->ov_stream)
        if (!st) FMRX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    pl->ov_ready = true;
    return FMRX_OK;
}

int fmrx_pipeline_process_dev(fmrx_pipeline *pl, const uint8_t *d_iq, size_t n_bytes, float *d_audio_f32,
                              int16_t *d_pcm16, int pcm_policy, void *stream)
{
    if (!pl || !d_iq) return fail(FMRX_EINVAL, "process_dev: null argument");
    FMRX_TRY(check_block(pl, n_bytes));
    FMRX_HIP(hipSetDevice(pl->device));   // the handle's buffers live there, whatever the caller's current device
    hipStream_t s = static_cast<hipStream_t>(stream);
    const fmrx_params &p = pl->p;
    const size_t n = n_bytes / 2;
    const size_t n_if = n / p.rf_decim;
    const size_t n_au = n_audio_of(pl, n_bytes);
    pl->last_n_if = n_if;
    pl->last_n_audio = n_au;
    // this block's discriminator output goes to the buffer the previous block did not use
    const int last = pl->demod_last, cur = last ^ 1;
    float *dbuf = pl->demod_buf[cur].p;
    float *demod = dbuf + pl->Hd;
    const float *hist_end = pl->demod_buf[last].p + pl->Hd + pl->demod_n_last;   // one past the previous block's last sample
    pl->demod_front[cur] = false;
    auto materialise_history = [&]() -> int {   // for kernels that index history at negative offsets of their input
        FMRX_HIP(hipMemcpyAsync(dbuf, hist_end - pl->Hd, pl->Hd * sizeof(float), hipMemcpyDeviceToDevice, s));
        pl->demod_front[cur] = true;
        return FMRX_OK;
    };

    // events around every prof_every-th call: a record costs ~5 us of stream time
    const bool prof = pl->profiling && (pl->seq++ % static_cast<unsigned long>(pl->prof_every) == 0);
    hipEvent_t *ev = pl->ev[pl->calls % fmrx_pipeline::kRing];
    if (prof) FMRX_HIP(hipEventRecord(ev[0], s));

    // ---- RF_FrontEnd: project.cpp:82-128 (u8 -> IF I/Q -> discriminator) ----
    const uint8_t *hist = pl->fe_hist[pl->fe_cur].p;
    uint8_t *hist_next = pl->fe_hist[pl->fe_cur ^ 1].p;
    const int hb = pl->fe.hist_bytes;
    const float *prev = pl->prev_iq[pl->prev_cur].p;
    float *prev_next = pl->prev_iq[pl->prev_cur ^ 1].p;
    bool hist_done = false;
    pl->demod_valid = true;
    const bool mfma = pl->opt.fe_variant == 0;
    // option demod = 1: the model's arctangent demodulator on the IF stream (materialised for it), instead of the front-end kernels' own discriminator
    const bool arctan = pl->opt.demod == 1;
    const bool keep_if = pl->keep_if || arctan;
    // option overlap_calls: front | PLL | output stage of the stereo chain of consecutive calls on internal streams (see the struct)
    const bool ovl = pl->opt.overlap_calls != 0 && pl->channels == 2 && !pl->resample && !pl->force_generic && !keep_if &&
                     !pl->profiling && pl->opt.pll_mode == 0 && mfma && n_if >= static_cast<size_t>(pl->Hd) &&
                     fe_mfma_available(pl->fe, d_iq, n, hist) && stereo_out_available(p.audio_taps, p.audio_decim);
    const int regime = ovl ? pl->opt.overlap_calls : 0;
    if (regime != pl->ov_active) {   // change of regime between calls: everything in flight first
        FMRX_HIP(hipDeviceSynchronize());
        pl->ov_active = regime;
    }
    if (ovl) FMRX_TRY(overlap_setup(pl));
    // overlap_calls 1: the front on an internal stream, PLL and output stage on the caller's (fewest event operations: at these
    // step sizes the host's enqueue rate is the next limit); 2: all three on internal streams
    const bool ovl3 = ovl && pl->opt.overlap_calls >= 2;
    hipStream_t sf = ovl ? pl->ov_stream[0] : s, sp = ovl3 ? pl->ov_stream[1] : s, so = ovl3 ? pl->ov_stream[2] : s;
    const int bs = ovl ? cur : 0;   // buffer set of the stereo intermediates
    pl->last_set = bs;
    float *carrier_b = bs ? pl->carrier1.p : pl->carrier.p, *bpf_b = bs ? pl->bpf1.p : pl->bpf.p, *pll_b = bs ? pl->pll1.p : pl->pll.p;
    if (ovl) {
        // the front writes this parity's buffers: the last call of the same parity must be through with them; the output
        // stage writes the caller's buffers: whatever the caller's stream still does with them comes first (the input is the
        // one thing the option vouches for)
        FMRX_HIP(hipStreamWaitEvent(sf, pl->ov_done[2][cur], 0));
        if (so != s) {
            FMRX_HIP(hipEventRecord(pl->ov_entry, s));
            FMRX_HIP(hipStreamWaitEvent(so, pl->ov_entry, 0));
        }
    }
    if (pl->channels == 1 && !pl->resample && !pl->force_generic && !keep_if && mfma &&
        n_if >= static_cast<size_t>(pl->Hd) && static_cast<long>(n_au) >= pl->opt.fused_min_audio &&
        mono_fused_available(pl->fe, pl->audio, d_iq, n, hist)) {
        // ---- RF_FrontEnd + RF_MONO of modes 0/1 in one kernel (kernels_fe_mfma.hip): the discriminator
        //      output stays on chip; only its tail (state_mono) is written for the next block ----
        hist_done = n_bytes >= static_cast<size_t>(hb);
        // f32 audio is written only where somebody will read it: the caller's buffer, or the handle's own when
        // no PCM was asked for either (read_tap); PCM-only callers get the reference's output format and nothing else
        float *dst = d_audio_f32 ? d_audio_f32 : (d_pcm16 ? nullptr : pl->mono.p);
        FMRX_TRY(mono_fused_launch(pl->fe, pl->audio, d_iq, n, hist, prev, hist_end, demod, pl->Hd, prev_next, dst, d_pcm16,
                                   pcm_policy, hist_done ? hist_next : nullptr, pl->opt, s));
        pl->if_valid = false;
        pl->demod_valid = false;
        pl->last_mono = dst;
        pl->prev_cur ^= 1;
        pl->prev_override = false;
        pl->fe_cur ^= 1;
        if (prof) FMRX_HIP(hipEventRecord(ev[1], s));
        if (!hist_done)
            hipLaunchKernelGGL(hist_update_kernel, dim3((hb + 255) / 256), dim3(256), 0, s, hist, d_iq,
                               static_cast<long>(n_bytes), hb, hist_next);
        pl->demod_last = cur;
        pl->demod_n_last = n_if;
        if (prof) {
            FMRX_HIP(hipEventRecord(ev[2], s));
            FMRX_HIP(hipEventRecord(ev[3], s));
            pl->calls++;
        }
        return FMRX_OK;
    }
    if (!pl->force_generic && mfma && fe_mfma_available(pl->fe, d_iq, n, hist)) {
        // matrix-core kernel: int8 MFMA FIR + discriminator, HBM-bound (kernels_fe_mfma.hip)
        hist_done = n_bytes >= static_cast<size_t>(hb);
        // everything but mono modes 0/1 reads the discriminator history at negative indices of this block's buffer:
        // the kernel copies it there itself (it is the tail of the previous block's buffer)
        const bool want_front = !(pl->channels == 1 && !pl->resample) && n_if >= static_cast<size_t>(pl->Hd);
        FMRX_TRY(fe_mfma_launch(pl->fe, d_iq, n, hist, pl->prev_override ? prev : nullptr, demod,
                                keep_if ? pl->ifb.p : nullptr, prev_next,
                                hist_done ? hist_next : nullptr, pl->opt, sf, want_front ? hist_end - pl->Hd : nullptr,
                                want_front ? dbuf : nullptr, pl->Hd));
        if (want_front) pl->demod_front[cur] = true;
        pl->if_valid = keep_if;
    } else if (!pl->force_generic && fe_fused_available(pl->fe, d_iq, n)) {
        // one kernel; the IF stream is written only when somebody asked to look at it, and the kernel
        // also leaves the stream's last bytes (I_state/Q_state, filter.cpp:182-187) for the next block
        hist_done = n_bytes >= static_cast<size_t>(hb);
        FMRX_TRY(fe_demod_launch(pl->fe, d_iq, n, hist, pl->prev_override ? prev : nullptr, demod,
                                 keep_if ? pl->ifb.p : nullptr, prev_next, hist_done ? hist_next : nullptr, pl->opt, s));
        pl->if_valid = keep_if;
    } else {
        FMRX_TRY(fe_launch(pl->fe, d_iq, n, hist, pl->ifb.p, pl->opt, s, pl->force_generic));
        FMRX_TRY(k_fm_demod_if(pl->ifb.p, n_if, prev, prev_next, demod, 0, s));
        pl->if_valid = true;
    }
    // fmDemodArctan (model/fmSupportLib.py:502-531) over the IF stream, IF[-1] = the carried prev_i / prev_q (their phase is the
    // model's state_phase modulo 2 pi; zeros at the start of a stream: atan2(0, 0) = 0 = the model's initial phase)
    if (arctan) FMRX_TRY(k_fm_demod_arctan_if(pl->ifb.p, n_if, prev, demod, s));
    pl->prev_cur ^= 1;
    pl->prev_override = false;
    pl->fe_cur ^= 1;
    if (prof) FMRX_HIP(hipEventRecord(ev[1], s));
    if (!hist_done)
        hipLaunchKernelGGL(hist_update_kernel, dim3((hb + 255) / 256), dim3(256), 0, sf, hist, d_iq,
                           static_cast<long>(n_bytes), hb, hist_next);
    // a block shorter than the history keeps its own front valid, so that "tail of the previous
    // buffer" stays a contiguous Hd samples for whoever comes next
    if (n_if < static_cast<size_t>(pl->Hd)) FMRX_TRY(materialise_history());
    pl->demod_last = cur;
    pl->demod_n_last = n_if;

    if (pl->channels == 1 && !pl->resample) {
        // ---- RF_MONO, modes 0/1: audio FIR + decimate + PCM in one kernel (project.cpp:346;
        //      threadMonoOnly.cpp:185-191), straight into the caller's buffers ----
        float *dst = d_audio_f32 ? d_audio_f32 : pl->mono.p;
        const bool fast = audio_fast_available(pl->audio, demod) && !pl->force_generic;
        if (!fast && !pl->demod_front[cur]) FMRX_TRY(materialise_history());
        FMRX_TRY(audio_fir_launch(pl->audio, demod, fast ? hist_end : nullptr, n_if, 0, dst, d_pcm16, pcm_policy, s,
                                  pl->force_generic));
        pl->last_mono = dst;
        if (prof) {
            FMRX_HIP(hipEventRecord(ev[2], s));
            FMRX_HIP(hipEventRecord(ev[3], s));
            pl->calls++;
        }
        return FMRX_OK;
    }

    if (!pl->demod_front[cur]) FMRX_TRY(materialise_history());
    float *out_l = pl->mono.p, *out_r = nullptr;
    pl->last_mono = pl->mono.p;
    if (pl->channels == 1) {
        // ---- RF_MONO, modes 2/3: rational resampler (project.cpp:353) ----
        if (!pl->force_generic && resample_mfma_available(pl->rs, demod, n_if, 0, pl->opt)) {
            // the matrix-core kernel writes the caller's buffers itself; PCM-only callers get the reference's output
            // format and nothing else (as the fused mono kernel of modes 0/1)
            float *dst = d_audio_f32 ? d_audio_f32 : (d_pcm16 ? nullptr : pl->mono.p);
            FMRX_TRY(resample_launch(pl->rs, demod, n_if, 0, dst, pl->opt, s, false, false, d_pcm16, pcm_policy,
                                     resample_margins(pl, demod, n_if, 0)));
            pl->last_mono = dst;
            if (prof) {
                FMRX_HIP(hipEventRecord(ev[2], s));
                FMRX_HIP(hipEventRecord(ev[3], s));
                pl->calls++;
            }
            return FMRX_OK;
        }
        FMRX_TRY(audio_stage(pl, demod, n_if, 0, pl->mono.p, s));
        if (prof) FMRX_HIP(hipEventRecord(ev[2], s));
    } else {
        // ---- RF_STEREO: project.cpp:194-280 ----
        const bool fused_out = !pl->resample && !pl->force_generic && stereo_out_available(p.audio_taps, p.audio_decim);
        float *mixer = pl->mixer.p + pl->Hm;
        if (!fused_out) {
            FMRX_TRY(audio_stage(pl, demod, n_if, pl->delay, pl->mono.p, s));  // all-pass = index offset
        }
        if (prof) FMRX_HIP(hipEventRecord(ev[2], s));
        FMRX_TRY(bpf_pair_launch(pl->bpf_plan, demod, n_if, bpf_b, carrier_b, sf, pl->force_generic));
        if (pl->force_generic || pl->opt.pll_mode != 0) {
            // the serial recurrence: glibc's functions in the bit-exact mode (and pll_mode 2), fast math for pll_mode 1
            const int fast = !pl->force_generic && pl->opt.pll_mode == 1;
            FMRX_TRY(k_fm_pll(carrier_b, n_if, pll_b, pl->pll_state.p, 19e3f, static_cast<float>(p.if_Fs), 2.0f,
                              0.0f, 0.01f, fast, s));
        } else {
            // a stream's first block starts unlocked: walk its first samples serially so that the
            // segment lanes extrapolate from a locked state; later blocks start locked already
            size_t head = 0;
            if (!pl->pll_warm) {
                const size_t head_len = pl->opt.pll_head >= 0 ? static_cast<size_t>(pl->opt.pll_head) : kPllHead;
                head = n_if < head_len ? n_if : head_len;
            }
            // what depends on the input alone (the linear system's chunk records) belongs to the front
            float *lti = ovl ? pl->lti_rec[cur].p : nullptr;
            if (ovl && n_if > head)
                FMRX_TRY(k_fm_pll_parallel(carrier_b + head, n_if - head, pll_b + head, pl->pll_state.p, 19e3f,
                                           static_cast<float>(p.if_Fs), 2.0f, 0.0f, 0.01f, pl->pll_scratch.p, pl->opt, sf,
                                           pl->pll_off + static_cast<double>(head), 1, lti));
            if (ovl) {   // read-after-write: the lanes read this call's carrier / chunk records, which the front wrote on its own stream
                FMRX_HIP(hipEventRecord(pl->ov_done[0][cur], sf));
                FMRX_HIP(hipStreamWaitEvent(sp, pl->ov_done[0][cur], 0));
            }
            if (head > 0)
                FMRX_TRY(k_fm_pll(carrier_b, head, pll_b, pl->pll_state.p, 19e3f, static_cast<float>(p.if_Fs),
                                  2.0f, 0.0f, 0.01f, 1, sp));
            if (n_if > head)
                FMRX_TRY(k_fm_pll_parallel(carrier_b + head, n_if - head, pll_b + head, pl->pll_state.p, 19e3f,
                                           static_cast<float>(p.if_Fs), 2.0f, 0.0f, 0.01f, pl->pll_scratch.p, pl->opt, sp,
                                           pl->pll_off + static_cast<double>(head), ovl ? 2 : 3, lti));
            pl->pll_warm = true;
            if (so != sp) {   // read-after-write: the output stage reads the NCO values and the PLL state the lanes / repair left
                FMRX_HIP(hipEventRecord(pl->ov_done[1][cur], sp));
                FMRX_HIP(hipStreamWaitEvent(so, pl->ov_done[1][cur], 0));
            }
        }
        pl->pll_off += static_cast<double>(n_if);
        const float *tail_in = pl->mix_tail[pl->mix_cur].p;
        float *tail_out = pl->mix_tail[pl->mix_cur ^ 1].p;
        if (fused_out) {
            // mixer, both audio FIRs, L/R combine and PCM in one kernel, straight into the caller's buffers
            FMRX_TRY(stereo_out_launch(demod, bpf_b, pll_b, tail_in, tail_out, pl->Hm, n_if, pl->delay, pl->audio.h.p,
                                       p.audio_taps, p.audio_decim, pl->mono.p, pl->st_final.p,
                                       d_audio_f32 ? d_audio_f32 : (d_pcm16 ? nullptr : pl->left.p),
                                       d_audio_f32 ? d_audio_f32 + n_au : (d_pcm16 ? nullptr : pl->right.p), d_pcm16,
                                       pcm_policy, pl->keep_if ? mixer : nullptr, so));
            if (ovl) {   // the caller's stream sees the call's output in its own order, as always
                FMRX_HIP(hipEventRecord(pl->ov_done[2][cur], so));
                if (so != s) FMRX_HIP(hipStreamWaitEvent(s, pl->ov_done[2][cur], 0));
            }
            pl->mixer_valid = pl->keep_if;
            pl->mix_cur ^= 1;
            if (prof) {
                FMRX_HIP(hipEventRecord(ev[3], s));
                pl->calls++;
            }
            return FMRX_OK;
        }
        FMRX_HIP(hipMemcpyAsync(pl->mixer.p, tail_in, pl->Hm * sizeof(float), hipMemcpyDeviceToDevice, s));
        FMRX_TRY(k_mix(bpf_b, pll_b, n_if, mixer, s));
        FMRX_TRY(audio_stage(pl, mixer, n_if, 0, pl->st_final.p, s));
        FMRX_TRY(k_combine(pl->st_final.p, pl->mono.p, n_au, pl->left.p, pl->right.p, s));
        // the mixer output's last Hm samples (of [history | block]) are the next block's state_stereofilt
        FMRX_HIP(hipMemcpyAsync(tail_out, pl->mixer.p + n_if, pl->Hm * sizeof(float), hipMemcpyDeviceToDevice, s));
        pl->mix_cur ^= 1;
        pl->mixer_valid = true;
        out_l = pl->left.p;
        out_r = pl->right.p;
    }

    // ---- outputs ----
    if (d_audio_f32) {
        FMRX_HIP(hipMemcpyAsync(d_audio_f32, out_l, n_au * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (out_r) FMRX_HIP(hipMemcpyAsync(d_audio_f32 + n_au, out_r, n_au * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (d_pcm16) {
        if (out_r) FMRX_TRY(k_pcm16_stereo(out_l, out_r, n_au, d_pcm16, pcm_policy, s));
        else FMRX_TRY(k_pcm16(out_l, n_au, d_pcm16, pcm_policy, s));
    }
    if (prof) {
        FMRX_HIP(hipEventRecord(ev[3], s));
        pl->calls++;
    }
    return FMRX_OK;
}

static int async_setup(fmrx_pipeline *pl)
{
    if (pl->async_ready) return FMRX_OK;
    const size_t n_au = n_audio_of(pl, pl->max_bytes) + 1;
    FMRX_TRY(pl->in2.ensure(pl->in.n));
    FMRX_TRY(pl->out_f32_2.ensure(2 * n_au));
    FMRX_TRY(pl->out_pcm_2.ensure(2 * n_au));
    if (!pl->stream2) FMRX_HIP(hipStreamCreateWithFlags(&pl->stream2, hipStreamNonBlocking));
    for (auto &e : pl->ev_kern)
        if (!e) FMRX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : pl->ev_done)
        if (!e) FMRX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    pl->async_ready = true;
    return FMRX_OK;
}

int fmrx_pipeline_wait(fmrx_pipeline *pl)
{
    if (!pl) return fail(FMRX_EINVAL, "wait: null handle");
    const int slot = pl->oldest_slot;
    if (!pl->slot_busy[slot]) return FMRX_OK;
    FMRX_HIP(hipSetDevice(pl->device));
    FMRX_HIP(hipEventSynchronize(pl->ev_done[slot]));
    pl->slot_busy[slot] = false;
    pl->oldest_slot = slot ^ 1;
    return FMRX_OK;
}

int fmrx_pipeline_submit(fmrx_pipeline *pl, const uint8_t *iq, size_t n_bytes, float *audio_f32, int16_t *pcm16, int pcm_policy)
{
    if (!pl || !iq) return fail(FMRX_EINVAL, "submit: null argument");
    FMRX_TRY(check_block(pl, n_bytes));
    FMRX_HIP(hipSetDevice(pl->device));
    FMRX_TRY(async_setup(pl));
    const int slot = pl->next_slot;
    if (pl->slot_busy[slot]) FMRX_TRY(fmrx_pipeline_wait(pl));   // two blocks in flight: the older one first
    hipStream_t s = slot ? pl->stream2 : pl->stream;
    uint8_t *d_in = slot ? pl->in2.p : pl->in.p;
    float *d_f32 = slot ? pl->out_f32_2.p : pl->out_f32.p;
    int16_t *d_pcm = slot ? pl->out_pcm_2.p : pl->out_pcm.p;
    const size_t n_au = n_audio_of(pl, n_bytes);
    const size_t nch = pl->channels;
    FMRX_HIP(hipMemcpyAsync(d_in, iq, n_bytes, hipMemcpyHostToDevice, s));
    // option overlap_calls vouches for inputs that are complete at the call; this one is still on its way: wait for it
    if (pl->opt.overlap_calls != 0) FMRX_HIP(hipStreamSynchronize(s));
    // the kernels of consecutive blocks carry the receiver's state: this block's follow the previous block's (the copies do not)
    if (pl->last_kern_slot >= 0 && pl->last_kern_slot != slot) FMRX_HIP(hipStreamWaitEvent(s, pl->ev_kern[pl->last_kern_slot], 0));
    FMRX_TRY(fmrx_pipeline_process_dev(pl, d_in, n_bytes, audio_f32 ? d_f32 : nullptr, pcm16 ? d_pcm : nullptr, pcm_policy, s));
    FMRX_HIP(hipEventRecord(pl->ev_kern[slot], s));
    pl->last_kern_slot = slot;
    if (audio_f32) FMRX_HIP(hipMemcpyAsync(audio_f32, d_f32, nch * n_au * sizeof(float), hipMemcpyDeviceToHost, s));
    if (pcm16) FMRX_HIP(hipMemcpyAsync(pcm16, d_pcm, nch * n_au * sizeof(int16_t), hipMemcpyDeviceToHost, s));
    FMRX_HIP(hipEventRecord(pl->ev_done[slot], s));
    pl->slot_busy[slot] = true;
    pl->next_slot = slot ^ 1;
    if (!pl->slot_busy[slot ^ 1]) pl->oldest_slot = slot;
    return FMRX_OK;
}

int fmrx_pipeline_process(fmrx_pipeline *pl, const uint8_t *iq, size_t n_bytes, float *audio_f32, int16_t *pcm16,
                          int pcm_policy)
{
    if (!pl || !iq) return fail(FMRX_EINVAL, "process: null argument");
    // the synchronous form = submit + wait for everything in flight (so mixing the two forms keeps the stream's order)
    FMRX_TRY(fmrx_pipeline_submit(pl, iq, n_bytes, audio_f32, pcm16, pcm_policy));
    FMRX_TRY(fmrx_pipeline_wait(pl));
    return fmrx_pipeline_wait(pl);
}

int fmrx_pipeline_read_tap(fmrx_pipeline *pl, int which, float *out, size_t *n)
{
    if (!pl || !n) return fail(FMRX_EINVAL, "read_tap: null argument");
    FMRX_HIP(hipSetDevice(pl->device));
    FMRX_HIP(hipDeviceSynchronize());
    const size_t n_if = pl->last_n_if, n_au = pl->last_n_audio;
    const float *src = nullptr;
    size_t cnt = 0;
    const bool st = pl->channels == 2;
    switch (which) {
    case FMRX_TAP_IF_I:
    case FMRX_TAP_IF_Q: cnt = n_if; break;
    case FMRX_TAP_DEMOD:
        // the block was shifted into the history position by carry_history: it is the
        // last n_if samples of [Hd | n_if] only until the next block; read it from the
        // tail copy kept in front when the block is shorter than that -- simplest: the
        // block region itself is still intact (carry copies, it does not move)
        src = pl->demod_buf[pl->demod_last].p + pl->Hd; cnt = n_if; break;
    case FMRX_TAP_MONO:
        src = pl->last_mono; cnt = n_au;
        if (!src && out && cnt) return fail(FMRX_EINVAL, "read_tap: the last call wrote s16 PCM only (no f32 audio buffer was passed)");
        break;
    case FMRX_TAP_CARRIER: if (st) { src = pl->last_set ? pl->carrier1.p : pl->carrier.p; cnt = n_if; } break;
    case FMRX_TAP_STEREO_BPF: if (st) { src = pl->last_set ? pl->bpf1.p : pl->bpf.p; cnt = n_if; } break;
    case FMRX_TAP_PLL: if (st) { src = pl->last_set ? pl->pll1.p : pl->pll.p; cnt = n_if + 1; } break;
    case FMRX_TAP_MIXER:
        if (st) { src = pl->mixer.p + pl->Hm; cnt = n_if; }
        if (st && out && cnt && !pl->mixer_valid)
            return fail(FMRX_EINVAL, "read_tap: the fused stereo kernel keeps the mixer output on chip; call "
                                     "fmrx_pipeline_set_keep_intermediates(pl, 1) before processing");
        break;
    case FMRX_TAP_STEREO_FINAL: if (st) { src = pl->st_final.p; cnt = n_au; } break;
    default: return fail(FMRX_EINVAL, "read_tap: unknown tap %d", which);
    }
    if (which >= FMRX_TAP_CARRIER && !st) return fail(FMRX_EINVAL, "read_tap: tap %d exists only in stereo pipelines", which);
    *n = cnt;
    if (!out || cnt == 0) return FMRX_OK;
    if (which == FMRX_TAP_DEMOD && !pl->demod_valid)
        return fail(FMRX_EINVAL, "read_tap: the fused mono kernel keeps the discriminator output on chip; "
                                 "call fmrx_pipeline_set_keep_intermediates(pl, 1) before processing");
    if ((which == FMRX_TAP_IF_I || which == FMRX_TAP_IF_Q) && !pl->if_valid)
        return fail(FMRX_EINVAL, "read_tap: the IF stream is not materialised by the fused front end; call "
                                 "fmrx_pipeline_set_keep_intermediates(pl, 1) before processing");
    if (which == FMRX_TAP_IF_I || which == FMRX_TAP_IF_Q) {
        // NOTE: demod of the block already shifted the history; IF is intact
        std::vector<float> z(2 * cnt);
        FMRX_HIP(hipMemcpy(z.data(), pl->ifb.p, 2 * cnt * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < cnt; k++) out[k] = z[2 * k + (which == FMRX_TAP_IF_Q)];
        return FMRX_OK;
    }
    FMRX_HIP(hipMemcpy(out, src, cnt * sizeof(float), hipMemcpyDeviceToHost));
    return FMRX_OK;
}

// ---- carried state -----------------------------------------------------------------------
size_t fmrx_pipeline_state_size(const fmrx_pipeline *pl)
{
    if (!pl) return 0;
    size_t n = 2 * (pl->p.rf_taps - 1) + 2 + pl->Ha;
    if (pl->channels == 2) n += 2 * (pl->St - 1) + pl->Ha + pl->delay + 6;
    return n;
}

int fmrx_pipeline_get_state(fmrx_pipeline *pl, float *state, size_t n)
{
    if (!pl || !state) return fail(FMRX_EINVAL, "get_state: null argument");
    if (n != fmrx_pipeline_state_size(pl)) return fail(FMRX_EINVAL, "get_state: expected %zu floats", fmrx_pipeline_state_size(pl));
    FMRX_HIP(hipSetDevice(pl->device));
    FMRX_HIP(hipDeviceSynchronize());
    const int T = pl->p.rf_taps, hb = pl->fe.hist_bytes, live = 2 * (T - 1);
    std::vector<uint8_t> hist(hb);
    FMRX_HIP(hipMemcpy(hist.data(), pl->fe_hist[pl->fe_cur].p, hb, hipMemcpyDeviceToHost));
    float *o = state;
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < T - 1; i++) *o++ = (static_cast<int>(hist[hb - live + 2 * i + c]) - 128) / 128.0f;
    FMRX_HIP(hipMemcpy(o, pl->prev_iq[pl->prev_cur].p, 2 * sizeof(float), hipMemcpyDeviceToHost));
    o += 2;
    std::vector<float> dh(pl->Hd);
    // the carried demod history = the last Hd samples of [front | block] of the buffer used last
    FMRX_HIP(hipMemcpy(dh.data(), pl->demod_buf[pl->demod_last].p + pl->demod_n_last, pl->Hd * sizeof(float),
                       hipMemcpyDeviceToHost));
    // the demod history is shared by every consumer of demod; each reference
    // state vector is a window of it
    const float *dend = dh.data() + pl->Hd;  // one past demod[-1]
    // state_mono: audio FIR input is the all-passed demod: samples demod[-delay-Ha .. -delay-1]
    std::memcpy(o, dend - pl->delay - pl->Ha, pl->Ha * sizeof(float));
    o += pl->Ha;
    if (pl->channels == 2) {
        std::memcpy(o, dend - (pl->St - 1), (pl->St - 1) * sizeof(float)); o += pl->St - 1;  // state_stereo
        std::memcpy(o, dend - (pl->St - 1), (pl->St - 1) * sizeof(float)); o += pl->St - 1;  // state_carrier
        FMRX_HIP(hipMemcpy(o, pl->mix_tail[pl->mix_cur].p + (pl->Hm - pl->Ha), pl->Ha * sizeof(float), hipMemcpyDeviceToHost)); o += pl->Ha;  // state_stereofilt
        std::memcpy(o, dend - pl->delay, pl->delay * sizeof(float)); o += pl->delay;        // state_allpass
        FMRX_HIP(hipMemcpy(o, pl->pll_state.p, 6 * sizeof(float), hipMemcpyDeviceToHost)); o += 6;
    }
    return FMRX_OK;
}

int fmrx_pipeline_set_state(fmrx_pipeline *pl, const float *state, size_t n)
{
    if (!pl || !state) return fail(FMRX_EINVAL, "set_state: null argument");
    if (n != fmrx_pipeline_state_size(pl)) return fail(FMRX_EINVAL, "set_state: expected %zu floats", fmrx_pipeline_state_size(pl));
    FMRX_HIP(hipSetDevice(pl->device));
    FMRX_HIP(hipDeviceSynchronize());
    const int T = pl->p.rf_taps, hb = pl->fe.hist_bytes, live = 2 * (T - 1);
    std::vector<uint8_t> hist(hb, 128);
    const float *o = state;
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < T - 1; i++) {
            const float v = *o++ * 128.0f + 128.0f;
            if (!(v >= 0.0f && v <= 255.0f) || v != static_cast<float>(static_cast<int>(v)))
                return fail(FMRX_EINVAL, "set_state: front-end state holds a value that is not (u8-128)/128");
            hist[hb - live + 2 * i + c] = static_cast<uint8_t>(v);
        }
    FMRX_HIP(hipMemcpy(pl->fe_hist[pl->fe_cur].p, hist.data(), hb, hipMemcpyHostToDevice));
    FMRX_HIP(hipMemcpy(pl->prev_iq[pl->prev_cur].p, o, 2 * sizeof(float), hipMemcpyHostToDevice));
    o += 2;
    std::vector<float> dh(pl->Hd, 0.0f);
    float *dend = dh.data() + pl->Hd;
    const float *s_mono = o; o += pl->Ha;
    if (pl->channels == 2) {
        const float *s_st = o; o += pl->St - 1;
        const float *s_car = o; o += pl->St - 1;
        const float *s_sf = o; o += pl->Ha;
        const float *s_ap = o; o += pl->delay;
        // widest window first, then the ones that must agree with it
        std::memcpy(dend - pl->delay - pl->Ha, s_mono, pl->Ha * sizeof(float));
        std::memcpy(dend - pl->delay, s_ap, pl->delay * sizeof(float));
        std::memcpy(dend - (pl->St - 1), s_st, (pl->St - 1) * sizeof(float));
        (void)s_car;  // identical to state_stereo by construction (same input stream)
        FMRX_HIP(hipMemset(pl->mix_tail[pl->mix_cur].p, 0, pl->Hm * sizeof(float)));
        FMRX_HIP(hipMemcpy(pl->mix_tail[pl->mix_cur].p + (pl->Hm - pl->Ha), s_sf, pl->Ha * sizeof(float), hipMemcpyHostToDevice));
        FMRX_HIP(hipMemcpy(pl->pll_state.p, o, 6 * sizeof(float), hipMemcpyHostToDevice));
        pl->pll_off = static_cast<double>(o[5]);                   // trigOffset (src/filter.cpp:37)
        o += 6;
    } else {
        std::memcpy(dend - pl->Ha, s_mono, pl->Ha * sizeof(float));
    }
    FMRX_HIP(hipMemcpy(pl->demod_buf[0].p, dh.data(), pl->Hd * sizeof(float), hipMemcpyHostToDevice));
    pl->demod_last = 0;
    pl->demod_n_last = 0;
    pl->demod_front[0] = true;
    pl->prev_override = true;   // the fused front end would otherwise recompute IF[-1] from the byte history
    pl->pll_warm = false;
    if (pl->channels == 2) FMRX_HIP(hipMemset(pl->pll_scratch.p + 5, 0, 3 * sizeof(float)));
    return FMRX_OK;
}

int fmrx_pipeline_last_timing(fmrx_pipeline *pl, float *t)
{
    int count = 0;
    FMRX_TRY(fmrx_pipeline_timing_sum(pl, t, &count, 1));
    return FMRX_OK;
}

int fmrx_pipeline_timing_sum(fmrx_pipeline *pl, float *t, int *count, int max_calls)
{
    if (!pl || !t || !count) return fail(FMRX_EINVAL, "timing_sum: null argument");
    if (pl->calls == 0) return fail(FMRX_EINVAL, "timing: no profiled process call (enable with fmrx_pipeline_set_profiling)");
    unsigned long n = pl->calls < static_cast<unsigned long>(fmrx_pipeline::kRing) ? pl->calls : fmrx_pipeline::kRing;
    if (max_calls > 0 && static_cast<unsigned long>(max_calls) < n) n = max_calls;
    t[0] = t[1] = t[2] = t[3] = 0.0f;
    for (unsigned long i = 0; i < n; i++) {
        hipEvent_t *ev = pl->ev[(pl->calls - 1 - i) % fmrx_pipeline::kRing];
        FMRX_HIP(hipEventSynchronize(ev[3]));
        float d[4];
        FMRX_HIP(hipEventElapsedTime(&d[0], ev[0], ev[1]));
        FMRX_HIP(hipEventElapsedTime(&d[1], ev[1], ev[2]));
        FMRX_HIP(hipEventElapsedTime(&d[2], ev[2], ev[3]));
        FMRX_HIP(hipEventElapsedTime(&d[3], ev[0], ev[3]));
        for (int k = 0; k < 4; k++) t[k] += d[k];
    }
    *count = static_cast<int>(n);
    return FMRX_OK;
}

}  // extern "C"

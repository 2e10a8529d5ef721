// channels.hip -- many independent mono channels per device call (include/fmrx.h: fmrx_channels_*).
//
// The reference runs one pipeline per process (src/project.cpp:460-468: one PARAMS / STATES set); a receiver
// bank is N processes.  A live channel delivers reference-size blocks (51 200 complex samples every 21.3 ms at
// 2.4 MS/s): far too little work for one launch to fill the chip, and N launches per block period cost N times
// the launch latency.  This entry point processes the current block of N channels in ONE launch of the fused
// mono kernel, with no per-channel device state beyond raw bytes:
//
// The mono chain of modes 0/1 is a sliding-window map of its input: audio sample k depends on the
// rf_decim*(audio_taps) + rf_taps - 1 complex input samples in front of it (1 110 at 101/101 taps) and on nothing
// else, so the whole carried state of a mono channel (I/Q FIR state, prev_i/prev_q, state_mono) is a function of
// the channel's last 1 110 input samples.  Each channel owns a slot [history | block] in one device buffer; the N
// slots back to back are processed as ONE pseudo-stream.  Audio samples whose window reaches back into the
// previous slot (the first hist_samples/(rf_decim*audio_decim) of every slot) are computed and thrown away -- 2.3 %
// extra work at the reference's block size -- and all others see exactly the samples the streaming pipeline would
// have seen: IF and discriminator values are bit-identical to fmrx_pipeline's (integer arithmetic, position
// independent), audio equal to within the float32 summation order of the fused kernel's MFMA tiles (<= 2e-6).
// After the launch a small kernel moves every slot's last hist_samples samples into its history and compacts
// the audio / PCM of all channels into the caller's [n_channels][n_audio] arrays.
#include "fmrx_internal.hpp"

using namespace fmrx;

struct fmrx_channels {
    fmrx_params p{};
    int n_channels = 0, device = 0;
    size_t block_bytes = 0;      // bytes per channel and call
    size_t hist_bytes = 0;       // bytes of history in front of every slot's block (multiple of 16 and of 2*rf_decim*audio_decim)
    size_t slot_bytes = 0;
    size_t n_audio = 0;          // audio samples per channel and call
    size_t junk_audio = 0;       // audio samples per slot computed from the history region (discarded)
    Options opt;
    FePlan fe;
    AudioPlan audio;
    hipStream_t stream = nullptr;
    DevBuf<uint8_t> slots;
    DevBuf<float> zeros, audio_all;
    DevBuf<int16_t> pcm_all, pcm_out;
    DevBuf<float> f32_out;
    // banks in the reference's evaluation order, and every stereo bank: channels_stereo.hip
    int audio_channels = 1, exact = 0;
    StereoBank *bank = nullptr;
};

namespace {

// per channel: history <- the slot's last hist_bytes bytes; audio / PCM of the slot, minus the junk in front,
// -> the caller's contiguous per-channel arrays
__global__ void channels_finish_kernel(uint8_t *__restrict__ slots, long slot_bytes, long hist_bytes, const float *__restrict__ audio_all,
                                       const int16_t *__restrict__ pcm_all, long slot_audio, long junk_audio, long n_audio,
                                       float *__restrict__ audio_out, int16_t *__restrict__ pcm_out)
{
    const long c = blockIdx.y;
    uint8_t *slot = slots + c * slot_bytes;
    const long t = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x, nt = static_cast<long>(gridDim.x) * blockDim.x;
    const long so = c * slot_audio + junk_audio, dof = c * n_audio;   // first element of this channel: source, destination
    // four outputs per access where source and destination allow it (8-byte s16 / 16-byte f32 pieces): true for every reference
    // shape (52 600 / 1 024 elements per slot / channel, 28 in front)
    if (n_audio % 4 == 0 && so % 4 == 0 && dof % 4 == 0 && (!pcm_out || (reinterpret_cast<uintptr_t>(pcm_out) % 8 == 0 &&
        reinterpret_cast<uintptr_t>(pcm_all) % 8 == 0)) && (!audio_out || (reinterpret_cast<uintptr_t>(audio_out) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(audio_all) % 16 == 0))) {
        typedef short s4v __attribute__((ext_vector_type(4)));
        typedef float f4v __attribute__((ext_vector_type(4)));
        for (long k = t; k < n_audio / 4; k += nt) {
            if (audio_out) reinterpret_cast<f4v *>(audio_out + dof)[k] = reinterpret_cast<const f4v *>(audio_all + so)[k];
            if (pcm_out) reinterpret_cast<s4v *>(pcm_out + dof)[k] = reinterpret_cast<const s4v *>(pcm_all + so)[k];
        }
    } else {
        for (long k = t; k < n_audio; k += nt) {
            if (audio_out) audio_out[dof + k] = audio_all[so + k];
            if (pcm_out) pcm_out[dof + k] = pcm_all[so + k];
        }
    }
    // history: 16-byte pieces; source and destination overlap only if the block is shorter than the history, which
    // create() rejects, so a plain copy is safe whatever the order
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 *src = reinterpret_cast<const u4 *>(slot + slot_bytes - hist_bytes);
    u4 *dst = reinterpret_cast<u4 *>(slot);
    if (blockIdx.x == 0)
        for (long i = threadIdx.x; i < hist_bytes / 16; i += blockDim.x) dst[i] = src[i];
}

}  // namespace

extern "C" {

int fmrx_channels_create(fmrx_channels **out, const fmrx_params *p, int n_channels, size_t block_bytes, int device)
{
    return fmrx_channels_create_ex(out, p, n_channels, 1, 0, block_bytes, device);
}

int fmrx_channels_create_ex(fmrx_channels **out, const fmrx_params *p, int n_channels, int audio_channels, int exact,
                            size_t block_bytes, int device)
{
    if (!out || !p) return fail(FMRX_EINVAL, "channels_create: null argument");
    if (n_channels < 1) return fail(FMRX_EINVAL, "channels_create: n_channels must be >= 1");
    if (audio_channels != 1 && audio_channels != 2) return fail(FMRX_EINVAL, "channels_create: audio_channels must be 1 (mono) or 2 (stereo)");
    if (audio_channels == 2 || exact || p->audio_upsamp != 0) {   // (mono, fast, modes 0/1: the fused-kernel bank below)
        // integer-decimation modes: whole audio samples; resampling modes: the finer rule (n_if * U % D == 0) is checked by the bank
        const size_t unit4 = static_cast<size_t>(2) * p->rf_decim * (p->audio_upsamp ? 1 : p->audio_decim);
        if (block_bytes == 0 || block_bytes % unit4 || block_bytes % 16)
            return fail(FMRX_EINVAL, "channels_create: block_bytes must be a multiple of 16 and of %zu", unit4);
        FMRX_TRY(require_device());
        FMRX_HIP(hipSetDevice(device));
        fmrx_channels *c = new fmrx_channels;
        c->p = *p;
        c->n_channels = n_channels;
        c->device = device;
        c->block_bytes = block_bytes;
        c->audio_channels = audio_channels;
        c->exact = exact ? 1 : 0;
        c->opt = options_snapshot();
        auto body = [&]() -> int {
            FMRX_TRY(stereo_bank_create(&c->bank, *p, n_channels, audio_channels, c->exact, block_bytes));
            c->n_audio = stereo_bank_n_audio(c->bank);
            FMRX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            FMRX_TRY(c->pcm_out.alloc(c->n_audio * n_channels * audio_channels));
            FMRX_TRY(c->f32_out.alloc(c->n_audio * n_channels * audio_channels));
            return FMRX_OK;
        };
        const int rc = body();
        if (rc != FMRX_OK) {
            fmrx_channels_destroy(c);
            return rc;
        }
        *out = c;
        return FMRX_OK;
    }
    const size_t unit = static_cast<size_t>(2) * p->rf_decim * p->audio_decim;
    if (block_bytes == 0 || block_bytes % unit || block_bytes % 16)
        return fail(FMRX_EINVAL, "channels_create: block_bytes must be a multiple of 16 and of 2*rf_decim*audio_decim = %zu", unit);
    FMRX_TRY(require_device());
    FMRX_HIP(hipSetDevice(device));
    fmrx_channels *c = new fmrx_channels;
    c->p = *p;
    c->n_channels = n_channels;
    c->device = device;
    c->block_bytes = block_bytes;
    c->opt = options_snapshot();
    auto body = [&]() -> int {
        std::vector<float> h(p->rf_taps);
        design_lpf(static_cast<float>(p->rf_Fs), 100000.0f, p->rf_taps, h.data());        // src/project.cpp:50
        FMRX_TRY(fe_plan_init(c->fe, h.data(), p->rf_taps, p->rf_decim));
        h.resize(p->audio_taps);
        design_lpf(static_cast<float>(p->if_Fs), 16000.0f, p->audio_taps, h.data());      // src/project.cpp:321
        FMRX_TRY(audio_plan_init(c->audio, h.data(), p->audio_taps, p->audio_decim));
        // samples an audio output reaches back: rf_decim*audio_taps + rf_taps - 1, plus what the kernel's tiles read in
        // front of a run (one IF sample for the discriminator, 16-byte rounding): rounded up to whole audio samples
        // and 16-byte multiples
        const size_t reach = static_cast<size_t>(p->rf_decim) * (p->audio_taps + 1) + p->rf_taps + 8 * p->rf_decim;
        size_t hs = (reach * 2 + unit - 1) / unit * unit;
        while (hs % 16) hs += unit;
        c->hist_bytes = hs;
        if (block_bytes < hs) return fail(FMRX_EINVAL, "channels_create: block_bytes %zu < the %zu bytes of history a channel carries", block_bytes, hs);
        c->slot_bytes = hs + block_bytes;
        c->n_audio = block_bytes / unit;
        c->junk_audio = hs / unit;
        const size_t total = c->slot_bytes * n_channels;
        if (!mono_fused_available(c->fe, c->audio, reinterpret_cast<const uint8_t *>(16), total / 2, reinterpret_cast<const uint8_t *>(16)))
            return fail(FMRX_EINVAL, "channels_create: no fused mono kernel for rf_taps %d / decim %d / audio_taps %d / decim %d",
                        p->rf_taps, p->rf_decim, p->audio_taps, p->audio_decim);
        FMRX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        FMRX_TRY(c->slots.alloc(total + 64));
        FMRX_TRY(k_fill_u8(c->slots.p, total + 64, 128, nullptr));                        // silence: the state of a stream that starts here
        FMRX_TRY(c->zeros.alloc(p->audio_taps + 64));
        FMRX_HIP(hipMemset(c->zeros.p, 0, (p->audio_taps + 64) * sizeof(float)));
        const size_t all = total / unit;
        FMRX_TRY(c->audio_all.alloc(all + 16));
        FMRX_TRY(c->pcm_all.alloc(all + 16));
        FMRX_TRY(c->pcm_out.alloc(c->n_audio * n_channels));
        FMRX_TRY(c->f32_out.alloc(c->n_audio * n_channels));
        FMRX_HIP(hipDeviceSynchronize());
        return FMRX_OK;
    };
    const int rc = body();
    if (rc != FMRX_OK) {
        fmrx_channels_destroy(c);
        return rc;
    }
    *out = c;
    return FMRX_OK;
}

int fmrx_channels_destroy(fmrx_channels *c)
{
    if (!c) return FMRX_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    if (c->bank) {
        (void)hipDeviceSynchronize();
        stereo_bank_destroy(c->bank);
    }
    delete c;
    return FMRX_OK;
}

size_t fmrx_channels_n_audio(const fmrx_channels *c) { return c ? c->n_audio : 0; }

int fmrx_channels_input_layout(const fmrx_channels *c, uint8_t **d_first_block, size_t *pitch_bytes)
{
    if (!c || !d_first_block || !pitch_bytes) return fail(FMRX_EINVAL, "channels_input_layout: null argument");
    if (c->bank) {
        *d_first_block = stereo_bank_first_block(c->bank);
        *pitch_bytes = stereo_bank_pitch(c->bank);
        return FMRX_OK;
    }
    *d_first_block = c->slots.p + c->hist_bytes;
    *pitch_bytes = c->slot_bytes;
    return FMRX_OK;
}

int fmrx_channels_reset(fmrx_channels *c, int channel)
{
    if (!c) return fail(FMRX_EINVAL, "channels_reset: null handle");
    if (channel >= c->n_channels) return fail(FMRX_EINVAL, "channels_reset: channel %d of %d", channel, c->n_channels);
    FMRX_HIP(hipSetDevice(c->device));
    if (c->bank) return stereo_bank_reset(c->bank, channel);
    FMRX_HIP(hipDeviceSynchronize());
    if (channel < 0) return k_fill_u8(c->slots.p, c->slot_bytes * c->n_channels, 128, nullptr);
    return k_fill_u8(c->slots.p + static_cast<size_t>(channel) * c->slot_bytes, c->hist_bytes, 128, nullptr);
}

int fmrx_channels_process_dev(fmrx_channels *c, float *d_audio_f32, int16_t *d_pcm16, int pcm_policy, void *stream)
{
    if (!c) return fail(FMRX_EINVAL, "channels_process_dev: null handle");
    FMRX_HIP(hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->bank) return stereo_bank_process_dev(c->bank, d_audio_f32, d_pcm16, pcm_policy, s);
    const size_t total = c->slot_bytes * c->n_channels;
    const float *zend = c->zeros.p + c->p.audio_taps + 32;     // "one past the previous block's last discriminator sample": zeros
    FMRX_TRY(mono_fused_launch(c->fe, c->audio, c->slots.p, total / 2, c->fe.silence.p, c->zeros.p, zend, nullptr, 0, nullptr,
                               d_audio_f32 ? c->audio_all.p : nullptr, c->pcm_all.p, pcm_policy, nullptr, c->opt, s));
    const size_t per4 = (c->n_audio + 3) / 4;                  // four outputs per thread where the shapes allow it (see the kernel)
    const unsigned gx = static_cast<unsigned>((per4 + 255) / 256 < 8 ? (per4 + 255) / 256 : 8);
    hipLaunchKernelGGL(channels_finish_kernel, dim3(gx ? gx : 1, c->n_channels), dim3(256), 0, s, c->slots.p,
                       static_cast<long>(c->slot_bytes), static_cast<long>(c->hist_bytes), c->audio_all.p, c->pcm_all.p,
                       static_cast<long>(c->junk_audio + c->n_audio), static_cast<long>(c->junk_audio), static_cast<long>(c->n_audio),
                       d_audio_f32, d_pcm16);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch channels_finish_kernel: %s", hipGetErrorString(e));
    return FMRX_OK;
}

int fmrx_channels_load_dev(fmrx_channels *c, const uint8_t *d_iq, void *stream)
{
    if (!c || !d_iq) return fail(FMRX_EINVAL, "channels_load_dev: null argument");
    FMRX_HIP(hipSetDevice(c->device));
    uint8_t *first = c->slots.p + c->hist_bytes;
    size_t pitch = c->slot_bytes;
    if (c->bank) {
        first = stereo_bank_first_block(c->bank);
        pitch = stereo_bank_pitch(c->bank);
    }
    FMRX_HIP(hipMemcpy2DAsync(first, pitch, d_iq, c->block_bytes, c->block_bytes, c->n_channels, hipMemcpyDeviceToDevice,
                              static_cast<hipStream_t>(stream)));
    return FMRX_OK;
}

int fmrx_channels_process(fmrx_channels *c, const uint8_t *iq, float *audio_f32, int16_t *pcm16, int pcm_policy)
{
    if (!c || !iq) return fail(FMRX_EINVAL, "channels_process: null argument");
    FMRX_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // channel-major host array [n_channels][block_bytes] -> the block region of every slot: one strided copy
    uint8_t *first = c->slots.p + c->hist_bytes;
    size_t pitch = c->slot_bytes;
    if (c->bank) {
        first = stereo_bank_first_block(c->bank);
        pitch = stereo_bank_pitch(c->bank);
    }
    FMRX_HIP(hipMemcpy2DAsync(first, pitch, iq, c->block_bytes, c->block_bytes, c->n_channels, hipMemcpyHostToDevice, s));
    FMRX_TRY(fmrx_channels_process_dev(c, audio_f32 ? c->f32_out.p : nullptr, pcm16 ? c->pcm_out.p : nullptr, pcm_policy, s));
    const size_t n = c->n_audio * c->n_channels * c->audio_channels;
    if (audio_f32) FMRX_HIP(hipMemcpyAsync(audio_f32, c->f32_out.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
    if (pcm16) FMRX_HIP(hipMemcpyAsync(pcm16, c->pcm_out.p, n * sizeof(int16_t), hipMemcpyDeviceToHost, s));
    FMRX_HIP(hipStreamSynchronize(s));
    return FMRX_OK;
}

int fmrx_channels_read_tap(fmrx_channels *c, int channel, int which, float *out, size_t *n)
{
    if (!c || !n) return fail(FMRX_EINVAL, "channels_read_tap: null argument");
    if (!c->bank) return fail(FMRX_EINVAL, "channels_read_tap: only banks created with exact = 1 keep their intermediates in memory");
    FMRX_HIP(hipSetDevice(c->device));
    return stereo_bank_read_tap(c->bank, channel, which, out, n);
}

}  // extern "C"

// channels_stereo.hip -- N independent STEREO receivers per device call (include/fmrx.h: fmrx_channels_create_ex,
// audio_channels = 2), and the bit-exact form of the mono bank.
//
// The reference runs one receiver per process (one PARAMS / STATES set, src/project.cpp:455-468); its stereo
// thread body RF_STEREO (src/project.cpp:154-309) is, per block of fm_demod:
//     allPass -> convolveBlockFastFIR (mono) | convolveBlockFIR (pilot 18.5-19.5 kHz) -> fmPLL |
//     convolveBlockFIR (22-54 kHz) -> mixer -> convolveBlockFastFIR (stereo) -> L = st + mono, R = mono - st.
// fmPLL (src/filter.cpp:32-80) is a serial float32 recurrence through sinf / cosf / atan2f: inside ONE channel
// it cannot be cut in time without leaving the reference's trajectory (the recurrence is
// chaotic on the float32 grid of its phase argument).  ACROSS channels it is embarrassingly parallel: every
// channel owns its STATES.  So the bank gives each channel ONE LANE that walks the exact recurrence, 64 channels
// per wave, and runs every other stage as wide kernels over (channel, sample):
//
//   chs_fe_exact_kernel      u8 I/Q -> (u-128)/128 -> rf FIR -> decimate -> discriminator   (src/project.cpp:82-128)
//   chs_bpf_exact_kernel     both band-pass filters of the stereo path in one pass           (:202, :207)
//   pll_channels_kernel      fmPLL, lane = channel (kernels_pll.hip)                          (:237)
//   chs_out_kernel           NCO cosine, mixer, both audio FIRs, L/R, PCM                     (:246-302)
//   chs_finish_kernel        carried state: every row's tail -> its history
//
// "Exact" means the reference's float32 operations in the reference's order -- separately rounded products and
// sums, taps ascending (src/filter.cpp:133-188), glibc 2.35's sinf / cosf / atan2f (glibc_libm.hpp) -- so that
// left and right equal the compiled reference's bit for bit, per channel, for any stream length.  That order
// fixes the arithmetic (2 vector instructions per tap and output, nothing for the matrix cores to do), not the
// schedule: a thread owns R = 8 consecutive outputs and visits its window newest sample first, so that one
// converted sample feeds all the outputs it belongs to while every output still meets its taps in ascending
// order; the taps of a step are one aligned scalar load (s_load_dwordx16) from a step-major table.
//
// Data layout: everything is channel-major rows with the carried history in front:
//   slots   u8  [n_channels][hist_bytes | block_bytes]   raw I/Q as on stdin; the history IS I_state/Q_state/prev_i/prev_q
//   demod   f32 [n_channels][Hd | n_if | pad]            discriminator output; history = state_mono / _stereo / _carrier / _allpass
//   carrier, bpf, trig f32 [n_channels][n_if + pad]      pilot band-pass, 22-54 kHz band-pass, raw trigArg of every PLL step
//   pll     f32 [n_channels][8]                          state_PLL (6) ; nco0 [n_channels] = PLL[0] of this call
//   mixtail f32 [2][n_channels][Hm]                      state_stereofilt, ping-pong
#include "device_math.hpp"
#include "fmrx_internal.hpp"
#include "glibc_libm.hpp"

#pragma clang fp contract(off)

namespace fmrx {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// taps of one or two steps: wave-uniform, straight into SGPRs (inline asm keeps the loads where they are written)
#define CHS_TAPS_ISSUE(hp, table, off) asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(hp) : "s"(table), "i"(off))
#define CHS_TAPS_WAIT(hp) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(hp))

#define CHS_LAUNCH_CHECK(name)                                                                    \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) return fail(FMRX_EHIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

constexpr int kR = 8;   // consecutive outputs per thread of the exact FIR kernels (= table entries per step)

// ---- front end, reference order ------------------------------------------------------------------------------
// Thread: outputs k0 .. k0+R-1 (I and Q in the two halves of v_pk_mul_f32 / v_pk_add_f32).  Window sample j
// (j = 0: stream sample k0*D - (T-1)) meets output r with tap n = r*D + (T-1) - j.  Step u visits j = W-1-u, i.e.
// the window newest sample first: for every output the taps then come in ascending n, as in the reference's loop
// (src/filter.cpp:166-177), and each sample is converted once.  Table entry [u][r] = h[n] / 128 (0 where n is
// out of range: those pairs are not executed, the entry only keeps the load shape); the bytes are flipped to
// int8 (u ^ 0x80 = u - 128), so h/128 * (u-128) is the reference's product h * ((u-128)/128) bit for bit (a power
// of two moves no rounding; create() checks that no tap is so small that h/128 would be subnormal).
template <int T, int D>
struct FeX {
    static constexpr int R = kR;
    static constexpr int LEAD = (8 - (T - 1) % 8) % 8;          // samples in front of the window so that it starts 16 B aligned
    static constexpr int W = D * (R - 1) + T;                   // samples one thread needs
    static constexpr int NB = (2 * (W + LEAD) + 15) / 16;       // 16-byte loads per thread
    static constexpr int NG = (W + 1) / 2;                      // scalar-load groups: two steps each
    static constexpr int TILE = 63 * R;                         // new outputs per wave (lane 0 recomputes the R in front)
    static constexpr int HIST = 2 * (T - 1 + LEAD) + 2 * D * R; // bytes of history in front of a block
    static_assert((2 * D * R) % 16 == 0, "thread windows must start 16-byte aligned");
};

template <int T, int D, int G>
__device__ __forceinline__ void fex_step(const uint32_t (&raw)[FeX<T, D>::NB * 4], const float *__restrict__ table,
                                         f2 (&acc)[kR], f16v &hA, f16v &hB)
{
    using C = FeX<T, D>;
    if constexpr (G < C::NG) {
        float hq[16];
        if constexpr (G % 2 == 0) {
            CHS_TAPS_WAIT(hA);
            if constexpr (G + 1 < C::NG) CHS_TAPS_ISSUE(hB, table, (G + 1) * 64);
#pragma unroll
            for (int k = 0; k < 16; k++) hq[k] = hA[k];
        } else {
            CHS_TAPS_WAIT(hB);
            if constexpr (G + 1 < C::NG) CHS_TAPS_ISSUE(hA, table, (G + 1) * 64);
#pragma unroll
            for (int k = 0; k < 16; k++) hq[k] = hB[k];
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int u = 2 * G + e;
            if (u < C::W) {
                const int j = C::W - 1 - u;
                const int bo = 2 * (j + C::LEAD);
                const uint32_t w = raw[bo / 4];
                f2 xs;   // v_cvt_f32_i32_sdwa sext(w) src0_sel:BYTE_n
                if ((bo % 4) == 0) {
                    xs.x = static_cast<float>(static_cast<int8_t>(w & 0xffu));
                    xs.y = static_cast<float>(static_cast<int8_t>((w >> 8) & 0xffu));
                } else {
                    xs.x = static_cast<float>(static_cast<int8_t>((w >> 16) & 0xffu));
                    xs.y = static_cast<float>(static_cast<int8_t>(w >> 24));
                }
#pragma unroll
                for (int r = 0; r < kR; r++) {
                    const int n = r * D + (T - 1) - j;
                    if (n >= 0 && n < T) {
                        const float h = hq[8 * e + r];
                        const f2 prod = xs * (f2){h, h};      // separately rounded product ...
                        acc[r] = acc[r] + prod;               // ... and sum (src/filter.cpp:169, 174)
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kR; r++) asm volatile("" : "+v"(acc[r]));
        fex_step<T, D, G + 1>(raw, table, acc, hA, hB);
    }
}

// One wave per tile of 63*R outputs of one channel; lane 0 recomputes the R outputs in front of the tile (from the
// history in front of the block for the first tile) only to hand IF[k0-1] to lane 1: every IF sample is produced by
// the same instruction sequence wherever it is computed.
template <int T, int D>
__global__ __launch_bounds__(256) void chs_fe_exact_kernel(const uint8_t *__restrict__ slots, long slot_bytes, int hist_bytes,
                                                            long n_if, long ntiles, long wgs_per_channel,
                                                            const float *__restrict__ table, float *__restrict__ demod,
                                                            long dpitch, int Hd)
{
    using C = FeX<T, D>;
    constexpr int R = kR;
    const int lane = threadIdx.x & 63;
    const long c = blockIdx.x / wgs_per_channel;
    const long tile = (blockIdx.x % wgs_per_channel) * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;                                  // wave-uniform
    const uint8_t *blk = slots + c * slot_bytes + hist_bytes;
    const long k0 = tile * C::TILE + static_cast<long>(lane - 1) * R;
    const long w0 = k0 * D - (T - 1) - C::LEAD;                  // first sample of the 16-byte aligned window
    const u4 *src = reinterpret_cast<const u4 *>(blk + 2 * w0);
    uint32_t raw[C::NB * 4];
#pragma unroll
    for (int i = 0; i < C::NB; i++) {
        const u4 v = src[i] ^ 0x80808080u;
        raw[4 * i] = v.x;
        raw[4 * i + 1] = v.y;
        raw[4 * i + 2] = v.z;
        raw[4 * i + 3] = v.w;
    }
    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
    f16v hA, hB;
    CHS_TAPS_ISSUE(hA, table, 0);
    fex_step<T, D, 0>(raw, table, acc, hA, hB);
    // fmDemod (src/filter.cpp:248-266): IF[k-1] is the previous accumulator, across threads the lane next door
    const float pi = __shfl_up(acc[R - 1].x, 1, 64), pq = __shfl_up(acc[R - 1].y, 1, 64);
    float d[R];
    d[0] = demod_exact(acc[0].x, acc[0].y, pi, pq);
#pragma unroll
    for (int r = 1; r < R; r++) d[r] = demod_exact(acc[r].x, acc[r].y, acc[r - 1].x, acc[r - 1].y);
    if (lane == 0 || k0 >= n_if) return;
    float *out = demod + c * dpitch + Hd + k0;
    if (k0 + R <= n_if) {
        reinterpret_cast<f4 *>(out)[0] = (f4){d[0], d[1], d[2], d[3]};
        reinterpret_cast<f4 *>(out)[1] = (f4){d[4], d[5], d[6], d[7]};
    } else {
#pragma unroll
        for (int r = 0; r < R; r++)
            if (k0 + r < n_if) out[r] = d[r];
    }
}

// ---- both band-pass filters, reference order -------------------------------------------------------------------
// convolveBlockFIR twice on the same input (src/project.cpp:202, 207 -> src/filter.cpp:133-154): the two filters
// ride in the two halves of the packed instructions, acc(st, car) = acc + x * (h_st[n], h_car[n]).  Same scheme as the
// front end with D = 1: window sample i (i = 0: x[k0 - (T-1)]) meets output r with tap n = r + (T-1) - i; step u
// visits i = W-1-u; table entry [u][r] = (h_st[n], h_car[n]).
template <int T>
struct BpX {
    static constexpr int R = kR;
    static constexpr int LEAD = (4 - (T - 1) % 4) % 4;
    static constexpr int W = R + T - 1;
    static constexpr int NW = W + LEAD;                         // floats loaded
    static_assert(NW % 4 == 0, "window must be whole 16-byte chunks");
};

template <int T, int G>
__device__ __forceinline__ void bpx_step(const float (&w)[BpX<T>::NW], const float *__restrict__ table, f2 (&acc)[kR], f16v &hA,
                                         f16v &hB)
{
    using C = BpX<T>;
    if constexpr (G < C::W) {
        float hq[16];
        if constexpr (G % 2 == 0) {
            CHS_TAPS_WAIT(hA);
            if constexpr (G + 1 < C::W) CHS_TAPS_ISSUE(hB, table, (G + 1) * 64);
#pragma unroll
            for (int k = 0; k < 16; k++) hq[k] = hA[k];
        } else {
            CHS_TAPS_WAIT(hB);
            if constexpr (G + 1 < C::W) CHS_TAPS_ISSUE(hA, table, (G + 1) * 64);
#pragma unroll
            for (int k = 0; k < 16; k++) hq[k] = hB[k];
        }
        constexpr int i = C::W - 1 - G;
        const float x = w[i + C::LEAD];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            const int n = r + (T - 1) - i;
            if (n >= 0 && n < T) {
                const f2 prod = (f2){x, x} * (f2){hq[2 * r], hq[2 * r + 1]};
                acc[r] = acc[r] + prod;
            }
        }
#pragma unroll
        for (int r = 0; r < kR; r++) asm volatile("" : "+v"(acc[r]));
        bpx_step<T, G + 1>(w, table, acc, hA, hB);
    }
}

template <int T>
__global__ __launch_bounds__(256) void chs_bpf_exact_kernel(const float *__restrict__ demod, long dpitch, int Hd, long n_if,
                                                             long wgs_per_channel, const float *__restrict__ table,
                                                             float *__restrict__ y_st, float *__restrict__ y_car, long ypitch)
{
    using C = BpX<T>;
    constexpr int R = kR;
    const long c = blockIdx.x / wgs_per_channel;
    const long k0 = ((blockIdx.x % wgs_per_channel) * 256 + threadIdx.x) * R;
    if (k0 >= n_if) return;
    const float *x = demod + c * dpitch + Hd;
    const f4 *src = reinterpret_cast<const f4 *>(x + k0 - (T - 1) - C::LEAD);
    float w[C::NW];
#pragma unroll
    for (int i = 0; i < C::NW / 4; i++) {
        const f4 v = src[i];
        w[4 * i] = v.x;
        w[4 * i + 1] = v.y;
        w[4 * i + 2] = v.z;
        w[4 * i + 3] = v.w;
    }
    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
    f16v hA, hB;
    CHS_TAPS_ISSUE(hA, table, 0);
    bpx_step<T, 0>(w, table, acc, hA, hB);
    float *ds = y_st + c * ypitch + k0, *dc = y_car + c * ypitch + k0;
    if (k0 + R <= n_if) {
        reinterpret_cast<f4 *>(ds)[0] = (f4){acc[0].x, acc[1].x, acc[2].x, acc[3].x};
        reinterpret_cast<f4 *>(ds)[1] = (f4){acc[4].x, acc[5].x, acc[6].x, acc[7].x};
        reinterpret_cast<f4 *>(dc)[0] = (f4){acc[0].y, acc[1].y, acc[2].y, acc[3].y};
        reinterpret_cast<f4 *>(dc)[1] = (f4){acc[4].y, acc[5].y, acc[6].y, acc[7].y};
    } else {
#pragma unroll
        for (int r = 0; r < R; r++)
            if (k0 + r < n_if) {
                ds[r] = acc[r].x;
                dc[r] = acc[r].y;
            }
    }
}

// ---- everything behind the PLL ------------------------------------------------------------------------------------
// The NCO output cosf(trigArg*ncoScale + phaseAdjust) (src/filter.cpp:72; off the recurrence's chain), the mixer
// (src/project.cpp:246-248), the two audio convolveBlockFastFIR calls of RF_STEREO -- mono branch on the all-passed
// discriminator output (:194, :219), stereo branch on the mixer output (:257) --, the L/R combine (:277-280) and the
// interleaved PCM writer (:292-302).  A workgroup stages the window of its NT*R audio outputs as (mono, mixer)
// pairs in LDS (the mixer products and the cosines are formed while staging and never go to HBM); the two FIRs share
// the taps, so they ride in the two halves of the packed instructions.  EXACT: glibc's cosf, products and sums rounded
// separately, taps ascending; otherwise one fma per tap and the hardware cosine (the fast bank).
// Mono banks (STEREO = false) run the same kernel with the mixer half compiled out.
template <int T, int D, int R, int NT, bool EXACT, bool STEREO>
__global__ __launch_bounds__(NT) void chs_out_kernel(const float *__restrict__ demod, long dpitch, int Hd, const float *__restrict__ bpf,
                                                      const float *__restrict__ trig, long ypitch, const float *__restrict__ nco0,
                                                      const float *__restrict__ mix_tail_in, float *__restrict__ mix_tail_out, int hm,
                                                      long n_if, int delay, float nco_scale, float phase_adjust,
                                                      const float *__restrict__ h, long wgs_per_channel, float *__restrict__ audio,
                                                      int16_t *__restrict__ pcm, int wrap, long n_out)
{
    constexpr int NOUT = NT * R;
    constexpr int WL = D * (NOUT - 1) + T;
    extern __shared__ f2 win[];
    const int t = threadIdx.x;
    const long c = blockIdx.x / wgs_per_channel;
    const long a0 = (blockIdx.x % wgs_per_channel) * NOUT;
    const long g0 = D * a0 - (T - 1);                      // IF index of window sample 0
    const float *dm = demod + c * dpitch + Hd;
    const float *bp = STEREO ? bpf + c * ypitch : nullptr, *tr = STEREO ? trig + c * ypitch : nullptr;
    const float *tin = STEREO ? mix_tail_in + c * hm : nullptr;
    float *tout = STEREO ? mix_tail_out + c * hm : nullptr;
    // D-1 samples past the window are visited too: when the block ends exactly on a tile boundary nobody's window reaches
    // the block's last D-1 samples, and they belong to the tail this call leaves behind
    constexpr int NJ = (WL + D - 1 + NT - 1) / NT;
    float vm[NJ], va[NJ], vb[NJ];
#pragma unroll
    for (int q = 0; q < NJ; q++) {
        const int j = t + q * NT;
        const long g = g0 + j;
        const bool valid = j < WL + D - 1 && g < n_if;
        vm[q] = 0.0f;
        va[q] = 0.0f;
        vb[q] = 0.0f;
        if (valid) {
            vm[q] = dm[g - delay];                             // history in front of the row: negative indices are valid
            if (STEREO) {
                va[q] = g >= 0 ? bp[g] : tin[hm + g];
                if (g > 0) vb[q] = tr[g - 1];                  // PLL[g] = cosf(trigArg[g-1]*ncoScale + phaseAdjust); PLL[0] = state[4]
            }
        }
    }
    const float first = STEREO ? nco0[c] : 0.0f;
#pragma unroll
    for (int q = 0; q < NJ; q++) {
        const int j = t + q * NT;
        const long g = g0 + j;
        if (j >= WL + D - 1) continue;
        float m = 0.0f, x = 0.0f;
        if (g < n_if) {
            m = vm[q];
            if (STEREO) {
                if (g >= 0) {
                    float nco;
                    if (EXACT) {
                        const float a = vb[q] * nco_scale + phase_adjust;
                        nco = g > 0 ? glibc235::cosf_glibc(a) : first;
                    } else {
                        const float a = vb[q] * nco_scale + phase_adjust;
                        const double rev = static_cast<double>(a) * 0.15915494309189533577;
                        nco = g > 0 ? __builtin_amdgcn_cosf(static_cast<float>(rev - rint(rev))) : first;
                    }
                    x = (va[q] * nco) * 2.0f;                  // the reference's order: (stereo_filt * PLL) * 2
                    if (g >= n_if - hm) tout[g - (n_if - hm)] = x;
                } else {
                    x = va[q];
                }
            }
        }
        if (j < WL) win[j] = (f2){m, x};
    }
    __syncthreads();
    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
    const f2 *w0 = win + D * t + (T - 1);
#pragma unroll 8
    for (int n = 0; n < T; n++) {
        const float hn = h[n];                             // wave-uniform: scalar load
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (EXACT) {
                const f2 prod = w0[D * NT * r - n] * (f2){hn, hn};
                acc[r] = acc[r] + prod;
            } else {
                acc[r] = __builtin_elementwise_fma(w0[D * NT * r - n], (f2){hn, hn}, acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const long k = a0 + static_cast<long>(r) * NT + t;
        if (k < n_out) {
            const float mo = acc[r].x, st = acc[r].y;
            if (STEREO) {
                const float l = st + mo, rr = mo - st;     // src/project.cpp:278-279
                if (audio) {
                    audio[c * 2 * n_out + k] = l;
                    audio[c * 2 * n_out + n_out + k] = rr;
                }
                if (pcm) {
                    using s2 = short __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<s2 *>(pcm + 2 * (c * n_out + k)) = (s2){pcm_pack_flat(l, wrap), pcm_pack_flat(rr, wrap)};
                }
            } else {
                if (audio) audio[c * n_out + k] = mo;
                if (pcm) pcm[c * n_out + k] = pcm_pack_flat(mo, wrap);
            }
        }
    }
}

// carried state: per channel, history <- the slot's last hist_bytes bytes; demod history <- the row's last Hd samples
// (create() rejects blocks shorter than either history, so source and destination never overlap)
__global__ void chs_finish_kernel(uint8_t *__restrict__ slots, long slot_bytes, long hist_bytes, float *__restrict__ demod,
                                  long dpitch, int Hd, long n_if)
{
    const long c = blockIdx.x;
    uint8_t *slot = slots + c * slot_bytes;
    const u4 *src = reinterpret_cast<const u4 *>(slot + slot_bytes - hist_bytes);
    u4 *dst = reinterpret_cast<u4 *>(slot);
    for (long i = threadIdx.x; i < hist_bytes / 16; i += blockDim.x) dst[i] = src[i];
    float *row = demod + c * dpitch;
    for (long i = threadIdx.x; i < Hd; i += blockDim.x) row[i] = row[n_if + i];
}

__global__ void chs_fill_state_kernel(float *__restrict__ pll, long n)
{
    const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int u = static_cast<int>(i % 8);
    pll[i] = (u == 2 || u == 4) ? 1.0f : 0.0f;                 // state_PLL = {0, 0, 1, 0, 1, 0}  (src/project.cpp:458)
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------
struct StereoBank {
    fmrx_params p{};
    int n_channels = 0, audio_channels = 2, exact = 1;
    size_t block_bytes = 0, hist_bytes = 0, slot_bytes = 0;
    long n = 0, n_if = 0, n_audio = 0;
    int Ha = 0, delay = 0, Hd = 0, Hm = 0, St = 0;
    long dpitch = 0, ypitch = 0;
    DevBuf<uint8_t> slots;
    DevBuf<float> fe_table, bpf_table, h_audio;
    DevBuf<float> demod, carrier, bpf, trig, pll, nco0, mixtail[2];
    int mix_cur = 0;
};

namespace {

template <int T, int D>
int fe_table_init(StereoBank &b, const float *h)
{
    using C = FeX<T, D>;
    for (int n = 0; n < T; n++) {
        const float a = std::fabs(h[n]);
        if (a != 0.0f && !(a >= 7.8886091e-31f && a <= 1.2676506e30f))   // 2^-100 .. 2^100
            return fail(FMRX_EINVAL, "channels (exact): rf tap %d = %g is outside the range in which h/128 * (u-128) is the reference's product", n, h[n]);
    }
    std::vector<float> tab(static_cast<size_t>(C::NG) * 16, 0.0f);
    for (int u = 0; u < C::W; u++)
        for (int r = 0; r < kR; r++) {
            const int n = r * D + (T - 1) - (C::W - 1 - u);
            if (n >= 0 && n < T) tab[static_cast<size_t>(u) * 8 + r] = h[n] * 0.0078125f;
        }
    FMRX_TRY(b.fe_table.alloc(tab.size()));
    FMRX_HIP(hipMemcpy(b.fe_table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    b.hist_bytes = C::HIST;
    return FMRX_OK;
}

template <int T>
int bpf_table_init(StereoBank &b, const float *h_st, const float *h_car)
{
    using C = BpX<T>;
    std::vector<float> tab(static_cast<size_t>(C::W) * 16, 0.0f);
    for (int u = 0; u < C::W; u++)
        for (int r = 0; r < kR; r++) {
            const int n = r + (T - 1) - (C::W - 1 - u);
            if (n >= 0 && n < T) {
                tab[static_cast<size_t>(u) * 16 + 2 * r] = h_st[n];
                tab[static_cast<size_t>(u) * 16 + 2 * r + 1] = h_car[n];
            }
        }
    FMRX_TRY(b.bpf_table.alloc(tab.size()));
    FMRX_HIP(hipMemcpy(b.bpf_table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return FMRX_OK;
}

#define CHS_FE_CASES(X) X(101, 10) X(101, 5) X(151, 10) X(151, 5) X(13, 10) X(13, 5)
#define CHS_BPF_CASES(X) X(101) X(151) X(13)
#define CHS_OUT_CASES(X) X(101, 5) X(101, 6) X(13, 5) X(13, 6)

template <int T, int D>
int launch_fe(const StereoBank &b, hipStream_t s)
{
    using C = FeX<T, D>;
    const long ntiles = (b.n_if + C::TILE - 1) / C::TILE;
    const long wgs = (ntiles + 3) / 4;
    hipLaunchKernelGGL((chs_fe_exact_kernel<T, D>), dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(256), 0, s, b.slots.p,
                       static_cast<long>(b.slot_bytes), static_cast<int>(b.hist_bytes), b.n_if, ntiles, wgs, b.fe_table.p, b.demod.p,
                       b.dpitch, b.Hd);
    CHS_LAUNCH_CHECK("chs_fe_exact_kernel");
    return FMRX_OK;
}

template <int T>
int launch_bpf(const StereoBank &b, hipStream_t s)
{
    const long wgs = (b.n_if + 256 * kR - 1) / (256 * kR);
    hipLaunchKernelGGL((chs_bpf_exact_kernel<T>), dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(256), 0, s, b.demod.p, b.dpitch,
                       b.Hd, b.n_if, wgs, b.bpf_table.p, b.bpf.p, b.carrier.p, b.ypitch);
    CHS_LAUNCH_CHECK("chs_bpf_exact_kernel");
    return FMRX_OK;
}

template <int T, int D, bool STEREO>
int launch_out(StereoBank &b, float *d_audio, int16_t *d_pcm, int wrap, hipStream_t s)
{
    constexpr int R = 2, NT = 256;
    constexpr size_t lds = (static_cast<size_t>(D) * (NT * R - 1) + T) * sizeof(f2);
    const long wgs = (b.n_audio + NT * R - 1) / (NT * R);
    hipLaunchKernelGGL((chs_out_kernel<T, D, R, NT, true, STEREO>), dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(NT), lds, s,
                       b.demod.p, b.dpitch, b.Hd, b.bpf.p, b.trig.p, b.ypitch, b.nco0.p, b.mixtail[b.mix_cur].p,
                       b.mixtail[b.mix_cur ^ 1].p, b.Hm, b.n_if, b.delay, 2.0f, 0.0f, b.h_audio.p, wgs, d_audio, d_pcm, wrap,
                       b.n_audio);
    CHS_LAUNCH_CHECK("chs_out_kernel");
    return FMRX_OK;
}

}  // namespace

bool stereo_bank_supported(const fmrx_params &p, int audio_channels)
{
    bool fe = false, au = false, st = audio_channels == 1;
#define X(T_, D_) if (p.rf_taps == T_ && p.rf_decim == D_) fe = true;
    CHS_FE_CASES(X)
#undef X
#define X(T_, D_) if (p.audio_taps == T_ && p.audio_decim == D_) au = true;
    CHS_OUT_CASES(X)
#undef X
#define X(T_) if (p.stereo_taps == T_) st = true;
    CHS_BPF_CASES(X)
#undef X
    return fe && au && st && p.audio_upsamp == 0;
}

void stereo_bank_destroy(StereoBank *b) { delete b; }

int stereo_bank_create(StereoBank **out, const fmrx_params &p, int n_channels, int audio_channels, size_t block_bytes)
{
    if (!stereo_bank_supported(p, audio_channels))
        return fail(FMRX_EINVAL, "channels (exact): no reference-order kernels for rf %d/%d, audio %d/%d, stereo %d taps (modes 0 and 1 of the "
                    "reference's tap sets are covered)", p.rf_taps, p.rf_decim, p.audio_taps, p.audio_decim, p.stereo_taps);
    StereoBank *b = new StereoBank;
    b->p = p;
    b->n_channels = n_channels;
    b->audio_channels = audio_channels;
    b->block_bytes = block_bytes;
    auto body = [&]() -> int {
        std::vector<float> h(p.rf_taps);
        design_lpf(static_cast<float>(p.rf_Fs), 100000.0f, p.rf_taps, h.data());             // src/project.cpp:50
#define X(T_, D_) if (p.rf_taps == T_ && p.rf_decim == D_) FMRX_TRY((fe_table_init<T_, D_>(*b, h.data())));
        CHS_FE_CASES(X)
#undef X
        std::vector<float> ha(p.audio_taps);
        design_lpf(static_cast<float>(p.if_Fs), 16000.0f, p.audio_taps, ha.data());           // src/project.cpp:321
        FMRX_TRY(b->h_audio.alloc(p.audio_taps));
        FMRX_HIP(hipMemcpy(b->h_audio.p, ha.data(), p.audio_taps * sizeof(float), hipMemcpyHostToDevice));
        b->n = static_cast<long>(block_bytes / 2);
        b->n_if = b->n / p.rf_decim;
        b->n_audio = b->n_if / p.audio_decim;
        b->Ha = p.audio_taps - 1;
        b->St = audio_channels == 2 ? p.stereo_taps : 0;
        b->delay = audio_channels == 2 ? (p.stereo_taps - 1) / 2 : 0;                        // allPass, src/filter.cpp:14-29
        b->Hd = b->Ha + b->delay;
        if (audio_channels == 2 && b->St - 1 + 3 > b->Hd) b->Hd = b->St - 1 + 3;
        b->Hd = (b->Hd + 3) / 4 * 4 + 4;
        b->Hm = (b->Ha + 3) / 4 * 4 + 4;
        if (block_bytes < b->hist_bytes || b->n_if < b->Hd)
            return fail(FMRX_EINVAL, "channels (exact): block of %zu bytes is shorter than the history a channel carries (%zu bytes, %d IF samples)",
                        block_bytes, b->hist_bytes, b->Hd);
        b->slot_bytes = b->hist_bytes + block_bytes;
        b->dpitch = (b->Hd + b->n_if + 16 + 3) / 4 * 4;
        b->ypitch = (b->n_if + 16 + 3) / 4 * 4;
        const size_t N = static_cast<size_t>(n_channels);
        // the last tile's lanes past the block read on (results discarded): 63*R outputs' worth of bytes behind the last slot
        FMRX_TRY(b->slots.alloc(b->slot_bytes * N + 2 * 64 * kR * p.rf_decim + 64));
        FMRX_TRY(k_fill_u8(b->slots.p, b->slots.n, 128, nullptr));                            // silence: a stream that starts here
        FMRX_TRY(b->demod.alloc(b->dpitch * N + 64));
        FMRX_HIP(hipMemset(b->demod.p, 0, b->demod.bytes()));
        if (audio_channels == 2) {
            std::vector<float> hc(p.stereo_taps), hs(p.stereo_taps);
            design_bpf(static_cast<float>(p.if_Fs), 18.5e3f, 19.5e3f, p.stereo_taps, hc.data());   // src/project.cpp:172
            design_bpf(static_cast<float>(p.if_Fs), 22e3f, 54e3f, p.stereo_taps, hs.data());       // :173
#define X(T_) if (p.stereo_taps == T_) FMRX_TRY(bpf_table_init<T_>(*b, hs.data(), hc.data()));
            CHS_BPF_CASES(X)
#undef X
            FMRX_TRY(b->carrier.alloc(b->ypitch * N + 64));
            FMRX_TRY(b->bpf.alloc(b->ypitch * N + 64));
            FMRX_TRY(b->trig.alloc(b->ypitch * N + 64));
            FMRX_HIP(hipMemset(b->carrier.p, 0, b->carrier.bytes()));
            FMRX_TRY(b->pll.alloc(8 * N));
            FMRX_TRY(b->nco0.alloc(N));
            for (auto &m : b->mixtail) {
                FMRX_TRY(m.alloc(static_cast<size_t>(b->Hm) * N));
                FMRX_HIP(hipMemset(m.p, 0, m.bytes()));
            }
            hipLaunchKernelGGL(chs_fill_state_kernel, dim3(static_cast<unsigned>((8 * N + 255) / 256)), dim3(256), 0, nullptr, b->pll.p,
                               static_cast<long>(8 * N));
            CHS_LAUNCH_CHECK("chs_fill_state_kernel");
        }
        FMRX_HIP(hipDeviceSynchronize());
        return FMRX_OK;
    };
    const int rc = body();
    if (rc != FMRX_OK) {
        delete b;
        return rc;
    }
    *out = b;
    return FMRX_OK;
}

size_t stereo_bank_n_audio(const StereoBank *b) { return static_cast<size_t>(b->n_audio); }
uint8_t *stereo_bank_first_block(const StereoBank *b) { return b->slots.p + b->hist_bytes; }
size_t stereo_bank_pitch(const StereoBank *b) { return b->slot_bytes; }

// back to the start-of-stream state (src/project.cpp:61-65, 446-458): one channel, or all of them (channel < 0)
int stereo_bank_reset(StereoBank *b, int channel)
{
    FMRX_HIP(hipDeviceSynchronize());
    const long lo = channel < 0 ? 0 : channel, hi = channel < 0 ? b->n_channels : channel + 1;
    for (long c = lo; c < hi && channel >= 0; c++) {
        FMRX_TRY(k_fill_u8(b->slots.p + c * b->slot_bytes, b->hist_bytes, 128, nullptr));
        FMRX_HIP(hipMemsetAsync(b->demod.p + c * b->dpitch, 0, b->Hd * sizeof(float), nullptr));
        if (b->audio_channels == 2) {
            for (auto &m : b->mixtail) FMRX_HIP(hipMemsetAsync(m.p + c * b->Hm, 0, b->Hm * sizeof(float), nullptr));
            hipLaunchKernelGGL(chs_fill_state_kernel, dim3(1), dim3(8), 0, nullptr, b->pll.p + 8 * c, 8L);
        }
    }
    if (channel < 0) {
        FMRX_TRY(k_fill_u8(b->slots.p, b->slot_bytes * b->n_channels, 128, nullptr));
        FMRX_HIP(hipMemsetAsync(b->demod.p, 0, b->demod.bytes(), nullptr));
        if (b->audio_channels == 2) {
            for (auto &m : b->mixtail) FMRX_HIP(hipMemsetAsync(m.p, 0, m.bytes(), nullptr));
            const long n8 = 8L * b->n_channels;
            hipLaunchKernelGGL(chs_fill_state_kernel, dim3(static_cast<unsigned>((n8 + 255) / 256)), dim3(256), 0, nullptr, b->pll.p, n8);
        }
    }
    CHS_LAUNCH_CHECK("chs_fill_state_kernel");
    FMRX_HIP(hipDeviceSynchronize());
    return FMRX_OK;
}

// d_audio: [n_channels][audio_channels][n_audio] (stereo: left, then right); d_pcm: [n_channels][n_audio][audio_channels]
int stereo_bank_process_dev(StereoBank *b, float *d_audio, int16_t *d_pcm, int wrap, hipStream_t s)
{
    const fmrx_params &p = b->p;
#define X(T_, D_) if (p.rf_taps == T_ && p.rf_decim == D_) FMRX_TRY((launch_fe<T_, D_>(*b, s)));
    CHS_FE_CASES(X)
#undef X
    if (b->audio_channels == 2) {
#define X(T_) if (p.stereo_taps == T_) FMRX_TRY(launch_bpf<T_>(*b, s));
        CHS_BPF_CASES(X)
#undef X
        // fmPLL(carrier_filt, 19 kHz, if_Fs, ncoScale 2, phaseAdjust 0, normBandwidth 0.01): src/project.cpp:237
        FMRX_TRY(k_fm_pll_channels(b->carrier.p, b->ypitch, static_cast<size_t>(b->n_if), b->n_channels, b->trig.p, b->ypitch, b->pll.p,
                                   b->nco0.p, 19e3f, static_cast<float>(p.if_Fs), 2.0f, 0.0f, 0.01f, s));
#define X(T_, D_) if (p.audio_taps == T_ && p.audio_decim == D_) FMRX_TRY((launch_out<T_, D_, true>(*b, d_audio, d_pcm, wrap, s)));
        CHS_OUT_CASES(X)
#undef X
        b->mix_cur ^= 1;
    } else {
#define X(T_, D_) if (p.audio_taps == T_ && p.audio_decim == D_) FMRX_TRY((launch_out<T_, D_, false>(*b, d_audio, d_pcm, wrap, s)));
        CHS_OUT_CASES(X)
#undef X
    }
    hipLaunchKernelGGL(chs_finish_kernel, dim3(static_cast<unsigned>(b->n_channels)), dim3(64), 0, s, b->slots.p,
                       static_cast<long>(b->slot_bytes), static_cast<long>(b->hist_bytes), b->demod.p, b->dpitch, b->Hd, b->n_if);
    CHS_LAUNCH_CHECK("chs_finish_kernel");
    return FMRX_OK;
}

// diagnostics / tests: one channel's row of an intermediate of the last call.  which: FMRX_TAP_DEMOD, _CARRIER, _STEREO_BPF,
// _PLL (n_if + 1 values: PLL[0] = the state's lastOut, then the finished NCO values)
int stereo_bank_read_tap(StereoBank *b, int channel, int which, float *out, size_t *n)
{
    if (channel < 0 || channel >= b->n_channels) return fail(FMRX_EINVAL, "channels_read_tap: channel %d of %d", channel, b->n_channels);
    FMRX_HIP(hipDeviceSynchronize());
    const size_t n_if = static_cast<size_t>(b->n_if);
    const float *src = nullptr;
    size_t cnt = n_if;
    switch (which) {
    // the finish kernel has copied the row's tail into its front already; the block itself is intact
    case FMRX_TAP_DEMOD: src = b->demod.p + channel * b->dpitch + b->Hd; break;
    case FMRX_TAP_CARRIER: if (b->audio_channels == 2) src = b->carrier.p + channel * b->ypitch; break;
    case FMRX_TAP_STEREO_BPF: if (b->audio_channels == 2) src = b->bpf.p + channel * b->ypitch; break;
    case FMRX_TAP_PLL: if (b->audio_channels == 2) { src = b->trig.p + channel * b->ypitch; cnt = n_if + 1; } break;
    default: break;
    }
    if (!src) return fail(FMRX_EINVAL, "channels_read_tap: tap %d is not kept by this bank", which);
    *n = cnt;
    if (!out) return FMRX_OK;
    if (which == FMRX_TAP_PLL) {
        std::vector<float> t(n_if);
        FMRX_HIP(hipMemcpy(t.data(), src, n_if * sizeof(float), hipMemcpyDeviceToHost));
        FMRX_HIP(hipMemcpy(out, b->nco0.p + channel, sizeof(float), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < n_if; k++) out[k + 1] = glibc235::cosf_glibc(t[k] * 2.0f + 0.0f);   // same function, host build
        return FMRX_OK;
    }
    FMRX_HIP(hipMemcpy(out, src, cnt * sizeof(float), hipMemcpyDeviceToHost));
    return FMRX_OK;
}

}  // namespace fmrx

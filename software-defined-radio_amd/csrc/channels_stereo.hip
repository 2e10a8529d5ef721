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
//   chs_nco_exact_kernel     the NCO output's cosine, off the recurrence's chain              (src/filter.cpp:72)
//   chs_out_exact_kernel     mixer, both audio FIRs, L/R, PCM                                 (:246-302)
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
//   carrier, bpf, trig f32 [n_channels][n_if + pad]      pilot band-pass, 22-54 kHz band-pass, raw trigArg of every PLL step -> NCO output
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
// Workgroups are single waves in all three wide kernels: a stereo call runs them next to the PLL's lanes (one long-lived wave
// per CU that keeps its SIMD's vector ALU nearly busy and, being the oldest wave there, wins every issue slot it wants); a
// workgroup of four waves would hold its slot until its wave on that SIMD is through (measured: 2.2 x the kernel time).
template <int T, int D>
__global__ __launch_bounds__(64) void chs_fe_exact_kernel(const uint8_t *__restrict__ slots, long slot_bytes, int hist_bytes,
                                                            long k_lo, long n_if, long ntiles, long wgs_per_channel,
                                                            const float *__restrict__ table, float *__restrict__ demod,
                                                            long dpitch, int Hd)
{
    using C = FeX<T, D>;
    constexpr int R = kR;
    const int lane = threadIdx.x & 63;
    const long c = blockIdx.x / wgs_per_channel;
    const long tile = blockIdx.x % wgs_per_channel;
    if (tile >= ntiles) return;                                  // wave-uniform
    const uint8_t *blk = slots + c * slot_bytes + hist_bytes;
    const long k0 = k_lo + tile * C::TILE + static_cast<long>(lane - 1) * R;   // outputs [k_lo, n_if) of the block: this launch's share
    const long w0 = k0 * D - (T - 1) - C::LEAD;                  // first sample of the 16-byte aligned window
    const u4 *src = reinterpret_cast<const u4 *>(blk + 2 * w0);
    uint32_t raw[C::NB * 4];
    // requested newest chunk first: that is the order the steps consume them in, so the arithmetic starts when the first load is back
#pragma unroll
    for (int i = C::NB - 1; i >= 0; i--) {
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

// The taps of consecutive steps overlap: step u needs the pairs P[u .. u+7] of the padded table P[m] = (h_st, h_car)[m - (R-1)]
// (zero outside the filter), i.e. one NEW pair per step.  They are fetched in groups of eight (one s_load_dwordx16 per EIGHT
// steps) into three rotating SGPR groups: at the start of block g the groups g and g+1 are resident and g+2 is requested --
// scalar loads return out of order, so every wait is for all of them, and a load requested eight steps (~500 cycles) ahead
// has long arrived (one load per step, waited for a step later, left the vector ALU idle half the time).
template <int T, bool EXACT, int G>
__device__ __forceinline__ void bpx_block(const float (&w)[BpX<T>::NW], const float *__restrict__ table, f2 (&acc)[kR], f16v &g0,
                                          f16v &g1, f16v &g2)
{
    using C = BpX<T>;
    constexpr int NBLK = (C::W + 7) / 8;
    if constexpr (G < NBLK) {
        // groups G (cur), G + 1 (next) resident after this wait; G + 2 requested behind it into the group G - 1 used
        f16v &cur = G % 3 == 0 ? g0 : (G % 3 == 1 ? g1 : g2);
        f16v &nxt = (G + 1) % 3 == 0 ? g0 : ((G + 1) % 3 == 1 ? g1 : g2);
        f16v &fut = (G + 2) % 3 == 0 ? g0 : ((G + 2) % 3 == 1 ? g1 : g2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(cur), "+s"(nxt));
        if constexpr (G + 2 <= NBLK) CHS_TAPS_ISSUE(fut, table, (G + 2) * 64);
        float hc[16], hn[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            hc[k] = cur[k];
            hn[k] = nxt[k];
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int u = 8 * G + e;
            if (u < C::W) {
                const int i = C::W - 1 - u;
                const float x = w[i + C::LEAD];
#pragma unroll
                for (int r = 0; r < kR; r++) {
                    const int n = r + (T - 1) - i;
                    if (n >= 0 && n < T) {
                        const int m = e + r;                       // pair P[8 G + m]: this group for m < 8, the next one after
                        const float hs = m < 8 ? hc[2 * m] : hn[2 * (m - 8)], hcar = m < 8 ? hc[2 * m + 1] : hn[2 * (m - 8) + 1];
                        if constexpr (EXACT) {
                            const f2 prod = (f2){x, x} * (f2){hs, hcar};
                            acc[r] = acc[r] + prod;
                        } else {   // the fast bank: one fused multiply-add per tap (float32-rounding-equal, as the single-stream kernels)
                            acc[r] = __builtin_elementwise_fma((f2){x, x}, (f2){hs, hcar}, acc[r]);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < kR; r++) asm volatile("" : "+v"(acc[r]));
        }
        bpx_block<T, EXACT, G + 1>(w, table, acc, g0, g1, g2);
    }
}

template <int T, bool EXACT>
__global__ __launch_bounds__(64) void chs_bpf_kernel(const float *__restrict__ demod, long dpitch, int Hd, long k_lo, long n_if,
                                                             long wgs_per_channel, const float *__restrict__ table,
                                                             float *__restrict__ y_st, float *__restrict__ y_car, long ypitch,
                                                             int8_t *__restrict__ y_car8, long cpitch)
{
    using C = BpX<T>;
    constexpr int R = kR;
    const long c = blockIdx.x / wgs_per_channel;
    const long k0 = k_lo + ((blockIdx.x % wgs_per_channel) * 64 + threadIdx.x) * R;
    if (k0 >= n_if) return;
    const float *x = demod + c * dpitch + Hd;
    const f4 *src = reinterpret_cast<const f4 *>(x + k0 - (T - 1) - C::LEAD);
    float w[C::NW];
#pragma unroll
    for (int i = C::NW / 4 - 1; i >= 0; i--) {                  // newest chunk first: the order the steps consume them in
        const f4 v = src[i];
        w[4 * i] = v.x;
        w[4 * i + 1] = v.y;
        w[4 * i + 2] = v.z;
        w[4 * i + 3] = v.w;
    }
    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
    f16v g0, g1, g2;
    CHS_TAPS_ISSUE(g0, table, 0);
    CHS_TAPS_ISSUE(g1, table, 64);
    bpx_block<T, EXACT, 0>(w, table, acc, g0, g1, g2);
    float *ds = y_st + c * ypitch + k0;
    if constexpr (!EXACT) {
        // fast bank: the pilot band-pass output only feeds the PLL's fast recurrence, which reads its SIGN: one signed byte per sample
        // (+1 / -1; 0 = not an ordinary sample: zero, denormal-small or not finite -- kernels_pll.hip: pll_ordinary)
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float v = acc[r].y;
            const bool ord = fabsf(v) > 1e-20f && fabsf(v) < 1e20f;
            const uint32_t code = ord ? (v > 0.0f ? 0x01u : 0xffu) : 0u;
            if (r < 4) lo |= code << (8 * r);
            else hi |= code << (8 * (r - 4));
        }
        int8_t *dc8 = y_car8 + c * cpitch + k0;
        if (k0 + R <= n_if) {
            reinterpret_cast<f4 *>(ds)[0] = (f4){acc[0].x, acc[1].x, acc[2].x, acc[3].x};
            reinterpret_cast<f4 *>(ds)[1] = (f4){acc[4].x, acc[5].x, acc[6].x, acc[7].x};
            *reinterpret_cast<uint2 *>(dc8) = make_uint2(lo, hi);
        } else {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (k0 + r < n_if) {
                    ds[r] = acc[r].x;
                    dc8[r] = static_cast<int8_t>(((r < 4 ? lo : hi) >> (8 * (r % 4))) & 0xff);
                }
        }
    } else {
        float *dc = y_car + c * ypitch + k0;
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
}

// ---- the NCO output ----------------------------------------------------------------------------------------------------
// PLL[k+1] = cosf(trigArg[k]*ncoScale + phaseAdjust) (src/filter.cpp:72) is not on the recurrence's chain: the lanes leave the
// raw trigArg of every step, this kernel turns a chunk's row segment into finished NCO values in place, four per thread
// (glibc's cosf: the branch-free form when the whole wave's arguments are ordinary, i.e. always but in a stream's first 120 samples).
template <bool EXACT>
__global__ __launch_bounds__(256) void chs_nco_kernel(float *__restrict__ trig, long ypitch, long k_lo, long k_hi, long wgs_per_channel,
                                                             float nco_scale, float phase_adjust, const float *__restrict__ bpf,
                                                             const float *__restrict__ nco0, float *__restrict__ mixer, long mpitch, int hm)
{
    __shared__ uint32_t w24[24];
    if (threadIdx.x < 24) w24[threadIdx.x] = glibc235::inv_pio4(threadIdx.x);
    __syncthreads();
    const long c = blockIdx.x / wgs_per_channel;
    const long k = k_lo + ((blockIdx.x % wgs_per_channel) * 256 + threadIdx.x) * 4;
    if (k >= k_hi) return;
    f4 *p = reinterpret_cast<f4 *>(trig + c * ypitch + k);
    const f4 v = *p;
    float a[4], o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = v[i] * nco_scale + phase_adjust;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; i++) ok = ok && glibc235::sincosf_large_ok(a[i]);
    if (!EXACT) {   // the fast bank: one argument reduction in double, the hardware cosine (kernels_pll.hip: nco_out<kFast>)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double rev = static_cast<double>(a[i]) * 0.15915494309189533577;
            o[i] = __builtin_amdgcn_cosf(static_cast<float>(rev - rint(rev)));
        }
    } else if (!__any(!ok)) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float sn;
            glibc235::sincosf_large_flat(a[i], w24, &sn, &o[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = glibc235::cosf_glibc(a[i]);
    }
    if (k + 4 <= k_hi) {
        *p = (f4){o[0], o[1], o[2], o[3]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (k + i < k_hi) trig[c * ypitch + k + i] = o[i];
    }
    // The resampling modes read the mixer output (src/project.cpp:246-248: (stereo_filt * PLL) * 2, PLL[g] = the NCO value of step
    // g - 1, PLL[0] = the incoming state's lastOut) as a row of its own (chs_resample_*): written here, rows [hm | n_if], history
    // carried by the finish kernel.  Entry k_lo takes the NCO value the previous chunk's launch left in trig[k_lo - 1].
    if (mixer) {
        const float *bp = bpf + c * ypitch;
        float *mx = mixer + c * mpitch + hm;
        if (k == k_lo) mx[k] = (bp[k] * (k > 0 ? trig[c * ypitch + k - 1] : nco0[c])) * 2.0f;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (k + 1 + i < k_hi) mx[k + 1 + i] = (bp[k + 1 + i] * o[i]) * 2.0f;
    }
}

// ---- everything behind the PLL ------------------------------------------------------------------------------------
// The mixer (src/project.cpp:246-248), the two audio convolveBlockFastFIR calls of RF_STEREO -- mono branch on the
// all-passed discriminator output (:194, :219), stereo branch on the mixer output (:257) --, the L/R combine (:277-280)
// and the interleaved PCM writer (:292-302), in the reference's evaluation order.  A single-wave workgroup stages the
// window of its 64*R audio outputs as (mono, mixer) pairs in LDS -- the mixer products are formed while staging and never
// go to HBM; the two FIRs share the taps, so they ride in the two halves of the packed instructions.  The FIR is the
// front end's scheme once more: a thread owns R = kRO ADJACENT outputs and visits its window newest sample first, one
// ds_read_b64 per sample feeds every output the sample belongs to (R = 4: 116 LDS reads per 404 tap-output pairs; the
// one-read-per-tap form is bound by LDS bandwidth at twice the time), the taps of two steps are one scalar load.
// Neighbouring lanes' windows start R*D pairs apart; the LDS index j + j / (R*D) makes that an odd number of
// pairs (R = 4: 21 / 25): conflict-free ds_read_b64.  Mono banks (STEREO = false) run the same kernel on the mono half alone.
constexpr int kRO = 4;   // adjacent audio outputs per thread of the output stage (8: 512 outputs and 22 KB of LDS per workgroup -- 7 single-wave
                         // workgroups per CU, whose staging latency then sets the kernel's time; 4: 256 outputs, 11 KB, 14 per CU)
template <int T, int D>
struct OutX {
    static constexpr int R = kRO;
    static constexpr int SPL = 16 / R;                          // steps per scalar load of 16 taps
    static constexpr int W = D * (R - 1) + T;                   // window samples (= steps) per thread
    static constexpr int NG = (W + SPL - 1) / SPL;              // scalar-load groups
    static constexpr int NOUT = 64 * R;                         // audio outputs per workgroup
    static constexpr int WL = D * (NOUT - 1) + T;               // window samples per workgroup
    static constexpr int SEG = R * D;                           // window samples between neighbouring lanes
    static constexpr int LDSN = WL + WL / SEG + 2;              // pairs in LDS
    __host__ __device__ static constexpr int idx(int j) { return j + j / SEG; }
};

template <int T, int D, bool STEREO, bool EXACT, int G>
__device__ __forceinline__ void outx_step(const f2 *__restrict__ wl, const float *__restrict__ table, f2 (&acc)[kRO], f16v &hA, f16v &hB)
{
    using C = OutX<T, D>;
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
        for (int e = 0; e < C::SPL; e++) {
            const int u = C::SPL * G + e;
            if (u < C::W) {
                const int jj = C::W - 1 - u;                    // window sample of this thread, newest first
                const f2 w = wl[C::idx(jj)];                    // lane base + compile-time offset
#pragma unroll
                for (int r = 0; r < kRO; r++) {
                    const int n = r * D + (T - 1) - jj;
                    if (n >= 0 && n < T) {
                        const float h = hq[kRO * e + r];
                        if constexpr (!EXACT) {
                            acc[r] = __builtin_elementwise_fma(w, (f2){h, h}, acc[r]);
                        } else if constexpr (STEREO) {
                            const f2 prod = w * (f2){h, h};
                            acc[r] = acc[r] + prod;
                        } else {
                            const float prod = w.x * h;
                            acc[r].x = acc[r].x + prod;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kRO; r++) asm volatile("" : "+v"(acc[r]));
        outx_step<T, D, STEREO, EXACT, G + 1>(wl, table, acc, hA, hB);
    }
}

template <int T, int D, bool STEREO, bool EXACT>
__global__ __launch_bounds__(64) void chs_out_kernel(const float *__restrict__ demod, long dpitch, int Hd, const float *__restrict__ bpf,
                                                            const float *__restrict__ nco, long ypitch, const float *__restrict__ nco0,
                                                            const float *__restrict__ mix_tail_in, float *__restrict__ mix_tail_out, int hm,
                                                            long n_if, long g_hi, int delay, float nco_scale, float phase_adjust,
                                                            const float *__restrict__ table, long wgs_per_channel, float *__restrict__ audio,
                                                            int16_t *__restrict__ pcm, int wrap, long a_lo, long a_hi, long n_out)
{
    using C = OutX<T, D>;
    constexpr int R = kRO;
    __shared__ f2 win[C::LDSN];
    const int t = threadIdx.x;
    const long c = blockIdx.x / wgs_per_channel;
    // this launch: audio outputs [a_lo, a_hi) of the block's n_out, from the IF samples [.., g_hi) that exist by now (a block is
    // walked in chunks so that the PLL's lanes of one chunk run next to the wide kernels of the next: stereo_bank_process_dev)
    const long a0 = a_lo + (blockIdx.x % wgs_per_channel) * C::NOUT;
    const long g0 = D * a0 - (T - 1);                      // IF index of window sample 0
    const float *dm = demod + c * dpitch + Hd;
    const float *bp = STEREO ? bpf + c * ypitch : nullptr, *nc = STEREO ? nco + c * ypitch : nullptr;
    const float *tin = STEREO ? mix_tail_in + c * hm : nullptr;
    float *tout = STEREO ? mix_tail_out + c * hm : nullptr;
    const float first = STEREO ? nco0[c] : 0.0f;
    // D-1 samples past the window are visited too: when the block ends exactly on a tile boundary nobody's window reaches
    // the block's last D-1 samples, and they belong to the tail this call leaves behind
    // Staging in batches of B samples per lane, the next batch's three loads per sample in flight while this one is turned into
    // LDS pairs.  EXACT: `nco` holds finished NCO values (chs_nco_kernel); fast bank: the raw trigArg of the PLL's steps, and the
    // cosine is taken here (one argument reduction in double + the hardware cosine, kernels_pll.hip: nco_out<kFast>).
    constexpr int NJ = (C::WL + D - 1 + 63) / 64, B = kRO == 4 ? 11 : 11, NBATCH = (NJ + B - 1) / B;
    float vm[2][B], va[2][B], vb[2][B];
    auto fetch = [&](int q0, float (&m)[B], float (&a)[B], float (&b)[B]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < B; q++) {
            const int j = t + (q0 + q) * 64;
            const long g = g0 + j;
            m[q] = 0.0f;
            a[q] = 0.0f;
            b[q] = 0.0f;
            if (j < C::WL + D - 1 && g < g_hi) {
                m[q] = dm[g - delay];                          // history in front of the row: negative indices are valid
                if (STEREO) {
                    a[q] = g >= 0 ? bp[g] : tin[hm + g];
                    if (g > 0) b[q] = nc[g - 1];               // PLL[g] = cos(trigArg[g-1] * ncoScale + phaseAdjust)
                }
            }
        }
    };
    fetch(0, vm[0], va[0], vb[0]);
#pragma unroll
    for (int bi = 0; bi < NBATCH; bi++) {
        if (bi + 1 < NBATCH) fetch((bi + 1) * B, vm[(bi + 1) & 1], va[(bi + 1) & 1], vb[(bi + 1) & 1]);
#pragma unroll
        for (int q = 0; q < B; q++) {
            const int j = t + (bi * B + q) * 64;
            const long g = g0 + j;
            if (j >= C::WL + D - 1) continue;
            float m = 0.0f, x = 0.0f;
            if (g < g_hi) {
                m = vm[bi & 1][q];
                if (STEREO) {
                    if (g >= 0) {
                        float pll = vb[bi & 1][q];
                        if (!EXACT) {
                            const double rev = static_cast<double>(pll * nco_scale + phase_adjust) * 0.15915494309189533577;
                            pll = __builtin_amdgcn_cosf(static_cast<float>(rev - rint(rev)));
                        }
                        if (g == 0) pll = first;               // PLL[0] = the incoming state's lastOut
                        x = (va[bi & 1][q] * pll) * 2.0f;      // the reference's order: (stereo_filt * PLL) * 2
                        if (g >= n_if - hm) tout[g - (n_if - hm)] = x;
                    } else {
                        x = va[bi & 1][q];
                    }
                }
            }
            if (j < C::WL) win[C::idx(j)] = (f2){m, x};
        }
    }
    __syncthreads();
    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
    f16v hA, hB;
    CHS_TAPS_ISSUE(hA, table, 0);
    outx_step<T, D, STEREO, EXACT, 0>(win + (C::SEG + 1) * t, table, acc, hA, hB);
    const long k0 = a0 + static_cast<long>(t) * R;
    if (k0 >= a_hi) return;
    if constexpr (STEREO) {
        float l[R], rr[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            l[r] = acc[r].y + acc[r].x;                        // src/project.cpp:278-279: left = stereo + mono, right = mono - stereo
            rr[r] = acc[r].x - acc[r].y;
        }
        if (k0 + R <= a_hi) {
            if (audio) {
                f4 *pl = reinterpret_cast<f4 *>(audio + c * 2 * n_out + k0), *pr = reinterpret_cast<f4 *>(audio + c * 2 * n_out + n_out + k0);
#pragma unroll
                for (int q = 0; q < R / 4; q++) {
                    pl[q] = (f4){l[4 * q], l[4 * q + 1], l[4 * q + 2], l[4 * q + 3]};
                    pr[q] = (f4){rr[4 * q], rr[4 * q + 1], rr[4 * q + 2], rr[4 * q + 3]};
                }
            }
            if (pcm) {
                using s8v = short __attribute__((ext_vector_type(8)));
                s8v *pp = reinterpret_cast<s8v *>(pcm + 2 * (c * n_out + k0));
#pragma unroll
                for (int q = 0; q < R / 4; q++)
                    pp[q] = (s8v){pcm_pack_flat(l[4 * q], wrap), pcm_pack_flat(rr[4 * q], wrap), pcm_pack_flat(l[4 * q + 1], wrap),
                                  pcm_pack_flat(rr[4 * q + 1], wrap), pcm_pack_flat(l[4 * q + 2], wrap), pcm_pack_flat(rr[4 * q + 2], wrap),
                                  pcm_pack_flat(l[4 * q + 3], wrap), pcm_pack_flat(rr[4 * q + 3], wrap)};
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (k0 + r < a_hi) {
                    if (audio) {
                        audio[c * 2 * n_out + k0 + r] = l[r];
                        audio[c * 2 * n_out + n_out + k0 + r] = rr[r];
                    }
                    if (pcm) {
                        pcm[2 * (c * n_out + k0 + r)] = pcm_pack_flat(l[r], wrap);
                        pcm[2 * (c * n_out + k0 + r) + 1] = pcm_pack_flat(rr[r], wrap);
                    }
                }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; r++)
            if (k0 + r < a_hi) {
                if (audio) audio[c * n_out + k0 + r] = acc[r].x;
                if (pcm) pcm[c * n_out + k0 + r] = pcm_pack_flat(acc[r].x, wrap);
            }
    }
}

// ---- the resampling modes (2, 3) behind the PLL, reference order ----------------------------------------------------------
// RF_STEREO with audio_upsamp > 0 (src/project.cpp:224-231, 262-269): convolveBlockResampleFIR (src/filter.cpp:191-223) on the
// all-passed discriminator output and on the mixer output.  In stream form: output k reads phase ph = (k D) mod U of the taps and
// the input samples n_k, n_k - 1, ... with n_k = (k D - ph) / U:  y = sum_j h[ph + j U] x[n_k - j]  (separately rounded products
// and sums, j ascending), then y += y * U.  A block's outputs restart at phase 0 (the bank accepts blocks with n_if U % D == 0, as
// the reference's own block sizes are).  The mixer output is materialised by the chunk's NCO pass (chs_nco_kernel; rows [Hm | n_if],
// history carried by the finish kernel): the same row serves every output's window.  chs_resample_exact_kernel: one thread per (channel, output) -- any ratio,
// and the outputs chs_resample_lanes_kernel (below: the reference's two ratios at 5-7 x its speed) leaves over.
template <bool STEREO>
__global__ void chs_resample_exact_kernel(const float *__restrict__ demod, long dpitch, int Hd, const float *__restrict__ mixer, long mpitch,
                                          int hm, int delay, const float *__restrict__ h, int taps, int decim, int upsamp,
                                          long wgs_per_channel, float *__restrict__ audio, int16_t *__restrict__ pcm, int wrap, long a_lo,
                                          long a_hi, long n_out)
{
    const long c = blockIdx.x / wgs_per_channel;
    const long k = a_lo + (blockIdx.x % wgs_per_channel) * 256 + threadIdx.x;
    if (k >= a_hi) return;
    const long long m = static_cast<long long>(k) * decim;
    const int ph = static_cast<int>(m % upsamp);
    const long n0 = static_cast<long>((m - ph) / upsamp);
    const float *xm = demod + c * dpitch + Hd + n0 - delay;                    // the all-pass is an index offset
    const float *xs = STEREO ? mixer + c * mpitch + hm + n0 : nullptr;
    float am = 0.0f, as = 0.0f;
    int j = 0;
    for (int n = ph; n < taps; n += upsamp, j++) {
        const float hn = h[n];
        const float pm = hn * xm[-j];
        am = am + pm;
        if (STEREO) {
            const float ps = hn * xs[-j];
            as = as + ps;
        }
    }
    const float gm = am * static_cast<float>(upsamp);                          // y += y * U  (src/filter.cpp:213)
    const float mono = am + gm;
    if (STEREO) {
        const float gs = as * static_cast<float>(upsamp);
        const float st = as + gs;
        const float l = st + mono, r = mono - st;                              // src/project.cpp:278-279
        if (audio) {
            audio[c * 2 * n_out + k] = l;
            audio[c * 2 * n_out + n_out + k] = r;
        }
        if (pcm) {
            pcm[2 * (c * n_out + k)] = pcm_pack_flat(l, wrap);
            pcm[2 * (c * n_out + k) + 1] = pcm_pack_flat(r, wrap);
        }
    } else {
        if (audio) audio[c * n_out + k] = mono;
        if (pcm) pcm[c * n_out + k] = pcm_pack_flat(mono, wrap);
    }
}

// The same resampler with the bank turned the other way: a lane is a CHANNEL.  Output k's phase and window position depend on k
// alone, so for 64 channels at once the taps are wave-uniform -- SGPR operands from a step-major table, as in the exact FIRs
// above -- and only the samples are per lane.  A wave owns kRS = 7 consecutive outputs of 64 channels at a time (both modes' U are
// multiples of 7: groups never straddle a period, the table has U / 7 groups) and visits their common window newest sample first:
// output r meets its taps j ascending (the reference's order, src/filter.cpp:205-212) while one sample per lane and step feeds all
// seven outputs; (mono, stereo) -- in mono banks two neighbouring outputs -- ride in the two halves of the packed instructions.
// Samples reach the lanes through LDS, 32 per channel and batch (below: why not lane = channel for the loads too); the taps of
// four steps are two s_load_dwordx16, requested one iteration ahead.  Outside a group's window the table holds zeros: acc + 0*x
// leaves acc as it is (acc is never -0: it starts at +0 and sums round to nearest), rows are finite, and the rows' histories
// are long enough for the window's rounding up to whole batches (StereoBank::res_hist).  Against one thread per (channel,
// output) -- 101 gathered taps and 101-202 gathered samples per output, bound by the texture addressers -- 16 384 receivers x 4
// blocks of mode 2: stereo 12 -> 3.2 ms per call, mono 9 -> 1.7 ms (bit-identical outputs).  What bounds it now is the traffic of
// re-reading the windows: neighbouring groups' windows overlap (160 samples per 38 of advance), and 64 channels' windows of 2-3
// waves per SIMD do not fit L2.  The kRW = 3 (mono: 7) waves of a workgroup therefore take NEIGHBOURING groups at the same time
// (each with its own staging area, no barrier): their loads meet in L1 / L2 (mono fast, mode 2: 3.6 -> 3.4 ms per call with 3 waves, 3.0
// with 7; stereo banks stage twice as much per wave and 7 waves leave one workgroup per CU: no gain over 3).
constexpr int kRS = 7;
// waves per workgroup: they take neighbouring groups (overlapping windows) at the same time; mono banks stage half as much per wave
template <bool STEREO> constexpr int kRW = STEREO ? 3 : 7;

template <bool STEREO, bool EXACT>
__global__ __launch_bounds__(64 * kRW<STEREO>) void chs_resample_lanes_kernel(const float *__restrict__ demod, long dpitch, int Hd, const float *__restrict__ mixer,
                                                                 long mpitch, int hm, int delay, const float *__restrict__ table,
                                                                 const int *__restrict__ top_of, int groups_per_period, int groups_per_wave,
                                                                 int iters, int decim, int upsamp, long periods_per_row, int n_channels,
                                                                 float *__restrict__ audio, int16_t *__restrict__ pcm, int wrap, long a_lo,
                                                                 long n_out)
{
    // workgroup (one wave) = (64 channels, period a_lo / U + `period`, groups [g_lo, g_hi) of that period)
    // Grid order: channel group fastest, then period, the part of the period slowest -- the waves resident at any moment walk the
    // SAME groups, whose taps (5 KB per group) then stay in the scalar cache; the whole table (100-300 KB) does not fit there, and
    // with the part fastest every tap load went to L2 (517 -> 455 us per chunk)
    const long n_cg = (n_channels + 63) / 64;
    const long cg = blockIdx.x % n_cg;
    const long rest = blockIdx.x / n_cg;
    const long period = rest % periods_per_row;
    const int g_lo = static_cast<int>(rest / periods_per_row) * groups_per_wave;
    const int g_hi = g_lo + groups_per_wave < groups_per_period ? g_lo + groups_per_wave : groups_per_period;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const long c_raw = cg * 64 + lane;
    const long c = c_raw < n_channels ? c_raw : n_channels - 1;     // spare lanes repeat the last channel (nothing stored)
    const long k_base = a_lo + period * upsamp;                      // first output of this period (a multiple of U: phase 0)
    const long x_base = k_base / upsamp * decim;                     // its window position
    const float fu = static_cast<float>(upsamp);
    typedef float q4 __attribute__((ext_vector_type(4), aligned(4)));   // four consecutive samples of a row, any alignment
    // Samples reach the lanes through LDS, a batch of kSB = 32 per channel at a time.  Loading them lane = channel (each lane its own
    // row) makes every load instruction touch 64 cache lines for 1 KiB: the texture addressers, not the ALUs, then set the kernel's
    // time (804 -> 517 us per chunk).  Instead 8 lanes share a channel's 128 bytes (instruction i: channels 8 i + lane / 8, quad lane % 8):
    // 8-16 lines per instruction; the quads go to LDS as [sample][channel] rows of 66 floats (lanes of one write hit banks
    // channel + 8 quad + 2 j: all different), and a step reads its sample for the lane's OWN channel back: consecutive banks.
    constexpr int kSB = 32, kLP = 66;
    __shared__ float lds_all[kRW<STEREO> * (STEREO ? 2 : 1) * kSB * kLP];
    float *lds = lds_all + wv * ((STEREO ? 2 : 1) * kSB * kLP);   // a wave's own staging area: no barriers
    const int sub = lane >> 3, quad = lane & 7;
    long cl[8];                                                      // the channels this lane loads for (instruction i)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const long cc = cg * 64 + 8 * i + sub;
        cl[i] = cc < n_channels ? cc : n_channels - 1;
    }
    const long m_off = Hd - delay + x_base, s_off = hm + x_base;     // the all-pass is an index offset
    for (int g = g_lo + wv; g < g_hi; g += kRW<STEREO>) {
        const int top = __builtin_amdgcn_readfirstlane(top_of[g]);  // newest sample of the group's window, relative to x_base
        const uint32_t tbase = static_cast<uint32_t>(g) * static_cast<uint32_t>(iters) * 128u;   // bytes: 4 steps x 8 floats per iteration
        f2 acc[kRS];
#pragma unroll
        for (int r = 0; r < kRS; r++) acc[r] = (f2){0.0f, 0.0f};
        f16v ta, tb, ua, ub;
        asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(ta) : "s"(table), "s"(tbase));
        asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(tb) : "s"(table), "s"(tbase + 64u));
        q4 gm[8], gs[8];                                             // batch in flight: 8 channels' quads per row
        // batch bt = samples top - 32 bt - 31 ... top - 32 bt of every channel
        auto fetch = [&](int bt) __attribute__((always_inline)) {
            const long first = static_cast<long>(top) - kSB * bt - (kSB - 1) + 4 * quad;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                gm[i] = *reinterpret_cast<const q4 *>(demod + cl[i] * dpitch + m_off + first);
                if (STEREO) gs[i] = *reinterpret_cast<const q4 *>(mixer + cl[i] * mpitch + s_off + first);
            }
        };
        auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    lds[(4 * quad + j) * kLP + 8 * i + sub] = gm[i][j];
                    if (STEREO) lds[kSB * kLP + (4 * quad + j) * kLP + 8 * i + sub] = gs[i][j];
                }
        };
        auto four_steps = [&](int t0, const f16v &h0, const f16v &h1) __attribute__((always_inline)) {
            float hq[32];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                hq[q] = h0[q];
                hq[16 + q] = h1[q];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int sidx = kSB - 1 - (t0 + e);                 // step t of the batch <-> its sample 31 - t (newest first)
                const float xm = lds[sidx * kLP + lane];
                const float xs = STEREO ? lds[kSB * kLP + sidx * kLP + lane] : 0.0f;
                const f2 w = (f2){xm, xs};
                if constexpr (STEREO) {                              // (mono, stereo) of output r in the two halves
#pragma unroll
                    for (int r = 0; r < kRS; r++) {
                        const float h = hq[8 * e + r];
                        if constexpr (!EXACT) {
                            acc[r] = __builtin_elementwise_fma(w, (f2){h, h}, acc[r]);
                        } else {
                            const f2 prod = w * (f2){h, h};
                            acc[r] = acc[r] + prod;
                        }
                    }
                } else {                                             // mono: outputs 2 q, 2 q + 1 in the two halves (the table's eighth tap is 0)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f2 hh = (f2){hq[8 * e + 2 * q], hq[8 * e + 2 * q + 1]};
                        if constexpr (!EXACT) {
                            acc[q] = __builtin_elementwise_fma((f2){xm, xm}, hh, acc[q]);
                        } else {
                            const f2 prod = (f2){xm, xm} * hh;
                            acc[q] = acc[q] + prod;
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < kRS; r++) asm volatile("" : "+v"(acc[r]));
        };
        const int nb = iters / 8;                                    // batches per group (`iters` is a multiple of 8: host)
        fetch(0);
        for (int bt = 0; bt < nb; bt++) {
            stage();                                                 // (the previous batch's reads are done: one wave, program order)
            if (bt + 1 < nb) fetch(bt + 1);                          // in flight while this batch is multiplied
            const uint32_t tb0 = tbase + static_cast<uint32_t>(bt) * 1024u;   // 8 iterations x 128 bytes
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                // the taps of iteration i + 1 are requested while iteration i runs (the last request of a group reads the next
                // group's first iteration or the table's pad)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ta), "+s"(tb));
                asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(ua) : "s"(table), "s"(tb0 + static_cast<uint32_t>(i + 1) * 128u));
                asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(ub) : "s"(table), "s"(tb0 + static_cast<uint32_t>(i + 1) * 128u + 64u));
                four_steps(4 * i, ta, tb);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ua), "+s"(ub));
                asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(ta) : "s"(table), "s"(tb0 + static_cast<uint32_t>(i + 2) * 128u));
                asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(tb) : "s"(table), "s"(tb0 + static_cast<uint32_t>(i + 2) * 128u + 64u));
                four_steps(4 * i + 4, ua, ub);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ta), "+s"(tb));   // the look-ahead's loads land before the registers are reused
        if (c_raw < n_channels) {
            const long k0 = k_base + static_cast<long>(g) * kRS;
#pragma unroll
            for (int r = 0; r < kRS; r++) {
                const float am = STEREO ? acc[r].x : acc[r / 2][r % 2];
                const float gm = am * fu;                            // y += y * U  (src/filter.cpp:213)
                const float mono = am + gm;
                if constexpr (STEREO) {
                    const float gs = acc[r].y * fu;
                    const float st = acc[r].y + gs;
                    const float l = st + mono, rr = mono - st;       // src/project.cpp:278-279
                    if (audio) {
                        audio[c * 2 * n_out + k0 + r] = l;
                        audio[c * 2 * n_out + n_out + k0 + r] = rr;
                    }
                    if (pcm) {
                        pcm[2 * (c * n_out + k0 + r)] = pcm_pack_flat(l, wrap);
                        pcm[2 * (c * n_out + k0 + r) + 1] = pcm_pack_flat(rr, wrap);
                    }
                } else {
                    if (audio) audio[c * n_out + k0 + r] = mono;
                    if (pcm) pcm[c * n_out + k0 + r] = pcm_pack_flat(mono, wrap);
                }
            }
        }
    }
}

// carried state: per channel, history <- the slot's last hist_bytes bytes; demod history <- the row's last Hd samples
// (create() rejects blocks shorter than either history, so source and destination never overlap)
__global__ void chs_finish_kernel(uint8_t *__restrict__ slots, long slot_bytes, long hist_bytes, float *__restrict__ demod,
                                  long dpitch, int Hd, long n_if, float *__restrict__ mixer, long mpitch, int hm)
{
    const long c = blockIdx.x;
    uint8_t *slot = slots + c * slot_bytes;
    const u4 *src = reinterpret_cast<const u4 *>(slot + slot_bytes - hist_bytes);
    u4 *dst = reinterpret_cast<u4 *>(slot);
    for (long i = threadIdx.x; i < hist_bytes / 16; i += blockDim.x) dst[i] = src[i];
    float *row = demod + c * dpitch;
    for (long i = threadIdx.x; i < Hd; i += blockDim.x) row[i] = row[n_if + i];
    if (mixer) {                                                   // resampling modes: state_stereofilt = the mixer row's tail
        float *mr = mixer + c * mpitch;
        for (long i = threadIdx.x; i < hm; i += blockDim.x) mr[i] = mr[n_if + i];
    }
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
    DevBuf<float> fe_table, bpf_table, out_table;
    FePlan fe;                      // fast banks: the matrix-core front end's tap image
    DevBuf<float> st_img, car_img;  // fast banks: the band-pass filters' Toeplitz images for the f32 matrix cores (fe_bpf_bank_kernel)
    bool fused_front = false;       // front end + band-pass pair in one kernel
    Options opt;
    DevBuf<float> demod, carrier, bpf, trig, pll, nco0, mixtail[2];
    DevBuf<int8_t> carrier8;        // fast banks: the sign of the pilot band-pass output, one byte per IF sample
    long cpitch = 0;
    // resampling modes (2, 3): the plain audio taps and, stereo, the mixer rows [Hm | n_if]
    bool resample = false;
    DevBuf<float> h_res, mixer;
    long mpitch = 0;
    // ... and, when U is a multiple of 7 (both of the reference's), the step-major tap table of chs_resample_lanes_kernel
    DevBuf<float> res_table;
    DevBuf<int> res_top;
    int res_groups = 0, res_iters = 0, res_hist = 0;   // res_hist: samples of history its windows reach (>= Ha, by the rounding to whole iterations)
    int mix_cur = 0;
    // A stereo call walks the block in chunks on two internal streams: `wide` carries the front end, the band-pass pair and the
    // output stage of every chunk, `lanes` the PLL -- the PLL's few waves (one per 64 channels, a dependent chain each) leave
    // the chip almost idle, and chunk c+1's wide kernels fill it meanwhile.  Events: chunk c's band-pass output is ready
    // (wide -> lanes), its PLL is through (lanes -> wide); fork / join with the caller's stream around the call.
    static constexpr int kMaxChunks = 8;
    int max_chunks = kMaxChunks;
    hipStream_t wide = nullptr, lanes = nullptr;
    hipStream_t front = nullptr;    // fast banks: the HBM-bound front end runs on its own stream, next to the vector-ALU-bound kernels
    hipStream_t post = nullptr;     // fast banks, option bank_streams = 4: the output stage on a stream of its own too
    hipEvent_t ev_bpf[kMaxChunks] = {}, ev_pll[kMaxChunks] = {}, ev_fe[kMaxChunks] = {}, ev_fork = nullptr, ev_join = nullptr;
    ~StereoBank()
    {
        for (hipStream_t st : {wide, lanes, front, post})
            if (st) {
                (void)hipStreamSynchronize(st);
                (void)hipStreamDestroy(st);
            }
        for (auto &e : ev_bpf)
            if (e) (void)hipEventDestroy(e);
        for (auto &e : ev_pll)
            if (e) (void)hipEventDestroy(e);
        for (auto &e : ev_fe)
            if (e) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }
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
    // P[m] = (h_st, h_car)[m - (R-1)], zero outside the filter, in groups of eight pairs; two groups of padding behind the last block
    constexpr int NBLK = (C::W + 7) / 8;
    std::vector<float> tab(static_cast<size_t>(NBLK + 2) * 16, 0.0f);
    for (int m = 0; m < (NBLK + 2) * 8; m++) {
        const int n = m - (kR - 1);
        if (n >= 0 && n < T) {
            tab[2 * static_cast<size_t>(m)] = h_st[n];
            tab[2 * static_cast<size_t>(m) + 1] = h_car[n];
        }
    }
    FMRX_TRY(b.bpf_table.alloc(tab.size()));
    FMRX_HIP(hipMemcpy(b.bpf_table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return FMRX_OK;
}

#define CHS_FE_CASES(X) X(101, 10) X(101, 5) X(101, 3) X(151, 10) X(151, 5) X(151, 3) X(13, 10) X(13, 5) X(13, 3)
#define CHS_BPF_CASES(X) X(101) X(151) X(13)
#define CHS_OUT_CASES(X) X(101, 5) X(101, 6) X(13, 5) X(13, 6)

// IF outputs [k_lo, k_hi) of every channel's block
template <int T, int D>
int launch_fe(const StereoBank &b, long k_lo, long k_hi, hipStream_t s)
{
    using C = FeX<T, D>;
    const long ntiles = (k_hi - k_lo + C::TILE - 1) / C::TILE;
    const long wgs = ntiles;
    hipLaunchKernelGGL((chs_fe_exact_kernel<T, D>), dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(64), 0, s, b.slots.p,
                       static_cast<long>(b.slot_bytes), static_cast<int>(b.hist_bytes), k_lo, k_hi, ntiles, wgs, b.fe_table.p, b.demod.p,
                       b.dpitch, b.Hd);
    CHS_LAUNCH_CHECK("chs_fe_exact_kernel");
    return FMRX_OK;
}

template <int T>
int launch_bpf(const StereoBank &b, long k_lo, long k_hi, hipStream_t s)
{
    const long wgs = (k_hi - k_lo + 64 * kR - 1) / (64 * kR);
    if (b.exact)
        hipLaunchKernelGGL((chs_bpf_kernel<T, true>), dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(64), 0, s, b.demod.p, b.dpitch,
                           b.Hd, k_lo, k_hi, wgs, b.bpf_table.p, b.bpf.p, b.carrier.p, b.ypitch, nullptr, 0L);
    else
        hipLaunchKernelGGL((chs_bpf_kernel<T, false>), dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(64), 0, s, b.demod.p, b.dpitch,
                           b.Hd, k_lo, k_hi, wgs, b.bpf_table.p, b.bpf.p, nullptr, b.ypitch, b.carrier8.p, b.cpitch);
    CHS_LAUNCH_CHECK("chs_bpf_kernel");
    return FMRX_OK;
}

// audio outputs [a_lo, a_hi) of every channel's block, from the IF samples [.., g_hi)
template <int T, int D, bool STEREO>
int launch_out(StereoBank &b, float *d_audio, int16_t *d_pcm, int wrap, long a_lo, long a_hi, long g_hi, hipStream_t s)
{
    using C = OutX<T, D>;
    if ((reinterpret_cast<uintptr_t>(d_audio) % 16) || (reinterpret_cast<uintptr_t>(d_pcm) % 16) || (STEREO && b.n_audio % 4))
        return fail(FMRX_EINVAL, "channels: output buffers must be 16-byte aligned");
    const long wgs = (a_hi - a_lo + C::NOUT - 1) / C::NOUT;
    if (b.exact)
        hipLaunchKernelGGL((chs_out_kernel<T, D, STEREO, true>), dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(64), 0, s, b.demod.p,
                           b.dpitch, b.Hd, b.bpf.p, b.trig.p, b.ypitch, b.nco0.p, b.mixtail[b.mix_cur].p, b.mixtail[b.mix_cur ^ 1].p, b.Hm,
                           b.n_if, g_hi, b.delay, 2.0f, 0.0f, b.out_table.p, wgs, d_audio, d_pcm, wrap, a_lo, a_hi, b.n_audio);
    else if constexpr (STEREO)
        hipLaunchKernelGGL((chs_out_kernel<T, D, true, false>), dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(64), 0, s, b.demod.p,
                           b.dpitch, b.Hd, b.bpf.p, b.trig.p, b.ypitch, b.nco0.p, b.mixtail[b.mix_cur].p, b.mixtail[b.mix_cur ^ 1].p, b.Hm,
                           b.n_if, g_hi, b.delay, 2.0f, 0.0f, b.out_table.p, wgs, d_audio, d_pcm, wrap, a_lo, a_hi, b.n_audio);
    CHS_LAUNCH_CHECK("chs_out_kernel");
    return FMRX_OK;
}

int launch_nco(const StereoBank &b, long k_lo, long k_hi, hipStream_t s)
{
    const long wgs = (k_hi - k_lo + 1023) / 1024;
    if (b.exact)
        hipLaunchKernelGGL(chs_nco_kernel<true>, dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(256), 0, s, b.trig.p, b.ypitch, k_lo, k_hi,
                           wgs, 2.0f, 0.0f, b.bpf.p, b.nco0.p, b.resample ? b.mixer.p : nullptr, b.mpitch, b.Hm);
    else
        hipLaunchKernelGGL(chs_nco_kernel<false>, dim3(static_cast<unsigned>(wgs * b.n_channels)), dim3(256), 0, s, b.trig.p, b.ypitch, k_lo, k_hi,
                           wgs, 2.0f, 0.0f, b.bpf.p, b.nco0.p, b.resample ? b.mixer.p : nullptr, b.mpitch, b.Hm);
    CHS_LAUNCH_CHECK("chs_nco_kernel");
    return FMRX_OK;
}

// step-major taps of chs_resample_lanes_kernel: [group of the period][step][8] (7 outputs + pad), steps newest sample first,
// rounded up to whole iterations of 4 (+ one iteration of zeros that the kernel's look-ahead reads); top_of[group] = the window's
// newest sample relative to the period's first
int resample_lanes_table_init(StereoBank &b, const float *h)
{
    const int U = b.p.audio_upsamp, D = b.p.audio_decim, T = b.p.audio_taps;
    if (U <= 0 || U % kRS) return FMRX_OK;                         // other ratios: the one-thread-per-output kernel
    const int G = U / kRS;
    std::vector<int> top(G), n_of(U), ph_of(U);
    int wmax = 0;
    for (int k = 0; k < U; k++) {
        const long m = static_cast<long>(k) * D;
        ph_of[k] = static_cast<int>(m % U);
        n_of[k] = static_cast<int>((m - ph_of[k]) / U);
    }
    for (int g = 0; g < G; g++) {
        top[g] = n_of[g * kRS + kRS - 1];
        for (int r = 0; r < kRS; r++) {
            const int k = g * kRS + r, jmax = (T - 1 - ph_of[k]) / U;   // taps ph, ph + U, ..., ph + jmax U
            const int w = top[g] - (n_of[k] - jmax) + 1;
            if (w > wmax) wmax = w;
        }
    }
    const int iters = ((wmax + 3) / 4 + 7) / 8 * 8;                // whole iterations of 4 steps, in batches of 8 (32 samples)
    // the oldest sample the kernel touches lies 4 iters - 1 behind a group's top: the rows' histories cover it
    for (int g = 0; g < G; g++)
        if (4 * iters - 1 - top[g] > b.res_hist) b.res_hist = 4 * iters - 1 - top[g];
    std::vector<float> tab(static_cast<size_t>(G) * iters * 32 + 32, 0.0f);
    for (int g = 0; g < G; g++)
        for (int st = 0; st < 4 * iters; st++)
            for (int r = 0; r < kRS; r++) {
                const int k = g * kRS + r, j = n_of[k] - (top[g] - st);
                const long n = static_cast<long>(ph_of[k]) + static_cast<long>(j) * U;
                if (j >= 0 && n < T) tab[(static_cast<size_t>(g) * iters * 4 + st) * 8 + r] = h[n];
            }
    FMRX_TRY(b.res_table.alloc(tab.size()));
    FMRX_HIP(hipMemcpy(b.res_table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    FMRX_TRY(b.res_top.alloc(G));
    FMRX_HIP(hipMemcpy(b.res_top.p, top.data(), G * sizeof(int), hipMemcpyHostToDevice));
    b.res_groups = G;
    b.res_iters = iters;
    return FMRX_OK;
}

template <int T, int D>
int out_table_init(StereoBank &b, const float *h)
{
    using C = OutX<T, D>;
    std::vector<float> tab(static_cast<size_t>(C::NG) * 16, 0.0f);
    for (int u = 0; u < C::W; u++)
        for (int r = 0; r < kRO; r++) {
            const int n = r * D + (T - 1) - (C::W - 1 - u);
            if (n >= 0 && n < T) tab[static_cast<size_t>(u) * kRO + r] = h[n];
        }
    FMRX_TRY(b.out_table.alloc(tab.size()));
    FMRX_HIP(hipMemcpy(b.out_table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return FMRX_OK;
}

// convolveBlockResampleFIR for the audio outputs [a_lo, a_hi) of every channel (a_lo: a multiple of U): whole periods by the
// lane-per-channel kernel when its table exists, what is left (and every other ratio) by one thread per (channel, output)
template <bool STEREO>
int launch_resample(StereoBank &b, float *d_audio, int16_t *d_pcm, int wrap, long a_lo, long a_hi, hipStream_t s)
{
    const fmrx_params &p = b.p;
    long done = a_lo;
    if (b.res_groups > 0 && a_lo % p.audio_upsamp == 0) {
        const long periods = (a_hi - a_lo) / p.audio_upsamp;
        if (periods > 0) {
            const long cgs = (b.n_channels + 63) / 64;
            // a wave takes `gpw` consecutive groups of a period (neighbouring groups share most of their window): as many as leave
            // the chip >= 8 waves per SIMD
            int gpw = b.res_groups;
            constexpr int NW = kRW<STEREO>;
            while (gpw > NW && cgs * periods * ((b.res_groups + gpw - 1) / gpw) * NW < 8192) gpw = (gpw + 1) / 2;
            gpw = (gpw + NW - 1) / NW * NW;                        // every wave of a workgroup gets a group
            const long parts = (b.res_groups + gpw - 1) / gpw;
            auto go = [&](auto exactc) {
                hipLaunchKernelGGL((chs_resample_lanes_kernel<STEREO, decltype(exactc)::value>), dim3(static_cast<unsigned>(cgs * periods * parts)),
                                   dim3(64 * kRW<STEREO>), 0, s, b.demod.p, b.dpitch, b.Hd, STEREO ? b.mixer.p : nullptr, b.mpitch, b.Hm, b.delay, b.res_table.p,
                                   b.res_top.p, b.res_groups, gpw, b.res_iters, p.audio_decim, p.audio_upsamp, periods, b.n_channels, d_audio, d_pcm,
                                   wrap, a_lo, b.n_audio);
            };
            if (b.exact) go(std::true_type{});
            else go(std::false_type{});
            CHS_LAUNCH_CHECK("chs_resample_lanes_kernel");
            done = a_lo + periods * p.audio_upsamp;
        }
    }
    if (done < a_hi) {
        const long wr = (a_hi - done + 255) / 256;
        hipLaunchKernelGGL(chs_resample_exact_kernel<STEREO>, dim3(static_cast<unsigned>(wr * b.n_channels)), dim3(256), 0, s, b.demod.p, b.dpitch,
                           b.Hd, STEREO ? b.mixer.p : nullptr, b.mpitch, b.Hm, b.delay, b.h_res.p, p.audio_taps, p.audio_decim, p.audio_upsamp, wr,
                           d_audio, d_pcm, wrap, done, a_hi, b.n_audio);
        CHS_LAUNCH_CHECK("chs_resample_exact_kernel");
    }
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
    if (p.audio_upsamp > 0) au = p.audio_taps >= 2 && p.audio_taps <= 65535;   // resampling modes: the batched reference-order resampler takes any taps
    return fe && au && st;
}

void stereo_bank_destroy(StereoBank *b) { delete b; }

int stereo_bank_create(StereoBank **out, const fmrx_params &p, int n_channels, int audio_channels, int exact, size_t block_bytes)
{
    if (!exact && audio_channels != 2 && p.audio_upsamp == 0)
        return fail(FMRX_EINVAL, "channels: the fast mono bank of the integer-decimation modes is fmrx_channels_create's");
    if (!stereo_bank_supported(p, audio_channels))
        return fail(FMRX_EINVAL, "channels (exact): no reference-order kernels for rf %d/%d, audio %d/%d, stereo %d taps (modes 0 and 1 of the "
                    "reference's tap sets are covered)", p.rf_taps, p.rf_decim, p.audio_taps, p.audio_decim, p.stereo_taps);
    StereoBank *b = new StereoBank;
    b->p = p;
    b->n_channels = n_channels;
    b->audio_channels = audio_channels;
    b->exact = exact ? 1 : 0;
    b->opt = options_snapshot();
    b->block_bytes = block_bytes;
    auto body = [&]() -> int {
        std::vector<float> h(p.rf_taps);
        design_lpf(static_cast<float>(p.rf_Fs), 100000.0f, p.rf_taps, h.data());             // src/project.cpp:50
#define X(T_, D_) if (p.rf_taps == T_ && p.rf_decim == D_) FMRX_TRY((fe_table_init<T_, D_>(*b, h.data())));
        CHS_FE_CASES(X)
#undef X
        if (!b->exact) {   // the matrix-core front end (int8 MFMA on the raw bytes): its tap image, and the history its windows reach
            FMRX_TRY(fe_plan_init(b->fe, h.data(), p.rf_taps, p.rf_decim));
            const int lead = fe_mfma_bank_lead(b->fe);
            if (!b->fe.mfma || lead < 0) return fail(FMRX_EINVAL, "channels: no matrix-core front end for rf %d taps / decim %d", p.rf_taps, p.rf_decim);
            const size_t need = (static_cast<size_t>(lead) + 15) / 16 * 16;
            if (need > b->hist_bytes) b->hist_bytes = need;
        }
        b->resample = p.audio_upsamp > 0;
        std::vector<float> ha(p.audio_taps);
        // src/project.cpp:321-323: the resampling modes design the filter at the upsampled rate
        design_lpf(static_cast<float>(b->resample ? p.if_Fs * p.audio_upsamp : p.if_Fs), 16000.0f, p.audio_taps, ha.data());
        if (b->resample) {
            FMRX_TRY(b->h_res.alloc(p.audio_taps));
            FMRX_HIP(hipMemcpy(b->h_res.p, ha.data(), p.audio_taps * sizeof(float), hipMemcpyHostToDevice));
            FMRX_TRY(resample_lanes_table_init(*b, ha.data()));
        } else {
#define X(T_, D_) if (p.audio_taps == T_ && p.audio_decim == D_) FMRX_TRY((out_table_init<T_, D_>(*b, ha.data())));
            CHS_OUT_CASES(X)
#undef X
        }
        b->n = static_cast<long>(block_bytes / 2);
        b->n_if = b->n / p.rf_decim;
        b->n_audio = b->resample ? b->n_if * p.audio_upsamp / p.audio_decim : b->n_if / p.audio_decim;
        if (b->resample && (b->n_if * p.audio_upsamp) % p.audio_decim)
            return fail(FMRX_EINVAL, "channels: n_if * upsamp = %ld is not a multiple of audio_decim %d (a block must end on an output boundary)",
                        b->n_if * p.audio_upsamp, p.audio_decim);
        b->Ha = b->resample ? (p.audio_taps - 1) / p.audio_upsamp : p.audio_taps - 1;
        b->St = audio_channels == 2 ? p.stereo_taps : 0;
        b->delay = audio_channels == 2 ? (p.stereo_taps - 1) / 2 : 0;                        // allPass, src/filter.cpp:14-29
        b->Hd = b->Ha + b->delay;
        if (audio_channels == 2 && b->St - 1 + 3 > b->Hd) b->Hd = b->St - 1 + 3;
        const int more = b->res_hist > b->Ha ? b->res_hist - b->Ha : 0;   // (the lane-per-channel resampler rounds its windows up to whole iterations)
        if (b->Ha + b->delay + more > b->Hd) b->Hd = b->Ha + b->delay + more;
        b->Hd = (b->Hd + 3) / 4 * 4 + 4;
        b->Hm = (b->Ha + more + 3) / 4 * 4 + 4;
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
            if (!b->exact && !b->resample && fe_bpf_bank_available(b->fe, p.stereo_taps)) {
                FMRX_TRY(fe_bpf_tables_init(b->st_img, b->car_img, hs.data(), hc.data(), p.stereo_taps));
                b->fused_front = true;
            }
            if (b->exact) {
                FMRX_TRY(b->carrier.alloc(b->ypitch * N + 64));
                FMRX_HIP(hipMemset(b->carrier.p, 0, b->carrier.bytes()));
            } else {
                b->cpitch = (b->n_if + 64 + 15) / 16 * 16;
                FMRX_TRY(b->carrier8.alloc(b->cpitch * N + 64));
                FMRX_HIP(hipMemset(b->carrier8.p, 0, b->carrier8.bytes()));
            }
            FMRX_TRY(b->bpf.alloc(b->ypitch * N + 64));
            FMRX_TRY(b->trig.alloc(b->ypitch * N + 64));
            FMRX_TRY(b->pll.alloc(8 * N));
            FMRX_TRY(b->nco0.alloc(N));
            for (auto &m : b->mixtail) {
                FMRX_TRY(m.alloc(static_cast<size_t>(b->Hm) * N));
                FMRX_HIP(hipMemset(m.p, 0, m.bytes()));
            }
            if (b->resample) {
                b->mpitch = (b->Hm + b->n_if + 16 + 3) / 4 * 4;
                FMRX_TRY(b->mixer.alloc(b->mpitch * N + 64));
                FMRX_HIP(hipMemset(b->mixer.p, 0, b->mixer.bytes()));
            }
            hipLaunchKernelGGL(chs_fill_state_kernel, dim3(static_cast<unsigned>((8 * N + 255) / 256)), dim3(256), 0, nullptr, b->pll.p,
                               static_cast<long>(8 * N));
            CHS_LAUNCH_CHECK("chs_fill_state_kernel");
            FMRX_HIP(hipStreamCreateWithFlags(&b->wide, hipStreamNonBlocking));
            FMRX_HIP(hipStreamCreateWithFlags(&b->lanes, hipStreamNonBlocking));
            FMRX_HIP(hipStreamCreateWithFlags(&b->front, hipStreamNonBlocking));
            FMRX_HIP(hipStreamCreateWithFlags(&b->post, hipStreamNonBlocking));
            for (auto &e : b->ev_fe) FMRX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto &e : b->ev_bpf) FMRX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto &e : b->ev_pll) FMRX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            FMRX_HIP(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
            FMRX_HIP(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
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
            if (b->resample) FMRX_HIP(hipMemsetAsync(b->mixer.p + c * b->mpitch, 0, b->Hm * sizeof(float), nullptr));
            hipLaunchKernelGGL(chs_fill_state_kernel, dim3(1), dim3(8), 0, nullptr, b->pll.p + 8 * c, 8L);
        }
    }
    if (channel < 0) {
        FMRX_TRY(k_fill_u8(b->slots.p, b->slot_bytes * b->n_channels, 128, nullptr));
        FMRX_HIP(hipMemsetAsync(b->demod.p, 0, b->demod.bytes(), nullptr));
        if (b->audio_channels == 2) {
            for (auto &m : b->mixtail) FMRX_HIP(hipMemsetAsync(m.p, 0, m.bytes(), nullptr));
            if (b->resample) FMRX_HIP(hipMemsetAsync(b->mixer.p, 0, b->mixer.bytes(), nullptr));
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
    auto fe = [&](long k_lo, long k_hi, hipStream_t st) -> int {
        if (!b->exact)
            return fe_mfma_bank_launch(b->fe, b->slots.p, static_cast<long>(b->slots.n), static_cast<long>(b->slot_bytes),
                                       static_cast<long>(b->hist_bytes), b->n_channels, k_lo, k_hi, b->demod.p, b->dpitch, b->Hd,
                                       b->opt.bank_fe_wgs, st);
#define X(T_, D_) if (p.rf_taps == T_ && p.rf_decim == D_) return launch_fe<T_, D_>(*b, k_lo, k_hi, st);
        CHS_FE_CASES(X)
#undef X
        return FMRX_EINVAL;
    };
    if (b->audio_channels == 2) {
        auto bpf = [&](long k_lo, long k_hi, hipStream_t st) -> int {
#define X(T_) if (p.stereo_taps == T_) return launch_bpf<T_>(*b, k_lo, k_hi, st);
            CHS_BPF_CASES(X)
#undef X
            return FMRX_EINVAL;
        };
        // IF samples [.., if_of(a)) are what the audio outputs [.., a) read: a * D in the integer-decimation modes, whole periods
        // (U outputs <-> D samples) in the resampling modes
        auto if_of = [&](long a) -> long { return b->resample ? a / p.audio_upsamp * p.audio_decim : a * p.audio_decim; };
        auto out = [&](long a_lo, long a_hi, long g_hi, hipStream_t st) -> int {
            if (b->resample) {
                (void)g_hi;   // the mixer rows of this chunk were written by its NCO pass (launch_nco)
                return launch_resample<true>(*b, d_audio, d_pcm, wrap, a_lo, a_hi, st);
            }
#define X(T_, D_) if (p.audio_taps == T_ && p.audio_decim == D_) return launch_out<T_, D_, true>(*b, d_audio, d_pcm, wrap, a_lo, a_hi, g_hi, st);
            CHS_OUT_CASES(X)
#undef X
            return FMRX_EINVAL;
        };
        // chunks of whole output workgroups (512 audio samples; whole periods of U outputs in the resampling modes);
        // chunk c = audio [a_c, a_c+1) = IF [if_of(a_c), if_of(a_c+1))
        const long unit = b->resample ? p.audio_upsamp : 512;
        long per = (b->n_audio + b->max_chunks - 1) / b->max_chunks;
        per = (per + unit - 1) / unit * unit;
        const int K = static_cast<int>((b->n_audio + per - 1) / per);
        hipStream_t sw = K > 1 ? b->wide : s, sl = K > 1 ? b->lanes : s;
        // fast banks: the front end is HBM-bound on the matrix cores, the band-pass pair and the output stage are bound by the
        // vector ALUs: on two streams they run side by side (option bank_streams = 2: everything but the PLL on one stream)
        const bool split = K > 1 && !b->exact && b->opt.bank_streams >= 3;
        hipStream_t sf = split ? b->front : sw;
        // bank_streams = 4: the output stage (bound by the latency of its staging, not by the vector ALUs) on a fourth stream, next to
        // the band-pass pair instead of behind it
        const bool split_out = split && b->opt.bank_streams >= 4;
        hipStream_t so = split_out ? b->post : sw;
        if (K > 1) {   // whatever the caller's stream did before the call (loading the slots, reading the last output) comes first
            FMRX_HIP(hipEventRecord(b->ev_fork, s));
            FMRX_HIP(hipStreamWaitEvent(sw, b->ev_fork, 0));
            FMRX_HIP(hipStreamWaitEvent(sl, b->ev_fork, 0));
            if (split) FMRX_HIP(hipStreamWaitEvent(sf, b->ev_fork, 0));
            if (split_out) FMRX_HIP(hipStreamWaitEvent(so, b->ev_fork, 0));
        }
        // The output stage of chunk c follows the band-pass pair of chunk c + lag on the wide stream.  Exact banks: lag 1 (the PLL is
        // the longest stage; nothing on the wide stream is waited for).  Fast banks: lag 2 -- the PLL of chunk c starts when its
        // band-pass pair ends and takes longer than the next chunk's band-pass pair: with lag 1 the wide stream idled a third
        // of the time waiting for it.
        const int lag = (b->exact || K < 3) ? 1 : 2;
        // the front end works in whole tiles of its own (63 x 8 outputs per wave; 120 per matrix-core tile): its share of a chunk
        // ends on the first tile boundary at or behind the chunk's end, so that no chunk pays for a partly filled last tile
        // (2560-sample chunks are 5.08 tiles of 504: 15 % of the exact front end's work was computed and thrown away)
        const long fe_tile = b->exact ? 63 * kR : 120;
        long fe_done = 0;
        for (int c = 0; c < K + lag; c++) {
            if (c < K) {
                const long a_lo = c * per, a_hi = a_lo + per < b->n_audio ? a_lo + per : b->n_audio;
                const long k_lo = if_of(a_lo), k_hi = if_of(a_hi);
                const bool fused = b->fused_front && b->opt.bank_fused != 0;
                if (fused) {
                    // fast banks: front end + band-pass pair in ONE kernel (int8 + f32 matrix cores), on the front stream
                    FMRX_TRY(fe_bpf_bank_launch(b->fe, p.stereo_taps, b->st_img.p, b->car_img.p, b->slots.p, static_cast<long>(b->slots.n),
                                                static_cast<long>(b->slot_bytes), static_cast<long>(b->hist_bytes), b->n_channels, k_lo, k_hi,
                                                b->demod.p, b->dpitch, b->Hd, b->bpf.p, b->ypitch, b->carrier8.p, b->cpitch,
                                                b->opt.bank_fe_wgs_fused, sf));
                    if (K > 1) {   // read-after-write: the PLL's lanes read the sign bytes this kernel wrote
                        FMRX_HIP(hipEventRecord(b->ev_bpf[c], sf));
                        FMRX_HIP(hipStreamWaitEvent(sl, b->ev_bpf[c], 0));
                    }
                } else {
                    long fe_hi = (k_hi + fe_tile - 1) / fe_tile * fe_tile;
                    if (fe_hi > b->n_if || c == K - 1) fe_hi = b->n_if;
                    if (fe_hi > fe_done) FMRX_TRY(fe(fe_done, fe_hi, sf));
                    fe_done = fe_hi;
                    if (split) {   // read-after-write: the band-pass pair reads the discriminator rows the front end wrote on its own stream
                        FMRX_HIP(hipEventRecord(b->ev_fe[c], sf));
                        FMRX_HIP(hipStreamWaitEvent(sw, b->ev_fe[c], 0));
                    }
                    FMRX_TRY(bpf(k_lo, k_hi, sw));
                    if (K > 1) {
                        FMRX_HIP(hipEventRecord(b->ev_bpf[c], sw));
                        FMRX_HIP(hipStreamWaitEvent(sl, b->ev_bpf[c], 0));
                    }
                }
                // fmPLL(carrier_filt, 19 kHz, if_Fs, ncoScale 2, phaseAdjust 0, normBandwidth 0.01): src/project.cpp:237
                if (b->exact)
                    FMRX_TRY(k_fm_pll_channels(b->carrier.p + k_lo, b->ypitch, static_cast<size_t>(k_hi - k_lo), b->n_channels,
                                               b->trig.p + k_lo, b->ypitch, b->pll.p, c == 0 ? b->nco0.p : nullptr, 19e3f,
                                               static_cast<float>(p.if_Fs), 2.0f, 0.0f, 0.01f, sl, true, true));
                else
                    FMRX_TRY(k_fm_pll_channels(reinterpret_cast<const float *>(b->carrier8.p + k_lo), b->cpitch,
                                               static_cast<size_t>(k_hi - k_lo), b->n_channels, b->trig.p + k_lo, b->ypitch, b->pll.p,
                                               c == 0 ? b->nco0.p : nullptr, 19e3f, static_cast<float>(p.if_Fs), 2.0f, 0.0f, 0.01f, sl, true,
                                               false, true));
                if (K > 1) FMRX_HIP(hipEventRecord(b->ev_pll[c], sl));
            }
            if (c >= lag) {   // the output stage of an earlier chunk, behind this chunk's band-pass pair on the wide stream
                const long a_lo = (c - lag) * per, a_hi = a_lo + per < b->n_audio ? a_lo + per : b->n_audio;
                if (K > 1) FMRX_HIP(hipStreamWaitEvent(so, b->ev_pll[c - lag], 0));   // (the PLL followed this chunk's band-pass pair: both are done)
                // fast banks, modes 0/1: inside the output stage; the resampling modes materialise the mixer rows from finished NCO values
                if (b->exact || b->resample) FMRX_TRY(launch_nco(*b, if_of(a_lo), if_of(a_hi), so));
                FMRX_TRY(out(a_lo, a_hi, if_of(a_hi), so));
            }
        }
        if (K > 1) {
            FMRX_HIP(hipEventRecord(b->ev_join, so));               // the last output stage follows everything else of the call
            FMRX_HIP(hipStreamWaitEvent(s, b->ev_join, 0));
        }
        b->mix_cur ^= 1;
    } else {
        FMRX_TRY(fe(0, b->n_if, s));
        if (b->resample) {
            FMRX_TRY(launch_resample<false>(*b, d_audio, d_pcm, wrap, 0, b->n_audio, s));
        } else {
#define X(T_, D_) if (p.audio_taps == T_ && p.audio_decim == D_) FMRX_TRY((launch_out<T_, D_, false>(*b, d_audio, d_pcm, wrap, 0, b->n_audio, b->n_if, s)));
            CHS_OUT_CASES(X)
#undef X
        }
    }
    hipLaunchKernelGGL(chs_finish_kernel, dim3(static_cast<unsigned>(b->n_channels)), dim3(64), 0, s, b->slots.p,
                       static_cast<long>(b->slot_bytes), static_cast<long>(b->hist_bytes), b->demod.p, b->dpitch, b->Hd, b->n_if,
                       b->resample && b->audio_channels == 2 ? b->mixer.p : nullptr, b->mpitch, b->Hm);
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
    case FMRX_TAP_CARRIER: if (b->audio_channels == 2 && b->exact) src = b->carrier.p + channel * b->ypitch; break;
    case FMRX_TAP_STEREO_BPF: if (b->audio_channels == 2) src = b->bpf.p + channel * b->ypitch; break;
    case FMRX_TAP_PLL: if (b->audio_channels == 2) { src = b->trig.p + channel * b->ypitch; cnt = n_if + 1; } break;
    default: break;
    }
    if (!src) return fail(FMRX_EINVAL, "channels_read_tap: tap %d is not kept by this bank", which);
    *n = cnt;
    if (!out) return FMRX_OK;
    if (which == FMRX_TAP_PLL) {
        FMRX_HIP(hipMemcpy(out, b->nco0.p + channel, sizeof(float), hipMemcpyDeviceToHost));
        if (!b->exact && !b->resample) {
            // the fast bank of modes 0/1 keeps the raw trigArg of every step (the cosine is taken inside the output stage): the same
            // NCO pass the other banks run, here on a copy of the one row
            DevBuf<float> tmp;
            const long pitch = (static_cast<long>(n_if) + 16 + 3) / 4 * 4;
            FMRX_TRY(tmp.alloc(static_cast<size_t>(pitch)));
            FMRX_HIP(hipMemcpy(tmp.p, src, n_if * sizeof(float), hipMemcpyDeviceToDevice));
            const long wgs = (static_cast<long>(n_if) + 1023) / 1024;
            hipLaunchKernelGGL(chs_nco_kernel<false>, dim3(static_cast<unsigned>(wgs)), dim3(256), 0, nullptr, tmp.p, pitch, 0L, static_cast<long>(n_if),
                               wgs, 2.0f, 0.0f, nullptr, nullptr, nullptr, 0L, 0);
            CHS_LAUNCH_CHECK("chs_nco_kernel");
            FMRX_HIP(hipMemcpy(out + 1, tmp.p, n_if * sizeof(float), hipMemcpyDeviceToHost));
            return FMRX_OK;
        }
        FMRX_HIP(hipMemcpy(out + 1, src, n_if * sizeof(float), hipMemcpyDeviceToHost));
        return FMRX_OK;
    }
    FMRX_HIP(hipMemcpy(out, src, cnt * sizeof(float), hipMemcpyDeviceToHost));
    return FMRX_OK;
}

}  // namespace fmrx

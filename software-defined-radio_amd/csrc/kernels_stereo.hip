// kernels_stereo.hip -- the two band-pass filters of the stereo path in one pass.
//
// Replaces the two convolveBlockFIR calls of RF_STEREO (src/project.cpp:202, 207 ->
// src/filter.cpp:133-154): stereo_filt = demod * h(22-54 kHz), carrier_filt =
// demod * h(18.5-19.5 kHz), no decimation, same input.  The two filters ride in the
// two halves of v_pk_fma_f32: acc(st, car) += x * (h_st[n], h_car[n]) with the tap
// PAIR as a wave-uniform SGPR operand and x broadcast to both halves.  A thread
// produces 8 consecutive outputs from a register-resident window of 8+T-1 samples
// read straight from global memory (27 x 16 B at T = 101; lanes are 32 B apart, so a
// wave's loads walk 2.4 KB of contiguous, L1-resident data) -- no LDS, no barrier.
// ~101 packed FMAs per IF sample for both filters.
//
// Numerics: one FMA per tap, taps ascending in n as the reference does; differs
// from its separate multiply/add by float32 rounding only (generic kernel = exact).
#include "device_math.hpp"
#include "fmrx_internal.hpp"

namespace fmrx {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kPC = 8;   // tap pairs per scalar load (16 SGPRs)

#define FMRX_PAIRS_ISSUE(hp, table, off) asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(hp) : "s"(table), "i"(off))
#define FMRX_PAIRS_WAIT(hp) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(hp))

template <int T, int R, int G>
__device__ __forceinline__ void bpf_step(const float (&w)[R + T - 1 + 3], const float *__restrict__ table, f2 (&acc)[R],
                                         f16v &hA, f16v &hB)
{
    constexpr int NG = (T + kPC - 1) / kPC;
    if constexpr (G < NG) {
        // table entry m = (h_st[T-1-m], h_car[T-1-m]): window sample i = r + m meets output r
        float hq[2 * kPC];
        if constexpr (G % 2 == 0) {
            FMRX_PAIRS_WAIT(hA);
            if constexpr (G + 1 < NG) FMRX_PAIRS_ISSUE(hB, table, (G + 1) * kPC * 8);
#pragma unroll
            for (int k = 0; k < 2 * kPC; k++) hq[k] = hA[k];
        } else {
            FMRX_PAIRS_WAIT(hB);
            if constexpr (G + 1 < NG) FMRX_PAIRS_ISSUE(hA, table, (G + 1) * kPC * 8);
#pragma unroll
            for (int k = 0; k < 2 * kPC; k++) hq[k] = hB[k];
        }
#pragma unroll
        for (int s = 0; s < R + kPC - 1; s++) {
            const int i = G * kPC + s;
            if (i < R + T - 1) {
                const float x = w[i];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int m = i - r;
                    if (m >= G * kPC && m < (G + 1) * kPC && m < T)
                        acc[r] = __builtin_elementwise_fma((f2){x, x}, (f2){hq[2 * (m - G * kPC)], hq[2 * (m - G * kPC) + 1]}, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" : "+v"(acc[r]));
        bpf_step<T, R, G + 1>(w, table, acc, hA, hB);
    }
}

template <int T, int R, int NT>
__global__ __launch_bounds__(NT) void bpf_pair_kernel(const float *__restrict__ x, long n, const float *__restrict__ table,
                                                       float *__restrict__ y_st, float *__restrict__ y_car)
{
    // x is 16-byte aligned at index 0 and has >= T-1+3 readable samples of history in front
    const long k0 = (static_cast<long>(blockIdx.x) * NT + threadIdx.x) * R;   // first output of this thread
    if (k0 >= n) return;
    constexpr int LEAD = (4 - (T - 1) % 4) % 4;          // so that the window starts on a 16-byte boundary
    constexpr int NW = R + T - 1 + LEAD;                 // floats loaded
    static_assert(NW % 4 == 0 && LEAD <= 3, "window must be whole 16-byte chunks");
    const f4 *src = reinterpret_cast<const f4 *>(x + k0 - (T - 1) - LEAD);
    float wl[NW];
#pragma unroll
    for (int i = 0; i < NW / 4; i++) {
        const f4 v = src[i];
        wl[4 * i] = v.x;
        wl[4 * i + 1] = v.y;
        wl[4 * i + 2] = v.z;
        wl[4 * i + 3] = v.w;
    }
    float w[R + T - 1 + 3];
#pragma unroll
    for (int i = 0; i < R + T - 1; i++) w[i] = wl[i + LEAD];
    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (f2){0.0f, 0.0f};
    f16v hA, hB;
    FMRX_PAIRS_ISSUE(hA, table, 0);
    bpf_step<T, R, 0>(w, table, acc, hA, hB);
    if (k0 + R <= n) {
        f4 *ds = reinterpret_cast<f4 *>(y_st + k0), *dc = reinterpret_cast<f4 *>(y_car + k0);
#pragma unroll
        for (int r = 0; r < R; r += 4) {
            ds[r / 4] = (f4){acc[r].x, acc[r + 1].x, acc[r + 2].x, acc[r + 3].x};
            dc[r / 4] = (f4){acc[r].y, acc[r + 1].y, acc[r + 2].y, acc[r + 3].y};
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; r++)
            if (k0 + r < n) {
                y_st[k0 + r] = acc[r].x;
                y_car[k0 + r] = acc[r].y;
            }
    }
}

template <int T>
int launch(const BpfPairPlan &pl, const float *d_x, size_t n, float *d_st, float *d_car, hipStream_t stream)
{
    constexpr int R = 8, NT = 256;
    const unsigned grid = static_cast<unsigned>((n + NT * R - 1) / (NT * R));
    hipLaunchKernelGGL((bpf_pair_kernel<T, R, NT>), dim3(grid), dim3(NT), 0, stream, d_x, static_cast<long>(n), pl.table.p,
                       d_st, d_car);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch bpf_pair_kernel<%d>: %s", T, hipGetErrorString(e));
    return FMRX_OK;
}

// ---- everything behind the PLL in one kernel (modes 0/1) --------------------------------------------
// Replaces, fused: the mixer loop (src/project.cpp:246-248), the two audio convolveBlockFastFIR calls of
// RF_STEREO -- mono branch on the all-passed discriminator output (:194, :219) and stereo branch on the
// mixer output (:257) -- the L/R combine (:277-280) and the interleaved PCM writer (:292-302).
// The two audio FIRs use the same taps on two streams, so they ride in the two halves of v_pk_fma_f32:
//   acc(mono, st) += h[n] * (demod[D k - n - delay], mixer[D k - n]),  mixer[i] = stereo_filt[i] * nco[i] * 2
// (the all-pass is the index offset `delay`).  A workgroup stages the window of NT*R audio outputs as (mono,
// mixer) pairs in LDS -- the mixer products are formed while staging and never go to HBM -- and lane t
// produces outputs t, t+NT, ...: neighbouring lanes are D pairs apart in LDS (conflict-free for D = 5, 2-way
// for D = 6) and store neighbouring outputs.  Taps are wave-uniform scalar loads.  Samples before the block:
// the discriminator history sits in front of its buffer (or at the tail of the previous block's), the mixer
// history (state_stereofilt) is the tail the previous call left in mix_tail_in; this call leaves its own in
// mix_tail_out (every workgroup that stages one of the block's last `hm` samples writes it: same values).
template <int T, int D, int R, int NT>
__global__ __launch_bounds__(NT) void stereo_out_kernel(const float *__restrict__ demod, const float *__restrict__ bpf,
                                                         const float *__restrict__ nco, const float *__restrict__ mix_tail_in,
                                                         float *__restrict__ mix_tail_out, int hm, long n_if, int delay,
                                                         const float *__restrict__ h, float *__restrict__ mono_out,
                                                         float *__restrict__ st_out, float *__restrict__ left, float *__restrict__ right,
                                                         int16_t *__restrict__ pcm, int wrap, float *__restrict__ mixer_out, long n_out)
{
    constexpr int NOUT = NT * R;
    constexpr int WL = D * (NOUT - 1) + T;
    extern __shared__ f2 win[];
    const int t = threadIdx.x;
    const long a0 = static_cast<long>(blockIdx.x) * NOUT;
    const long g0 = D * a0 - (T - 1);                      // IF index of window sample 0
    // D-1 samples past the window are visited too: when the block ends exactly on a tile boundary nobody's window
    // reaches the block's last D-1 samples, and they belong to the tail this call leaves behind
    // every load of the staging in flight before the first LDS write (a loop of load -> branch -> store pays one HBM / L2 round
    // trip per iteration: 11 of them were most of this kernel's time)
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
            vm[q] = demod[g - delay];                          // history in front of the buffer: negative indices are valid
            const float *pa = g >= 0 ? bpf + g : mix_tail_in + (hm + g);
            va[q] = *pa;
            if (g >= 0) vb[q] = nco[g];
        }
    }
#pragma unroll
    for (int q = 0; q < NJ; q++) asm volatile("" : "+v"(vm[q]), "+v"(va[q]), "+v"(vb[q]));
#pragma unroll
    for (int q = 0; q < NJ; q++) {
        const int j = t + q * NT;
        const long g = g0 + j;
        if (j >= WL + D - 1) continue;
        float m = 0.0f, x = 0.0f;
        if (g < n_if) {
            m = vm[q];
            if (g >= 0) {
                x = (va[q] * vb[q]) * 2.0f;                    // the reference's order: (stereo_filt * PLL) * 2
                if (mixer_out) mixer_out[g] = x;
                if (g >= n_if - hm) mix_tail_out[g - (n_if - hm)] = x;
            } else {
                x = va[q];
                if (n_if < hm && hm + g >= n_if) mix_tail_out[hm + g - n_if] = x;   // a block shorter than the history keeps what it must
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
        for (int r = 0; r < R; r++) acc[r] = __builtin_elementwise_fma(w0[D * NT * r - n], (f2){hn, hn}, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const long k = a0 + static_cast<long>(r) * NT + t;
        if (k < n_out) {
            const float mo = acc[r].x, st = acc[r].y;
            const float l = st + mo, rr = mo - st;         // src/project.cpp:278-279
            if (mono_out) mono_out[k] = mo;
            if (st_out) st_out[k] = st;
            if (left) left[k] = l;
            if (right) right[k] = rr;
            if (pcm) {
                using s2 = short __attribute__((ext_vector_type(2)));
                *reinterpret_cast<s2 *>(pcm + 2 * k) = (s2){pcm_pack_flat(l, wrap), pcm_pack_flat(rr, wrap)};
            }
        }
    }
}

// stereo tap counts in the reference: 13 (project.cpp:429), 151 (model), 101 (the report's final choice)
#define FMRX_BPF_CASES(X) X(13) X(101) X(151)

}  // namespace

int bpf_pair_plan_init(BpfPairPlan &pl, const float *h_stereo, const float *h_carrier, int taps)
{
    pl.taps = taps;
    pl.fast = false;
    FMRX_TRY(pl.h_st.alloc(taps));
    FMRX_TRY(pl.h_car.alloc(taps));
    FMRX_HIP(hipMemcpy(pl.h_st.p, h_stereo, taps * sizeof(float), hipMemcpyHostToDevice));
    FMRX_HIP(hipMemcpy(pl.h_car.p, h_carrier, taps * sizeof(float), hipMemcpyHostToDevice));
#define X(T_) \
    if (taps == T_) pl.fast = true;
    FMRX_BPF_CASES(X)
#undef X
    if (pl.fast) {
        const int ng = (taps + kPC - 1) / kPC;
        std::vector<float> tab(static_cast<size_t>(ng) * kPC * 2, 0.0f);
        for (int m = 0; m < taps; m++) {
            tab[2 * m] = h_stereo[taps - 1 - m];
            tab[2 * m + 1] = h_carrier[taps - 1 - m];
        }
        FMRX_TRY(pl.table.alloc(tab.size()));
        FMRX_HIP(hipMemcpy(pl.table.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return FMRX_OK;
}

// d_x: block start (16-byte aligned), taps-1+3 samples of history in front
int bpf_pair_launch(const BpfPairPlan &pl, const float *d_x, size_t n, float *d_st, float *d_car, hipStream_t stream,
                    bool force_generic)
{
    if (n == 0) return FMRX_OK;
    if (pl.fast && !force_generic && reinterpret_cast<uintptr_t>(d_x) % 16 == 0 && reinterpret_cast<uintptr_t>(d_st) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(d_car) % 16 == 0) {
#define X(T_) \
    if (pl.taps == T_) return launch<T_>(pl, d_x, n, d_st, d_car, stream);
        FMRX_BPF_CASES(X)
#undef X
    }
    FMRX_TRY(k_fir_generic(d_x, n, pl.h_st.p, pl.taps, 1, d_st, stream));
    return k_fir_generic(d_x, n, pl.h_car.p, pl.taps, 1, d_car, stream);
}

// d_demod: block start, with >= taps-1+delay readable samples in front; d_nco[i] = PLL[i] (i < n_if);
// d_mix_tail_in / _out: hm floats each (hm >= taps-1; index hm+g holds mixer sample g < 0)
bool stereo_out_available(int taps, int decim)
{
    return (taps == 101 || taps == 13) && (decim == 5 || decim == 6);
}

int stereo_out_launch(const float *d_demod, const float *d_bpf, const float *d_nco, const float *d_mix_tail_in,
                      float *d_mix_tail_out, int hm, size_t n_if, int delay, const float *d_h, int taps, int decim, float *d_mono,
                      float *d_st, float *d_left, float *d_right, int16_t *d_pcm, int wrap, float *d_mixer, hipStream_t stream)
{
    const long n_out = static_cast<long>(n_if / decim);
    if (n_out == 0) return FMRX_OK;
    if (hm < taps - 1) return fail(FMRX_EINVAL, "stereo_out: mixer history shorter than taps-1");
    constexpr int R = 2, NT = 256;
    const unsigned grid = static_cast<unsigned>((n_out + NT * R - 1) / (NT * R));
#define X(T_, D_)                                                                                                       \
    if (taps == T_ && decim == D_) {                                                                                    \
        constexpr size_t lds = (static_cast<size_t>(D_) * (NT * R - 1) + T_) * sizeof(f2);                              \
        hipLaunchKernelGGL((stereo_out_kernel<T_, D_, R, NT>), dim3(grid), dim3(NT), lds, stream, d_demod, d_bpf, d_nco, \
                           d_mix_tail_in, d_mix_tail_out, hm, static_cast<long>(n_if), delay, d_h, d_mono, d_st, d_left,  \
                           d_right, d_pcm, wrap, d_mixer, n_out);                                                        \
        hipError_t e = hipGetLastError();                                                                               \
        if (e != hipSuccess) return fail(FMRX_EHIP, "launch stereo_out_kernel<%d,%d>: %s", T_, D_, hipGetErrorString(e)); \
        return FMRX_OK;                                                                                                 \
    }
    X(101, 5) X(101, 6) X(13, 5) X(13, 6)
#undef X
    return fail(FMRX_EINVAL, "stereo_out_launch: no kernel for taps=%d decim=%d", taps, decim);
}

}  // namespace fmrx

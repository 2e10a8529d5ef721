// kernels_pll.hip -- the stereo pilot PLL / NCO.
//
// Replaces fmPLL (src/filter.cpp:32-80; called at src/project.cpp:237): per IF
// sample k
//     eD = atan2f(in*(-fbQ), in*fbI);  integ += Ki*eD;  phase += Kp*eD + integ;
//     trigOff += 1;  trigArg = (float)(2*PI*(freq/Fs)*trigOff + phase);
//     fbI = cosf(trigArg);  fbQ = sinf(trigArg);  out[k+1] = cosf(trigArg*ncoScale + phaseAdjust)
// with state {integ, phase, fbI, fbQ, lastOut, trigOff} carried between blocks.
// It is a serial float32 recurrence (the reference's most expensive stereo block,
// report Table 4) and it is what SURVEY 7.3 calls the hard part on a GPU.
//
// What "the reference's result" is.  trigArg grows by ~0.5 rad per IF sample and is
// rounded to float32 every step, so the NCO phase lives on a grid of ulp(trigArg)
// (1e-3 rad after 0.1 s, 8e-3 rad after 1 s of stream: SURVEY Q9), and the loop
// feeds that grid back.  The recurrence is therefore chaotic at the grid level: ANY
// one-ulp difference in an input sample or in one sinf/cosf/atan2f result eventually
// moves one rounding of trigArg, and from then on the two trajectories differ by
// independent +-1-grid-step flips at a few per cent of the samples (audio error
// ~0.2 ulp(trigArg): 1e-4 RMS at 0.15 s, 7e-4 at 1-4 s; measured for the oracle
// against itself with one input sample moved by one ulp).
// Parity with the reference over a long stream is thus all or nothing, and two forms
// are built:
//
// EXACT (MATH = kExact).  sinf / cosf / atan2f are glibc 2.35's own algorithms
//    (glibc_libm.hpp: pinned against the C library over all 2^32 arguments), the
//    float32 operations are the reference's in its order, and the recurrence is walked
//    serially: given bit-identical input the output is bit-identical to the reference's,
//    for any stream length.  The stage API (fmrx_fm_pll) and the bit-exact pipeline
//    (set_force_generic: every upstream kernel in the reference's evaluation order too)
//    use it.  The chain's own work per step is atan2f + sincosf; the NCO output's
//    cosf(trigArg*ncoScale + phaseAdjust) is not on the chain and is evaluated by a
//    second, parallel kernel from the stored trigArg.
//
// FAST (MATH = kFast), the throughput form of the specialised pipeline (whose upstream
//    kernels already differ from the reference by summation order, i.e. by ulps):
//  1. Cheaper steps: one argument reduction in double shared by the three
//     trigonometric values, hardware v_sin_f32 / v_cos_f32, and the phase detector in
//     closed form (atan2f(v*-sin t, v*cos t) = -t, turned by pi for v < 0).
//  2. Parallel in time.  A locked loop forgets its past (contraction |1 - Kp| per
//     step, damping 0.707) down to the trigArg grid.  The block is cut into segments
//     of L samples, one lane each; a lane starts W samples early from the block's
//     initial state (phase extrapolated with the slope observed over the previous
//     block) and runs the same recurrence; after the warm-up it agrees with the serial
//     trajectory TO WITHIN THE GRID, not bit for bit.  A second kernel compares, for
//     every segment, the state a lane had at its segment start with the state its
//     predecessor ended on, against a tolerance (kPllTolPhase + 2 ulp(trigArg) on the
//     phase, kPllTolInteg + 2 Ki ulp(trigArg) on the integrator); a third walks the
//     recurrence serially from every segment that is outside it (loop not locked:
//     stream start, drop-outs, phase jumps) until it is inside again.
//    What FAST guarantees is therefore the error floor described above, not identity:
//    tests/test_gpu_parity.py::test_stereo_error_envelope_long_stream states and checks
//    the envelope against the oracle.
#include "fmrx_internal.hpp"
#include "glibc_libm.hpp"

#pragma clang fp contract(off)

namespace fmrx {

namespace {

struct PllCoef {
    float Kp, Ki, ncoScale, phaseAdjust;
    double w;   // 2*PI*(freq/Fs), freq/Fs a float division as in the reference
};

struct PllState {
    float integ, phase, fbI, fbQ, last, off;
    float fr;   // FAST only: trigArg / 2 pi of the step that produced fbI/fbQ, reduced to [-0.5, 0.5] revolutions
};

enum PllMath { kExact = 0, kFast = 1 };

// One step.  kExact: the reference's operations in its order, glibc's functions; s.last holds the
// raw trigArg (the NCO output cosf(trigArg*ncoScale + phaseAdjust) is not part of the recurrence:
// nco_out_kernel applies it afterwards).  kFast: see the file header; s.last is the NCO output.
template <int MATH>
__device__ __forceinline__ void pll_step(PllState &s, float v, const PllCoef &c)
{
    float eD;
    if (MATH == kFast && fabsf(v) > 1e-20f && fabsf(v) < 1e20f) {
        // atan2f(v * -sin t, v * cos t) is -t for v > 0 and -t turned by pi for v < 0: no arctangent
        // needed.  Zero / non-finite / denormal-product samples take the library path below: there the
        // reference's result hangs on signed zeros and infinities.
        const float er = v > 0.0f ? -s.fr : (s.fr >= 0.0f ? 0.5f - s.fr : -0.5f - s.fr);
        eD = er * 6.28318530717958647692f;
    } else {
        const float eI = v * s.fbI;
        const float eQ = v * (-1 * s.fbQ);
        eD = glibc235::atan2f_glibc(eQ, eI);
    }
    s.integ = s.integ + c.Ki * eD;
    const float pe = c.Kp * eD;
    s.phase = (s.phase + pe) + s.integ;
    s.off += 1;
    const float trigArg = static_cast<float>(c.w * static_cast<double>(s.off) + static_cast<double>(s.phase));
    if (MATH == kFast) {
        const float sc = trigArg * c.ncoScale;
        const double inv2pi = 0.15915494309189533577;
        const double rev = static_cast<double>(trigArg) * inv2pi;
        const float fr = static_cast<float>(rev - rint(rev));          // [-0.5, 0.5] revolutions
        s.fr = fr;
        s.fbI = __builtin_amdgcn_cosf(fr);
        s.fbQ = __builtin_amdgcn_sinf(fr);
        const double rev2 = static_cast<double>(sc + c.phaseAdjust) * inv2pi;
        s.last = __builtin_amdgcn_cosf(static_cast<float>(rev2 - rint(rev2)));
    } else {
        glibc235::sincosf_glibc(trigArg, &s.fbQ, &s.fbI);
        s.last = trigArg;
    }
}

__device__ __forceinline__ float nco_out(float trigArg, const PllCoef &c)
{
    return glibc235::cosf_glibc(trigArg * c.ncoScale + c.phaseAdjust);
}

__device__ __forceinline__ PllState load_state(const float *st)
{
    // fr: the angle whose cosine / sine the carried feedback pair is
    return PllState{st[0], st[1], st[2], st[3], st[4], st[5], atan2f(st[3], st[2]) * 0.15915494309189533577f};
}
__device__ __forceinline__ void store_state(float *st, const PllState &s)
{
    st[0] = s.integ; st[1] = s.phase; st[2] = s.fbI; st[3] = s.fbQ; st[4] = s.last; st[5] = s.off;
}

// ---- serial form: one lane walks the block -------------------------------------------------
template <int MATH>
__global__ void pll_serial_kernel(const float *__restrict__ in, size_t n, float *__restrict__ out, float *__restrict__ state,
                                  PllCoef c)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    PllState s = load_state(state);
    out[0] = s.last;
    if (n == 0) return;
    float vn = in[0];                                     // next sample, fetched one step ahead of the chain
    for (size_t k = 0; k < n; k++) {
        const float v = vn;
        vn = in[k + 1 < n ? k + 1 : k];
        pll_step<MATH>(s, v, c);
        out[k + 1] = s.last;                              // kExact: the raw trigArg, see nco_out_kernel
    }
    if (MATH == kExact) s.last = nco_out(s.last, c);
    store_state(state, s);
}

// kExact, second pass: out[k] = cosf(trigArg[k]*ncoScale + phaseAdjust) for k = 1..n, in place
__global__ void nco_out_kernel(float *__restrict__ out, size_t n, PllCoef c)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) out[k + 1] = nco_out(out[k + 1], c);
}

// ---- parallel in time -------------------------------------------------------------------------
// seg[s*16 + 0..5]  state at the END of segment s      (after sample a_s + L - 1)
// seg[s*16 + 8..9]  (integ, phase) this lane had at the START of segment s (after its warm-up)
__global__ void pll_segments_kernel(const float *__restrict__ in, long n, float *__restrict__ out,
                                    const float *__restrict__ state, PllCoef c, int L, int W, long nseg,
                                    float *__restrict__ seg, const float *__restrict__ hdr)
{
    const long sg = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (sg >= nseg) return;
    const long a = sg * L;
    const long b = a + L < n ? a + L : n;
    const PllState s0 = load_state(state);
    PllState s = s0;
    long k = 0;
    if (a > W) {
        // warm start W samples early; hdr[5..7] = {phase at the start of the previous call, its length, valid}
        k = a - W;
        const float slope = hdr[7] != 0.0f ? (s0.phase - hdr[5]) / hdr[6] : 0.0f;
        s.phase = s0.phase + slope * static_cast<float>(k);
        s.off = s0.off + static_cast<float>(k);
        const float trigArg = static_cast<float>(c.w * static_cast<double>(s.off) + static_cast<double>(s.phase));
        const double rev = static_cast<double>(trigArg) * 0.15915494309189533577;
        const float fr = static_cast<float>(rev - rint(rev));
        s.fr = fr;
        s.fbI = __builtin_amdgcn_cosf(fr);
        s.fbQ = __builtin_amdgcn_sinf(fr);
    }
    // The recurrence is one long dependency chain; the samples it eats are not part of it.  They are
    // fetched one step ahead so that a load's latency is never on the chain.
    float vn = k < b ? in[k] : 0.0f;
    for (; k < a; k++) {                                  // warm-up (or exact replay from the block start)
        const float v = vn;
        vn = in[k + 1 < b ? k + 1 : k];
        pll_step<kFast>(s, v, c);
    }
    seg[sg * 16 + 8] = s.integ;
    seg[sg * 16 + 9] = s.phase;
    if (sg == 0) out[0] = s0.last;
    for (; k < b; k++) {
        const float v = vn;
        vn = in[k + 1 < b ? k + 1 : k];
        pll_step<kFast>(s, v, c);
        out[k + 1] = s.last;
    }
    store_state(seg + sg * 16, s);
}

// The loop cannot tell phases apart that round to the same float32 trigArg, so the merge
// tolerance follows that grid: base + 2 ulp(trigArg) at the end of the block (ulp grows from 1e-3
// at 1e4 rad to 0.25 at 3e6 rad -- the reference's own resolution loss).
__device__ __forceinline__ float pll_trig_ulp(const float *state, long n, const PllCoef &c)
{
    const float top = static_cast<float>(c.w * (static_cast<double>(state[5]) + static_cast<double>(n)));
    return __uint_as_float((__float_as_uint(top) & 0x7f800000u)) * 1.1920929e-7f;   // 2^(e-23)
}
__device__ __forceinline__ float pll_phase_tol(float base, const float *state, long n, const PllCoef &c)
{
    return base + 2.0f * pll_trig_ulp(state, n, c);
}
// the integrator moves by Ki * (phase error) per sample, and the phase error is only known to that
// grid: two trajectories cannot agree better than a couple of such steps
__device__ __forceinline__ float pll_integ_tol(float base, const float *state, long n, const PllCoef &c)
{
    return base + 2.0f * c.Ki * pll_trig_ulp(state, n, c);
}

// mark every segment whose start state differs from its predecessor's end state by more than the
// merge tolerance; record the largest differences seen among the accepted ones (diagnostics)
__global__ void pll_check_kernel(const float *__restrict__ seg, long nseg, unsigned long long *__restrict__ badmask,
                                 float tol_phase_base, float tol_integ, unsigned *__restrict__ diag,
                                 const float *__restrict__ state, long n, PllCoef c)
{
    const long sg = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (sg < 1 || sg >= nseg) return;
    const float tol_phase = pll_phase_tol(tol_phase_base, state, n, c);
    tol_integ = pll_integ_tol(tol_integ, state, n, c);
    const float di = fabsf(seg[sg * 16 + 8] - seg[(sg - 1) * 16 + 0]);
    const float dp = fabsf(seg[sg * 16 + 9] - seg[(sg - 1) * 16 + 1]);
    if (!(dp <= tol_phase && di <= tol_integ)) {
        atomicOr(badmask + sg / 64, 1ull << (sg % 64));
    } else {
        atomicMax(diag + 3, __float_as_uint(dp));   // non-negative floats order like their bit patterns
        atomicMax(diag + 4, __float_as_uint(di));
    }
}

// Repair (a no-op when every segment merged).  A mismatching segment whose predecessor is valid is
// walked again from the predecessor's (true) end state; then its successor is judged again against
// the new end state.  Mismatching segments that are not neighbours do not depend on each other, so
// every round repairs all of them at once, one lane each, and a run of r consecutive bad segments
// takes r rounds: typically one or two rounds of one segment's time instead of one lane walking all
// of them in turn.  The result is the serial recurrence's, whatever the order.  Then publish the
// block's end state.
constexpr int kRepairThreads = 1024;
__global__ __launch_bounds__(kRepairThreads) void pll_repair_kernel(
    const float *__restrict__ in, long n, float *__restrict__ out, float *__restrict__ state, PllCoef c, int L, long nseg,
    float *__restrict__ seg, unsigned long long *__restrict__ badmask, float tol_phase_base, float tol_integ,
    unsigned *__restrict__ n_repaired, float *__restrict__ hdr)
{
    __shared__ int any_todo;
    const float tol_phase = pll_phase_tol(tol_phase_base, state, n, c);
    tol_integ = pll_integ_tol(tol_integ, state, n, c);
    auto is_bad = [&](long sg) { return (badmask[sg / 64] >> (sg % 64)) & 1ull; };
    unsigned repaired = 0;
    for (;;) {
        if (threadIdx.x == 0) any_todo = 0;
        __syncthreads();
        // this round's work is fixed before anything changes: bad segments with a valid predecessor
        // (their successors are then not in the round, so a lane's updates touch nobody else's input)
        constexpr int kMaxPer = 8;                     // segments per lane and round (nseg <= 8192 per round; more: next round)
        long todo[kMaxPer];
        int nt = 0;
        for (long sg = 1 + threadIdx.x; sg < nseg && nt < kMaxPer; sg += kRepairThreads)
            if (is_bad(sg) && !is_bad(sg - 1)) todo[nt++] = sg;
        if (nt) any_todo = 1;
        __syncthreads();
        if (!any_todo) break;
        for (int i = 0; i < nt; i++) {
            const long sg = todo[i];
            PllState s = load_state(seg + (sg - 1) * 16);   // true state at the start of segment sg
            const long a = sg * L, b = a + L < n ? a + L : n;
            float vn = a < b ? in[a] : 0.0f;
            for (long k = a; k < b; k++) {
                const float v = vn;
                vn = in[k + 1 < b ? k + 1 : k];
                pll_step<kFast>(s, v, c);
                out[k + 1] = s.last;
            }
            store_state(seg + sg * 16, s);
            repaired++;
            atomicAnd(badmask + sg / 64, ~(1ull << (sg % 64)));
            if (sg + 1 < nseg) {
                const bool merged = fabsf(seg[(sg + 1) * 16 + 9] - s.phase) <= tol_phase &&
                                    fabsf(seg[(sg + 1) * 16 + 8] - s.integ) <= tol_integ;
                if (merged) atomicAnd(badmask + (sg + 1) / 64, ~(1ull << ((sg + 1) % 64)));
                else atomicOr(badmask + (sg + 1) / 64, 1ull << ((sg + 1) % 64));
            }
        }
        __threadfence();
        __syncthreads();
    }
    if (n_repaired && repaired) atomicAdd(n_repaired, repaired);
    if (threadIdx.x == 0) {
        // remember where this call's phase started, for the next call's extrapolation
        hdr[5] = state[1];
        hdr[6] = static_cast<float>(n);
        hdr[7] = 1.0f;
        PllState e = load_state(seg + (nseg - 1) * 16);
        store_state(state, e);
    }
}

__global__ void libm_eval_kernel(int fn, const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                 float *__restrict__ out)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = fn == 0 ? glibc235::sinf_glibc(a[i]) : fn == 1 ? glibc235::cosf_glibc(a[i]) : glibc235::atan2f_glibc(a[i], b[i]);
}

PllCoef make_coef(float freq, float Fs, float ncoScale, float phaseAdjust, float normBandwidth)
{
    PllCoef c;
    const float Cp = 2.666f, Ci = 3.555f;   // float Cp = 2.666 in the reference
    c.Kp = normBandwidth * Cp;
    c.Ki = (normBandwidth * normBandwidth) * Ci;
    c.ncoScale = ncoScale;
    c.phaseAdjust = phaseAdjust;
    c.w = 2 * 3.14159265358979323846 * static_cast<double>(freq / Fs);
    return c;
}

#define FMRX_LAUNCH_CHECK(name)                                                                   \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) return fail(FMRX_EHIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

int k_fm_pll(const float *d_in, size_t n, float *d_out, float *d_state, float freq, float Fs, float ncoScale,
             float phaseAdjust, float normBandwidth, int fast, hipStream_t s)
{
    const PllCoef c = make_coef(freq, Fs, ncoScale, phaseAdjust, normBandwidth);
    if (fast) {
        hipLaunchKernelGGL(pll_serial_kernel<kFast>, dim3(1), dim3(64), 0, s, d_in, n, d_out, d_state, c);
        FMRX_LAUNCH_CHECK("pll_serial");
        return FMRX_OK;
    }
    hipLaunchKernelGGL(pll_serial_kernel<kExact>, dim3(1), dim3(64), 0, s, d_in, n, d_out, d_state, c);
    FMRX_LAUNCH_CHECK("pll_serial");
    if (n) {
        hipLaunchKernelGGL(nco_out_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, d_out, n, c);
        FMRX_LAUNCH_CHECK("nco_out");
    }
    return FMRX_OK;
}

int k_libm_eval(int fn, const float *d_a, const float *d_b, size_t n, float *d_out, hipStream_t s)
{
    if (n == 0) return FMRX_OK;
    hipLaunchKernelGGL(libm_eval_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, fn, d_a, d_b, n, d_out);
    FMRX_LAUNCH_CHECK("libm_eval");
    return FMRX_OK;
}

size_t pll_parallel_scratch_floats(size_t n)
{
    const size_t nseg = n / kPllSegment + 2;
    return 8 + nseg * 16 + 2 * (nseg / 64 + 2);
}

int k_fm_pll_parallel(const float *d_in, size_t n, float *d_out, float *d_state, float freq, float Fs, float ncoScale,
                      float phaseAdjust, float normBandwidth, float *d_scratch, const Options &o, hipStream_t s)
{
    int L = kPllSegment, W = kPllWarmup;
    if (o.pll_warmup >= 0 && o.pll_warmup <= 65536) W = o.pll_warmup;                 // tuning: warm-up samples per lane
    if (o.pll_segment >= kPllSegment && o.pll_segment <= 65536) L = o.pll_segment;   // tuning: samples per lane (scratch is sized for >= kPllSegment)
    if (n < static_cast<size_t>(4 * L))   // nothing to gain
        return k_fm_pll(d_in, n, d_out, d_state, freq, Fs, ncoScale, phaseAdjust, normBandwidth, 1, s);
    const PllCoef c = make_coef(freq, Fs, ncoScale, phaseAdjust, normBandwidth);
    const long nseg = static_cast<long>((n + L - 1) / L);
    // scratch: [2] repaired-segment counter (u32), [3],[4] largest accepted |dphase|,|dinteg| (diagnostics),
    // [5..7] previous call's start phase / length / valid; [8..] per-segment records, then the mismatch bitmask
    unsigned *n_repaired = reinterpret_cast<unsigned *>(d_scratch + 2);
    float *seg = d_scratch + 8;
    unsigned long long *badmask = reinterpret_cast<unsigned long long *>(seg + (nseg + 1) * 16);
    FMRX_HIP(hipMemsetAsync(badmask, 0, (nseg / 64 + 1) * sizeof(unsigned long long), s));
    const unsigned grid = static_cast<unsigned>((nseg + 63) / 64);
    hipLaunchKernelGGL(pll_segments_kernel, dim3(grid), dim3(64), 0, s, d_in, static_cast<long>(n), d_out, d_state, c, L, W,
                       nseg, seg, d_scratch);
    FMRX_LAUNCH_CHECK("pll_segments");
    hipLaunchKernelGGL(pll_check_kernel, dim3(grid), dim3(64), 0, s, seg, nseg, badmask, kPllTolPhase, kPllTolInteg,
                       reinterpret_cast<unsigned *>(d_scratch), d_state, static_cast<long>(n), c);
    FMRX_LAUNCH_CHECK("pll_check");
    hipLaunchKernelGGL(pll_repair_kernel, dim3(1), dim3(kRepairThreads), 0, s, d_in, static_cast<long>(n), d_out, d_state, c, L, nseg, seg,
                       badmask, kPllTolPhase, kPllTolInteg, n_repaired, d_scratch);
    FMRX_LAUNCH_CHECK("pll_repair");
    return FMRX_OK;
}

}  // namespace fmrx

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
//     of L = 64 samples, one lane each; a lane starts W samples early from an estimate
//     of the loop's state there and runs the same recurrence; after the warm-up it
//     agrees with the serial trajectory TO WITHIN THE GRID, not bit for bit.  The
//     estimate (option pll_start): 1 (default) -- the closed-form phase detector reads
//     only the SIGN of the input, so the locked loop is a linear time-invariant system
//     driven by a staircase that climbs half a turn per sign change: its state at
//     every 64th sample follows from the signs alone (pll_lti_chunks_kernel +
//     lti_start_state below), W = 64 true steps then put the lane on the float32 grid;
//     0 -- the block's initial state with the phase drift of the previous block,
//     W = 512.  A second kernel compares, for every segment, the state a lane had at
//     its segment start with the state its predecessor ended on, against a tolerance
//     (kPllTolPhase + 2 ulp(trigArg) on the phase, modulo whole turns; kPllTolInteg +
//     2 or 6 Ki ulp(trigArg) on the integrator); a third walks the
//     recurrence serially from every segment that is outside it (loop not locked:
//     stream start, drop-outs, phase jumps, no pilot) until it is inside again.
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
    float integ_tol_ulps = 2.0f;   // merge tolerance on the integrator, in Ki * ulp(trigArg)
    // the locked loop as a linear system (pll_start = 1): Q = A^64, G = one chunk's response to the constant 1 (host: make_coef)
    double q00 = 1.0, q01 = 0.0, q10 = 0.0, q11 = 1.0, g0 = 0.0, g1 = 0.0;
};

struct PllState {
    float integ, phase, fbI, fbQ, last, off;
    float fr;   // FAST only: trigArg / 2 pi of the step that produced fbI/fbQ, reduced to [-0.5, 0.5] revolutions
};

enum PllMath { kExact = 0, kFast = 1 };
constexpr int kLtiChunk = 64;   // samples per chunk of the linear-system start (pll_lti_*_kernel)

// One step of the recurrence.  Both forms leave the RAW trigArg in s.last: the NCO output
// cosf(trigArg*ncoScale + phaseAdjust) is not part of the recurrence and is applied afterwards, in parallel,
// by nco_out_kernel<MATH>.  kExact: the reference's operations in its order, glibc's functions.  kFast: see the
// file header; the feedback pair (fbI, fbQ) is carried as the angle s.fr it is the cosine / sine of.
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
        if (MATH == kFast) {
            s.fbI = __builtin_amdgcn_cosf(s.fr);
            s.fbQ = __builtin_amdgcn_sinf(s.fr);
        }
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
        const double rev = static_cast<double>(trigArg) * 0.15915494309189533577;
        s.fr = static_cast<float>(rev - rint(rev));                    // [-0.5, 0.5] revolutions
    } else {
        glibc235::sincosf_glibc(trigArg, &s.fbQ, &s.fbI);
    }
    s.last = trigArg;
}

// The exact step without control flow, for a wave whose lanes are different receivers (pll_channels_kernel): the same float
// operations as pll_step<kExact>, every choice inside atan2f / sincosf a select (glibc_libm.hpp: *_flat, pinned against the C
// library like the originals).  Those variants cover the ordinary arguments -- finite non-zero phase-detector inputs, |trigArg|
// >= 120 (every sample of a stream but its first 240) --; one wave-uniform test each sends the whole wave through the general
// functions when any lane holds anything else (silence, a drop-out, a stream that has just started).  w24: 4/pi's 24 windows, in LDS.
__device__ __forceinline__ void pll_step_exact_flat(PllState &s, float v, const PllCoef &c, const uint32_t *w24)
{
    const float eI = v * s.fbI;
    const float eQ = v * (-1 * s.fbQ);
    float eD;
    if (__builtin_expect(!__any(!glibc235::atan2f_flat_ok(eQ, eI)), 1)) eD = glibc235::atan2f_flat(eQ, eI);
    else eD = glibc235::atan2f_glibc(eQ, eI);
    s.integ = s.integ + c.Ki * eD;
    const float pe = c.Kp * eD;
    s.phase = (s.phase + pe) + s.integ;
    s.off += 1;
    const float trigArg = static_cast<float>(c.w * static_cast<double>(s.off) + static_cast<double>(s.phase));
    // 120 <= trigArg < 2^25 (a stream's samples 240 .. 6.7e7): the table look-up of the reduction is two selects between constants;
    // beyond, the LDS table; below 120 (or not finite), the general function
    if (__builtin_expect(!__any(!glibc235::sincosf_mid_ok(trigArg)), 1)) glibc235::sincosf_mid_flat(trigArg, &s.fbQ, &s.fbI);
    else if (!__any(!glibc235::sincosf_large_ok(trigArg))) glibc235::sincosf_large_flat(trigArg, w24, &s.fbQ, &s.fbI);
    else glibc235::sincosf_glibc(trigArg, &s.fbQ, &s.fbI);
    s.last = trigArg;
}

// The fast step for an ordinary sample (|v| in (1e-20, 1e20): everything but exact zeros, denormal products and
// non-finite values), without a branch: what the lanes of the parallel form run.  Identical arithmetic to
// pll_step<kFast>'s closed-form path; written with selects so that the recurrence's critical path holds no
// exec-mask manipulation (a divergent branch puts several scalar instructions and their vector<->scalar
// hand-offs between two samples).
__device__ __forceinline__ void pll_step_clean(PllState &s, float v, const PllCoef &c)
{
    const float half = s.fr >= 0.0f ? 0.5f : -0.5f;
    const float turn = v > 0.0f ? 0.0f : half;
    const float eD = (turn - s.fr) * 6.28318530717958647692f;
    s.integ = s.integ + c.Ki * eD;
    const float pe = c.Kp * eD;
    s.phase = (s.phase + pe) + s.integ;
    s.off += 1;
    const float trigArg = static_cast<float>(c.w * static_cast<double>(s.off) + static_cast<double>(s.phase));
    const double rev = static_cast<double>(trigArg) * 0.15915494309189533577;
    s.fr = static_cast<float>(rev - rint(rev));   // (a float32-only reduction -- product split by fma -- was measured: no faster)
    s.last = trigArg;
}
__device__ __forceinline__ bool pll_ordinary(float v) { return fabsf(v) > 1e-20f && fabsf(v) < 1e20f; }

template <int MATH>
__device__ __forceinline__ float nco_out(float trigArg, const PllCoef &c)
{
    const float a = trigArg * c.ncoScale + c.phaseAdjust;
    if (MATH == kExact) return glibc235::cosf_glibc(a);
    const double rev = static_cast<double>(a) * 0.15915494309189533577;
    return __builtin_amdgcn_cosf(static_cast<float>(rev - rint(rev)));
}

// state[2], state[3] (feedbackI, feedbackQ) and state[4] (lastOut) from the raw end of a run
template <int MATH>
__device__ __forceinline__ void finish_state(PllState &s, const PllCoef &c)
{
    if (MATH == kFast) {
        s.fbI = __builtin_amdgcn_cosf(s.fr);
        s.fbQ = __builtin_amdgcn_sinf(s.fr);
    }
    s.last = nco_out<MATH>(s.last, c);
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
// One wave: all 64 lanes move the samples (64 at a time, coalesced, through LDS), lane 0 walks the recurrence on LDS only --
// no memory access sits between two steps of the chain.  kExact runs the branch-free forms of glibc's functions where they are
// defined (pll_step_exact_flat: same values), the general ones elsewhere.
template <int MATH>
__global__ __launch_bounds__(64) void pll_serial_kernel(const float *__restrict__ in, size_t n, float *__restrict__ out, float *__restrict__ state,
                                                        PllCoef c)
{
    __shared__ uint32_t w24[24];
    __shared__ float lin[64], lout[64];
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    if (lane < 24) w24[lane] = glibc235::inv_pio4(lane);
    PllState s = load_state(state);
    if (lane == 0) out[0] = s.last;
    for (size_t b = 0; b < n; b += 64) {
        const size_t m = n - b < 64 ? n - b : 64;
        __syncthreads();                                      // the previous batch's results have left LDS
        if (static_cast<size_t>(lane) < m) lin[lane] = in[b + lane];
        __syncthreads();
        if (lane == 0) {
            for (size_t k = 0; k < m; k++) {
                if (MATH == kExact) pll_step_exact_flat(s, lin[k], c, w24);
                else pll_step<MATH>(s, lin[k], c);
                lout[k] = s.last;                             // the raw trigArg, see nco_out_kernel
            }
        }
        __syncthreads();
        if (static_cast<size_t>(lane) < m) out[b + lane + 1] = lout[lane];
    }
    if (lane == 0) {
        finish_state<MATH>(s, c);
        store_state(state, s);
    }
}

// second pass, in parallel: out[k] = cosf(trigArg[k]*ncoScale + phaseAdjust) for k = 1..n, in place
template <int MATH>
__global__ void nco_out_kernel(float *__restrict__ out, size_t n, PllCoef c)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n) out[k + 1] = nco_out<MATH>(out[k + 1], c);
}

// ---- where the lanes start, second form: the locked loop as a linear system ---------------------------------
// The fast phase detector reads only the SIGN of its input: its output is  T - theta,  theta = trigArg / 2 pi in
// revolutions, T the point of the lattice Z (input > 0) or Z + 1/2 (input < 0) nearest theta.  While the loop is
// locked theta follows the pilot, so T is a staircase that climbs 1/2 at every sign change of the input: it does not
// depend on the loop's state at all.  With  C[k] = T[k] - f * off[k]  (f = freq/Fs revolutions per sample) the
// recurrence is linear and time-invariant in (phi, iota) = (phase, integrator) / 2 pi:
//     iota' = iota + Ki (C - phi),   phi' = phi + Kp (C - phi) + iota'      i.e.  s' = A s + B C,
// its memory is ~75 samples (|eig A| = 1 - 0.0133).  So the state at every 64th sample follows from the signs alone:
//   pll_lti_chunks_kernel  one thread per 64-sample chunk: its count of sign changes and its zero-state response to
//                          (T - T_chunk_start) - f j, j = 0..63; prefix of the counts inside a workgroup, workgroup totals;
//   lti_start_state        (in pll_segments_kernel) T at a chunk's start from those prefixes, then
//                          s[chunk i] = sum_{m=1..kLtiTerms} (A^64)^(m-1) R[i-m]  (+ (A^64)^i s[0] near the block start).
// What this ignores is the float32 grid of trigArg (the true loop sees theta rounded to ulp(trigArg)): the lanes
// therefore still run W true steps from there before their own segment.  A sign pattern that is not a locked
// pilot's (a glitch: two sign changes within a sample or two) breaks the staircase rule; the lanes' ends then do not
// meet their successors' starts and pll_repair_kernel walks those stretches serially, as before.
constexpr int kLtiTerms = 20;          // (A^64)^20 ~ 5e-8: what is dropped of the state 1280 samples back
struct LtiMat {
    double a00, a01, a10, a11, b0, b1;   // s' = A s + B x, s = (phi, iota)
};
__device__ __forceinline__ LtiMat lti_of(const PllCoef &c)
{
    const double Kp = c.Kp, Ki = c.Ki;
    return LtiMat{1.0 - Kp - Ki, 1.0, -Ki, 1.0, Kp + Ki, Ki};
}
__device__ __forceinline__ void lti_step(const LtiMat &m, double &phi, double &iota, double x)
{
    const double p = m.a00 * phi + m.a01 * iota + m.b0 * x;
    iota = m.a10 * phi + m.a11 * iota + m.b1 * x;
    phi = p;
}

// rec[4 i ..] = {R_phi, R_iota: zero-state response of chunk i to its own staircase; the climb of the chunks in front of
// it inside its workgroup of 64 chunks; its own climb (sign changes inside it and into the next chunk's first sample, / 2)};
// wgtot[w] = climb of workgroup w's 64 chunks.  A lane reads its chunk as 16 aligned 16-byte groups.
__global__ __launch_bounds__(64) void pll_lti_chunks_kernel(const float *__restrict__ in, long n, PllCoef c, long nchunk,
                                                            double *__restrict__ rec, double *__restrict__ wgtot)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const long i = static_cast<long>(blockIdx.x) * 64 + threadIdx.x;
    const LtiMat m = lti_of(c);
    const double f = c.w * 0.15915494309189533577;
    double phi = 0.0, iota = 0.0, dT = 0.0;
    if (i < nchunk) {
        const long k0 = i * kLtiChunk;
        const f4 *in4 = reinterpret_cast<const f4 *>(in + k0);
        const long last4 = (n + 3) / 4 + 1 - k0 / 4;                   // groups of this chunk that are safe to read (host contract)
        f4 v[kLtiChunk / 4];
#pragma unroll
        for (int q = 0; q < kLtiChunk / 4; q++) v[q] = in4[q < last4 ? q : last4];
        const float nxt = k0 + kLtiChunk < n ? in[k0 + kLtiChunk] : 0.0f;
        bool pos = v[0][0] > 0.0f;
#pragma unroll
        for (int j = 0; j < kLtiChunk; j++) {
            lti_step(m, phi, iota, dT - f * j);
            const long k = k0 + j + 1;
            const float vn = j + 1 < kLtiChunk ? v[(j + 1) / 4][(j + 1) % 4] : nxt;
            const bool pn = k < n ? vn > 0.0f : pos;
            dT += pn != pos ? 0.5 : 0.0;
            pos = pn;
        }
    }
    // exclusive prefix of the climbs over the workgroup's 64 chunks (one wave)
    double incl = dT;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(incl, d, 64);
        if (static_cast<int>(threadIdx.x) >= d) incl += o;
    }
    if (i < nchunk) {
        rec[4 * i + 0] = phi;
        rec[4 * i + 1] = iota;
        rec[4 * i + 2] = incl - dT;
        rec[4 * i + 3] = dT;
    }
    if (threadIdx.x == 63) wgtot[blockIdx.x] = incl;
}

// (integ, phase) the locked loop has in front of sample 64 i, from the chunk records: Horner over the kLtiTerms chunks
// behind it, oldest first; near the block start the true initial state takes the place of what was dropped.  `wg_climb`
// = climb of all chunks in front of workgroup i / 64 (the caller sums wgtot once per wave).
struct LtiStart {
    double q00, q01, q10, q11, g0, g1;      // Q = A^64; G = one chunk's response to the constant 1
    double f, phi0, iota0, off0, T0;
};
__device__ __forceinline__ LtiStart lti_start_setup(const float in0, const float *state, const PllCoef &c)
{
    LtiStart L;
    L.q00 = c.q00; L.q01 = c.q01; L.q10 = c.q10; L.q11 = c.q11; L.g0 = c.g0; L.g1 = c.g1;
    const double inv2pi = 0.15915494309189533577;
    L.f = c.w * inv2pi;
    L.phi0 = static_cast<double>(state[1]) * inv2pi;
    L.iota0 = static_cast<double>(state[0]) * inv2pi;
    L.off0 = static_cast<double>(state[5]);
    // the lattice point the detector sees at sample 0: nearest to theta (the float32 trigArg the state stands for)
    const float trig0 = static_cast<float>(c.w * L.off0 + static_cast<double>(state[1]));
    const double th0 = static_cast<double>(trig0) * inv2pi;
    L.T0 = in0 > 0.0f ? rint(th0) : rint(th0 - 0.5) + 0.5;
    return L;
}
// wg_climb0 = climb of all chunks in front of workgroup wg0 (of 64 chunks); chunk i lies in wg0 or wg0 + 1.
// The records of the kLtiTerms chunks behind chunk i (lti_fetch) are requested by the caller together with everything else its
// prologue reads, ahead of the wave's sum of the workgroup totals: one memory round trip instead of four in a row.
struct LtiRecs {
    double a[kLtiTerms], b[kLtiTerms], t[kLtiTerms];          // a chunk's zero-state response (2) and the staircase's prefix at its start
    double tot0, totm;                                         // climb of workgroups wg0 and wg0 - 1
};
__device__ __forceinline__ void lti_fetch(LtiRecs &R, const double *__restrict__ rec, const double *__restrict__ wgtot, long i, long wg0)
{
    R.tot0 = wgtot[wg0];
    R.totm = wg0 > 0 ? wgtot[wg0 - 1] : 0.0;
#pragma unroll
    for (int t = 0; t < kLtiTerms; t++) {                     // term t: chunk j = i - (kLtiTerms - t), oldest first
        const long jj = i - (kLtiTerms - t);
        const long j = jj > 0 ? jj : 0;
        R.a[t] = rec[4 * j + 0];
        R.b[t] = rec[4 * j + 1];
        R.t[t] = rec[4 * j + 2];
    }
}
__device__ __forceinline__ void lti_start_state(const LtiStart &L, const LtiRecs &R, long i, long wg0, double wg_climb0, float &integ,
                                                float &phase)
{
    double r0[kLtiTerms], r1[kLtiTerms];
#pragma unroll
    for (int t = 0; t < kLtiTerms; t++) {
        const long jj = i - (kLtiTerms - t);
        const long j = jj > 0 ? jj : 0;
        const long wj = j / 64;
        // T at the start of chunk j, from its workgroup's base (wg0 - 1, wg0 or wg0 + 1) and its prefix inside it
        const double wbase = wj == wg0 ? wg_climb0 : (wj > wg0 ? wg_climb0 + R.tot0 : wg_climb0 - R.totm);
        const double base = (L.T0 + wbase + R.t[t]) - L.f * (L.off0 + static_cast<double>(j * kLtiChunk)) - L.phi0;
        r0[t] = R.a[t] + base * L.g0;
        r1[t] = R.b[t] + base * L.g1;
    }
    double p = 0.0, q = 0.0;
#pragma unroll
    for (int t = 0; t < kLtiTerms; t++) {
        const long jj = i - (kLtiTerms - t);
        if (jj == 0) {                                        // the block's true initial state, in deviations from (phi0, 0)
            p = 0.0;
            q = L.iota0;
        }
        const double np = L.q00 * p + L.q01 * q + r0[t];
        const double nq = L.q10 * p + L.q11 * q + r1[t];
        p = jj >= 0 ? np : p;
        q = jj >= 0 ? nq : q;
    }
    if (i == 0) {
        p = 0.0;
        q = L.iota0;
    }
    integ = static_cast<float>(q * 6.28318530717958647692);
    phase = static_cast<float>((L.phi0 + p) * 6.28318530717958647692);
}

// The loop cannot tell phases apart that round to the same float32 trigArg, so the merge
// tolerance follows that grid: base + 2 ulp(trigArg) at the end of the block (ulp grows from 1e-3
// at 1e4 rad to 0.25 at 3e6 rad -- the reference's own resolution loss).
__device__ __forceinline__ float pll_trig_ulp(const float *state, long n, const PllCoef &c)
{
    const float top = static_cast<float>(c.w * (static_cast<double>(state[5]) + static_cast<double>(n)));
    return __uint_as_float((__float_as_uint(top) & 0x7f800000u)) * 1.1920929e-7f;   // 2^(e-23)
}
// Two phase estimates that differ by whole turns stand for the same loop state (trigArg only enters through sin / cos and the
// NCO's cos): a lane that counted one pilot cycle more or less than its neighbour is not wrong.  |a - b| modulo 2 pi.
__device__ __forceinline__ float pll_phase_dist(float a, float b)
{
    const float d = a - b;
    return fabsf(d - 6.28318530717958647692f * rintf(d * 0.15915494309189533577f));
}
__device__ __forceinline__ float pll_phase_tol(float base, const float *state, long n, const PllCoef &c)
{
    return base + 2.0f * pll_trig_ulp(state, n, c);
}
// the integrator moves by Ki * (phase error) per sample, and the phase error is only known to that
// grid: two trajectories cannot agree better than a couple of such steps
__device__ __forceinline__ float pll_integ_tol(float base, const float *state, long n, const PllCoef &c)
{
    return base + c.integ_tol_ulps * c.Ki * pll_trig_ulp(state, n, c);
}

// ---- parallel in time -------------------------------------------------------------------------
// lanes per workgroup of pll_segments_kernel: one wave (256 was measured: 23.1 vs 21.2 us); also what lets a lane keep its
// chunk records in registers (the register budget of a kernel follows its largest legal workgroup)
constexpr int kSegThreads = 64;
// seg[s*16 + 0..5]  state at the END of segment s      (after sample a_s + L - 1; fbI/fbQ/last finished)
// seg[s*16 + 8..9]  (integ, phase) this lane had at the START of segment s (after its warm-up)
//
// Where a lane starts.  A locked loop is not just "near a constant": its phase carries the ripple of the
// phase detector (a sawtooth per pilot half cycle, +-0.07 rad), and on a pilot that is on frequency that
// ripple repeats with the pilot: every P = Fs / gcd(Fs, freq) samples (240 at 240 kHz: 19 pilot cycles)
// the loop is in the same state again, up to its slow drift.  So a lane does not start "W samples early"
// but at the last multiple of P (counted from the block start, where the true state is known) that is at
// least W samples early, from the block's initial state plus the drift: its guess is then already within a
// few grid steps of the true trajectory, and W only has to cover what the drift estimate misses (64
// samples instead of the 768 it takes to forget a guess that ignores the ripple).  P = 0: no alignment.
__global__ __launch_bounds__(kSegThreads) void pll_segments_kernel(const float *__restrict__ in, long n, float *__restrict__ out,
                                    const float *__restrict__ state, PllCoef c, int L, int W, int P, long nseg,
                                    float *__restrict__ seg, float *hdr, const double *__restrict__ lti_rec,
                                    const double *__restrict__ lti_wgtot, unsigned long long *__restrict__ badmask,
                                    float tol_phase_base, float tol_integ_base)
{
    const long sg = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;                        // a wave = 64 consecutive segments = one word of the mismatch mask
    // linear-system start: the climb of the chunk workgroups in front of the one this wave's first lane starts in, summed by
    // the whole wave (its 64 lanes start in that workgroup of 64 chunks or the next one)
    long lti_wg0 = 0;
    double lti_climb0 = 0.0;
    // Everything the prologue reads is requested here, in one go and in front of the first use of any of it (the wave's sum
    // of the workgroup totals): the carried state, the block's first sample, this lane's chunk records, its first input
    // groups.  Lanes past the block read what the last lane reads.
    const long sgc = sg < nseg ? sg : nseg - 1;
    const long a = sgc * L;
    const long b = a + L < n ? a + L : n;
    long k = 0;
    if (a > W) {
        k = a - W;
        if (P > 0) k -= k % P;
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 *in4 = reinterpret_cast<const f4 *>(in);
    const long last4 = (n + 3) / 4 + 1;                   // last group that is safe to read (floats up to n+10)
    auto grp = [&](long g) { return in4[g < last4 ? g : last4]; };
    float st[6];
#pragma unroll
    for (int u = 0; u < 6; u++) st[u] = state[u];
    const float in0 = in[0];
    long g = k / 4;
    f4 q0 = grp(g), q1 = grp(g + 1), q2 = grp(g + 2);
    LtiRecs R;
    if (lti_rec) {
        const long a0 = (sg - lane) * L;                       // the wave's first lane
        const long k0 = a0 > W ? a0 - W : 0;
        lti_wg0 = (k0 / kLtiChunk) / 64;
        double part = 0.0;
        // the first 512 totals as 8 loads in flight per lane (a loop of dependent iterations pays a memory round trip each)
        double tw[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const long u = lane + 64 * r;
            tw[r] = lti_wgtot[u < lti_wg0 ? u : 0];
        }
        lti_fetch(R, lti_rec, lti_wgtot, k / kLtiChunk, lti_wg0);
        __builtin_amdgcn_sched_barrier(0);                    // every request above is out before the first sum waits for one
#pragma unroll
        for (int r = 0; r < 8; r++) part += lane + 64 * r < lti_wg0 ? tw[r] : 0.0;
        for (long u = lane + 512; u < lti_wg0; u += 64) part += lti_wgtot[u];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        lti_climb0 = part;
    }
    if (sg >= nseg) return;
    // fr: the angle whose cosine / sine the carried feedback pair is (load_state)
    const PllState s0{st[0], st[1], st[2], st[3], st[4], st[5], atan2f(st[3], st[2]) * 0.15915494309189533577f};
    PllState s = s0;
    if (k > 0) {
        if (lti_rec) {
            // the linear system's state in front of sample k (a multiple of 64: host contract)
            const LtiStart Ls = lti_start_setup(in0, st, c);
            lti_start_state(Ls, R, k / kLtiChunk, lti_wg0, lti_climb0, s.integ, s.phase);
        } else {
            // hdr[5..7] = {phase at the start of the previous call, its length, valid}: the drift
            const float slope = hdr[7] != 0.0f ? (s0.phase - hdr[5]) / hdr[6] : 0.0f;
            s.phase = s0.phase + slope * static_cast<float>(k);
        }
        s.off = s0.off + static_cast<float>(k);
        const float trigArg = static_cast<float>(c.w * static_cast<double>(s.off) + static_cast<double>(s.phase));
        const double rev = static_cast<double>(trigArg) * 0.15915494309189533577;
        s.fr = static_cast<float>(rev - rint(rev));
    }
    // The recurrence is one long dependency chain; the samples it eats are not part of it, and a load from
    // L2 / HBM takes longer than several steps of the chain.  They are fetched as 16-byte groups, three groups
    // (12 samples, ~1 us of chain) ahead.  k, a and L are multiples of 4 and `in` is 16-byte aligned with at
    // least 12 readable floats behind in[n-1] (host contract), so every group is one aligned global_load_dwordx4.
    for (; k < a; k += 4) {                               // warm-up (or exact replay from the block start)
        const f4 cur = q0;
        q0 = q1;
        q1 = q2;
        q2 = grp(++g + 2);
        // one wave-uniform test per group: an odd sample anywhere in the wave sends the whole wave through the
        // general step (same values for ordinary samples), so the common path has no divergent branch at all
        if (__builtin_expect(__any(!(pll_ordinary(cur.x) && pll_ordinary(cur.y) && pll_ordinary(cur.z) && pll_ordinary(cur.w))), 0)) {
            pll_step<kFast>(s, cur.x, c);
            pll_step<kFast>(s, cur.y, c);
            pll_step<kFast>(s, cur.z, c);
            pll_step<kFast>(s, cur.w, c);
        } else {
            pll_step_clean(s, cur.x, c);
            pll_step_clean(s, cur.y, c);
            pll_step_clean(s, cur.z, c);
            pll_step_clean(s, cur.w, c);
        }
    }
    seg[sg * 16 + 8] = s.integ;
    seg[sg * 16 + 9] = s.phase;
    const float start_integ = s.integ, start_phase = s.phase;
    if (sg == 0) out[0] = s0.last;
    for (; k + 4 <= b; k += 4) {
        const f4 cur = q0;
        q0 = q1;
        q1 = q2;
        q2 = grp(++g + 2);
        float r0, r1, r2, r3;                              // raw trigArg of the four steps
        if (__builtin_expect(__any(!(pll_ordinary(cur.x) && pll_ordinary(cur.y) && pll_ordinary(cur.z) && pll_ordinary(cur.w))), 0)) {
            pll_step<kFast>(s, cur.x, c); r0 = s.last;
            pll_step<kFast>(s, cur.y, c); r1 = s.last;
            pll_step<kFast>(s, cur.z, c); r2 = s.last;
            pll_step<kFast>(s, cur.w, c); r3 = s.last;
        } else {
            pll_step_clean(s, cur.x, c); r0 = s.last;
            pll_step_clean(s, cur.y, c); r1 = s.last;
            pll_step_clean(s, cur.z, c); r2 = s.last;
            pll_step_clean(s, cur.w, c); r3 = s.last;
        }
        // finished NCO values (off the recurrence's dependency chain: the lane's issue slots are mostly idle)
        out[k + 1] = nco_out<kFast>(r0, c);
        out[k + 2] = nco_out<kFast>(r1, c);
        out[k + 3] = nco_out<kFast>(r2, c);
        out[k + 4] = nco_out<kFast>(r3, c);
    }
    for (int i = 0; k < b; k++, i++) {                    // the block's ragged end (last segment only)
        pll_step<kFast>(s, q0[i], c);
        out[k + 1] = nco_out<kFast>(s.last, c);
    }
    finish_state<kFast>(s, c);
    store_state(seg + sg * 16, s);
    // Judge: this lane's start state against its predecessor's end state, to the merge tolerance.  The predecessor is the
    // lane next door (the wave's lanes are 64 consecutive segments and have all arrived here); the wave's first lane has
    // its predecessor in another workgroup: pll_repair_kernel judges those (bit 0 of every mask word is left clear).
    // The mask words are written whole: no memset.
    {
        const float tol_phase = pll_phase_tol(tol_phase_base, st, n, c);   // st: the state as fetched in the prologue
        const float tol_integ = pll_integ_tol(tol_integ_base, st, n, c);
        const float pe_integ = __shfl_up(s.integ, 1, 64), pe_phase = __shfl_up(s.phase, 1, 64);
        bool bad = false;
        float dp = 0.0f, di = 0.0f;
        if (lane > 0) {
            di = fabsf(start_integ - pe_integ);
            dp = pll_phase_dist(start_phase, pe_phase);
            bad = !(dp <= tol_phase && di <= tol_integ);
            if (bad) dp = di = 0.0f;
        }
        const unsigned long long m = __ballot(bad);
        for (int o = 32; o; o >>= 1) {
            dp = fmaxf(dp, __shfl_xor(dp, o, 64));
            di = fmaxf(di, __shfl_xor(di, o, 64));
        }
        if (lane == 0) {
            unsigned *diag = reinterpret_cast<unsigned *>(hdr);
            badmask[sg / 64] = m;
            if (m) hdr[1] = 1.0f;                              // "some segment needs repair"; cleared by pll_repair_kernel
            // the largest accepted differences (diagnostics).  Read before the atomic: hundreds of waves hammering two
            // addresses cost more than the rest
            if (__float_as_uint(dp) > diag[3]) atomicMax(diag + 3, __float_as_uint(dp));   // non-negative floats order like their bit patterns
            if (__float_as_uint(di) > diag[4]) atomicMax(diag + 4, __float_as_uint(di));
        }
    }
}

// Finish (one workgroup): walk every flagged segment again from its predecessor's true end state, round by round, until
// the successors merge (usually there is nothing to do: a locked pilot flags no segment), and store the block's end
// state.  The segments' outputs are finished NCO values already; a repaired segment's are rewritten.
constexpr int kRepairThreads = 256;
__global__ __launch_bounds__(kRepairThreads) void pll_repair_kernel(
    const float *__restrict__ in, long n, float *__restrict__ out, float *__restrict__ state, PllCoef c, int L, long nseg,
    float *__restrict__ seg, unsigned long long *__restrict__ badmask, float tol_phase_base, float tol_integ_base,
    unsigned *__restrict__ n_repaired, float *__restrict__ hdr)
{
    __shared__ int any_todo;
    __shared__ int flagged;
    // Everything the common case (nothing to repair) reads, requested in one go: the state (tolerances, next call's drift
    // record), the lanes' flag, the diagnostics, this thread's first boundary, the last segment's end state.
    unsigned *diag = reinterpret_cast<unsigned *>(hdr);
    const long sg1 = 64 * (1 + static_cast<long>(threadIdx.x));            // the segments whose predecessor ran in another
    const long sg1c = sg1 < nseg ? sg1 : (nseg > 1 ? nseg - 1 : 1);        // workgroup (every 64th) are judged here
    float st[6];
#pragma unroll
    for (int u = 0; u < 6; u++) st[u] = state[u];
    const float h1 = hdr[1];
    const unsigned d3 = diag[3], d4 = diag[4];
    const float b_integ = seg[sg1c * 16 + 8], b_phase = seg[sg1c * 16 + 9], p_integ = seg[(sg1c - 1) * 16 + 0],
                p_phase = seg[(sg1c - 1) * 16 + 1];
    float last[6];
#pragma unroll
    for (int u = 0; u < 6; u++) last[u] = seg[(nseg - 1) * 16 + u];
    __builtin_amdgcn_sched_barrier(0);
    const float tol_phase = pll_phase_tol(tol_phase_base, st, n, c);
    const float tol_integ = pll_integ_tol(tol_integ_base, st, n, c);
    if (threadIdx.x == 0) flagged = h1 != 0.0f;
    __syncthreads();
    {
        auto judge = [&](long sg, float di, float dp) {
            if (!(dp <= tol_phase && di <= tol_integ)) {
                atomicOr(badmask + sg / 64, 1ull);
                flagged = 1;
            } else {
                // the largest accepted differences (diagnostics); compared with the value read above first: an atomic per thread
                // on two addresses costs more than the rest (a stale smaller value only means an atomic that changes nothing)
                if (__float_as_uint(dp) > d3) atomicMax(diag + 3, __float_as_uint(dp));
                if (__float_as_uint(di) > d4) atomicMax(diag + 4, __float_as_uint(di));
            }
        };
        if (sg1 < nseg) judge(sg1, fabsf(b_integ - p_integ), pll_phase_dist(b_phase, p_phase));
        for (long sg = sg1 + 64 * kRepairThreads; sg < nseg; sg += 64 * kRepairThreads)
            judge(sg, fabsf(seg[sg * 16 + 8] - seg[(sg - 1) * 16 + 0]), pll_phase_dist(seg[sg * 16 + 9], seg[(sg - 1) * 16 + 1]));
    }
    __threadfence_block();
    __syncthreads();
    auto is_bad = [&](long sg) { return (__atomic_load_n(badmask + sg / 64, __ATOMIC_RELAXED) >> (sg % 64)) & 1ull; };
    unsigned repaired = 0;
    const bool nothing_to_do = !flagged;                   // the common case: every segment merged
    for (; !nothing_to_do;) {
        if (threadIdx.x == 0) any_todo = 0;
        __syncthreads();
        // this round's work is fixed before anything changes: bad segments with a valid predecessor
        // (their successors are then not in the round, so a lane's updates touch nobody else's input)
        constexpr int kMaxPer = 8;                     // segments per lane and round; more: next round
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
                out[k + 1] = nco_out<kFast>(s.last, c);
            }
            finish_state<kFast>(s, c);
            store_state(seg + sg * 16, s);
            repaired++;
            atomicAnd(badmask + sg / 64, ~(1ull << (sg % 64)));
            if (sg + 1 < nseg) {
                // does the successor's lane stand on the repaired state?  (its outputs are finished values either way)
                const bool merged = pll_phase_dist(seg[(sg + 1) * 16 + 9], s.phase) <= tol_phase &&
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
        // remember where this call's phase started, for the next call's extrapolation (pll_start = 0)
        hdr[1] = 0.0f;                                         // for the next call's lanes to raise again
        hdr[5] = st[1];
        hdr[6] = static_cast<float>(n);
        hdr[7] = 1.0f;
        if (nothing_to_do) {                                   // the last segment's end state as fetched above
#pragma unroll
            for (int u = 0; u < 6; u++) state[u] = last[u];
        } else {
            PllState e = load_state(seg + (nseg - 1) * 16);
            store_state(state, e);
        }
    }
}

// ---- many channels: lane = channel, every lane the EXACT serial recurrence ---------------------------------------
// The recurrence cannot be cut in time without leaving the reference's trajectory, but channels are independent
// (one STATES set per receiver, src/project.cpp:455-468): a wave walks 64 channels' loops in lock step, the bank's
// rows are [channel][sample].  in: the pilot band-pass output; trig: the raw trigArg of every step (the NCO output
// cosf(trigArg*ncoScale + phaseAdjust) is off the chain: the bank's output kernel applies it); state: 8 floats per
// channel, the reference's six first; nco0[channel] = PLL[0] of this call = the incoming state's lastOut.
// Rows are 16-byte aligned with >= 16 readable floats behind their n samples (host contract): the input is fetched as
// 16-byte groups three groups ahead of the chain.
// IN8 (fast bank): the input is the SIGN of the pilot band-pass output, one signed byte per sample (+1 / -1; 0 = a sample that
// is zero, denormal-small or not finite) -- all the fast recurrence reads of an ordinary sample is its sign (pll_step_clean), and a
// byte instead of a float per IF sample is 0.6 of the bank's 6 bytes of HBM traffic per input sample.
template <int MATH, bool FLAT, bool IN8 = false>
__global__ __launch_bounds__(64, 1) void pll_channels_kernel(const float *__restrict__ in, long pitch_in, long n, long n_ch,
                                                           float *__restrict__ trig, long pitch_trig, float *__restrict__ state,
                                                           float *__restrict__ nco0, PllCoef c)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    // samples per lane and batch.  Exact: 32 (one 128-byte line of a float row; 14 us of chain).  Fast: 64 -- a batch must outlast the
    // memory latency UNDER LOAD (the lanes run next to kernels that saturate HBM: a request then takes ~5 us, and 32 fast steps are
    // 2.7 us: the recurrence ran 2.7 x slower next to the bank's other kernels than alone)
    constexpr int kBatch = MATH == kFast ? 64 : 32;
    __shared__ uint32_t w24[24];
    __shared__ float lin[kBatch * 64], lout[kBatch * 64];      // [sample][lane]: every lane reads and writes its own column
    if (threadIdx.x < 24) w24[threadIdx.x] = glibc235::inv_pio4(threadIdx.x);
    __syncthreads();
    // These waves are a dependent chain each and run next to wide kernels that fill every other issue slot of their SIMD: the
    // highest wave priority gives the chain every slot it can use (it can use few).
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x;
    const long ch = static_cast<long>(blockIdx.x) * 64 + threadIdx.x;
    if (ch >= n_ch) return;
    float *st = state + 8 * ch;
    PllState s{st[0], st[1], st[2], st[3], st[4], st[5], 0.0f};
    if (MATH == kFast) s.fr = atan2f(st[3], st[2]) * 0.15915494309189533577f;
    if (nco0) nco0[ch] = s.last;
    const f4 *in4 = reinterpret_cast<const f4 *>(in + ch * pitch_in);
    const int8_t *in8 = reinterpret_cast<const int8_t *>(in) + ch * pitch_in;    // IN8: pitch in bytes
    f4 *out4 = reinterpret_cast<f4 *>(trig + ch * pitch_trig);
    typedef int i4v __attribute__((ext_vector_type(4)));
    // four steps of the recurrence: the samples in, the raw trigArg of each step out
    auto four = [&](const f4 cur) __attribute__((always_inline)) -> f4 {   // (called twice: outlined, the state would live in scratch)
        f4 r;
        if (MATH == kFast) {
            // the specialised path's step (closed-form phase detector, one double argument reduction); one wave-uniform test
            // per group sends the whole wave through the general fast step when any lane meets a zero / non-finite sample
            if (__builtin_expect(__any(!(pll_ordinary(cur.x) && pll_ordinary(cur.y) && pll_ordinary(cur.z) && pll_ordinary(cur.w))), 0)) {
                pll_step<kFast>(s, cur.x, c); r.x = s.last;
                pll_step<kFast>(s, cur.y, c); r.y = s.last;
                pll_step<kFast>(s, cur.z, c); r.z = s.last;
                pll_step<kFast>(s, cur.w, c); r.w = s.last;
            } else {
                pll_step_clean(s, cur.x, c); r.x = s.last;
                pll_step_clean(s, cur.y, c); r.y = s.last;
                pll_step_clean(s, cur.z, c); r.z = s.last;
                pll_step_clean(s, cur.w, c); r.w = s.last;
            }
        } else if (MATH == kExact && FLAT) {
            pll_step_exact_flat(s, cur.x, c, w24); r.x = s.last;
            pll_step_exact_flat(s, cur.y, c, w24); r.y = s.last;
            pll_step_exact_flat(s, cur.z, c, w24); r.z = s.last;
            pll_step_exact_flat(s, cur.w, c, w24); r.w = s.last;
        } else {
            pll_step<MATH>(s, cur.x, c); r.x = s.last;
            pll_step<MATH>(s, cur.y, c); r.y = s.last;
            pll_step<MATH>(s, cur.z, c); r.z = s.last;
            pll_step<MATH>(s, cur.w, c); r.w = s.last;
        }
        return r;
    };
    // Whole batches.  The recurrence must never wait for memory: a batch's input is requested a whole batch (3-15 us of chain)
    // before it is used, passes through registers into LDS, and the inner loop touches LDS only; the batch's output leaves
    // from LDS behind the next requests.  (The compiler waits for ALL outstanding global accesses where it needs one of them:
    // with loads and stores inside the loop of steps every group of four steps paid a memory round trip.)
    const long nb = n / kBatch;
    f4 pre[IN8 ? 1 : kBatch / 4];
    i4v pre8[kBatch / 16];                                     // IN8: kBatch signed bytes
    if (nb > 0) {
        if (IN8) {
#pragma unroll
            for (int g = 0; g < kBatch / 16; g++) pre8[g] = reinterpret_cast<const i4v *>(in8)[g];
        } else {
#pragma unroll
            for (int g = 0; g < kBatch / 4; g++) pre[g] = in4[g];
        }
    }
    for (long b = 0; b < nb; b++) {
        // (1) this batch's input -> LDS (requested a batch ago); (2) request the next batch
        if (IN8) {
#pragma unroll
            for (int j = 0; j < kBatch; j++)
                lin[j * 64 + lane] = static_cast<float>(static_cast<int8_t>((pre8[j / 16][(j / 4) % 4] >> (8 * (j % 4))) & 0xff));
            if (b + 1 < nb) {
#pragma unroll
                for (int g = 0; g < kBatch / 16; g++) pre8[g] = reinterpret_cast<const i4v *>(in8 + (b + 1) * kBatch)[g];
            }
        } else {
#pragma unroll
            for (int g = 0; g < kBatch / 4; g++)
#pragma unroll
                for (int e = 0; e < 4; e++) lin[(4 * g + e) * 64 + lane] = pre[g][e];
            if (b + 1 < nb) {
#pragma unroll
                for (int g = 0; g < kBatch / 4; g++) pre[g] = in4[(b + 1) * (kBatch / 4) + g];
            }
        }
        // (3) the previous batch's output -> memory (its stores have this whole batch to complete)
        if (b > 0) {
#pragma unroll
            for (int g = 0; g < kBatch / 4; g++)
                out4[(b - 1) * (kBatch / 4) + g] = (f4){lout[(4 * g) * 64 + lane], lout[(4 * g + 1) * 64 + lane], lout[(4 * g + 2) * 64 + lane],
                                                        lout[(4 * g + 3) * 64 + lane]};
        }
        // (4) the chain: LDS in, LDS out
        f4 cur = (f4){lin[lane], lin[64 + lane], lin[128 + lane], lin[192 + lane]};
#pragma unroll 1
        for (int g = 0; g < kBatch / 4; g++) {
            const int gn = g + 1 < kBatch / 4 ? g + 1 : g;
            const f4 nxt = (f4){lin[(4 * gn) * 64 + lane], lin[(4 * gn + 1) * 64 + lane], lin[(4 * gn + 2) * 64 + lane],
                                lin[(4 * gn + 3) * 64 + lane]};
            const f4 r = four(cur);
            lout[(4 * g) * 64 + lane] = r.x;
            lout[(4 * g + 1) * 64 + lane] = r.y;
            lout[(4 * g + 2) * 64 + lane] = r.z;
            lout[(4 * g + 3) * 64 + lane] = r.w;
            cur = nxt;
        }
    }
    if (nb > 0) {
#pragma unroll
        for (int g = 0; g < kBatch / 4; g++)
            out4[(nb - 1) * (kBatch / 4) + g] = (f4){lout[(4 * g) * 64 + lane], lout[(4 * g + 1) * 64 + lane], lout[(4 * g + 2) * 64 + lane],
                                                     lout[(4 * g + 3) * 64 + lane]};
    }
    // what is left of a block that is not a multiple of the batch: groups of four, then single samples
    long k = nb * kBatch;
    for (; k + 4 <= n; k += 4)
        out4[k / 4] = four(IN8 ? (f4){static_cast<float>(in8[k]), static_cast<float>(in8[k + 1]), static_cast<float>(in8[k + 2]),
                                      static_cast<float>(in8[k + 3])}
                               : in4[k / 4]);
    for (; k < n; k++) {
        pll_step<MATH>(s, IN8 ? static_cast<float>(in8[k]) : in[ch * pitch_in + k], c);
        trig[ch * pitch_trig + k] = s.last;
    }
    finish_state<MATH>(s, c);
    store_state(st, s);
}

__global__ void libm_eval_kernel(int fn, const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                 float *__restrict__ out)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn <= 2) {
        out[i] = fn == 0 ? glibc235::sinf_glibc(a[i]) : fn == 1 ? glibc235::cosf_glibc(a[i]) : glibc235::atan2f_glibc(a[i], b[i]);
        return;
    }
    // 3, 4, 5: the branch-free variants where they are defined (the general function elsewhere: what a wave of the bank's PLL does)
    uint32_t w24[24];
    glibc235::inv_pio4_table(w24);
    if (fn == 5) {
        out[i] = glibc235::atan2f_flat_ok(a[i], b[i]) ? glibc235::atan2f_flat(a[i], b[i]) : glibc235::atan2f_glibc(a[i], b[i]);
    } else {
        float sn, cs;
        if (glibc235::sincosf_mid_ok(a[i])) glibc235::sincosf_mid_flat(a[i], &sn, &cs);
        else if (glibc235::sincosf_large_ok(a[i])) glibc235::sincosf_large_flat(a[i], w24, &sn, &cs);
        else glibc235::sincosf_glibc(a[i], &sn, &cs);
        out[i] = fn == 3 ? sn : cs;
    }
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
    // s' = A s + B x in (phase, integrator) / 2 pi: Q = A^64 and the response G of 64 steps to x = 1 (pll_lti_chunks_kernel)
    const double Kp = c.Kp, Ki = c.Ki, a00 = 1.0 - Kp - Ki, a01 = 1.0, a10 = -Ki, a11 = 1.0, b0 = Kp + Ki, b1 = Ki;
    for (int j = 0; j < kLtiChunk; j++) {
        const double g0 = a00 * c.g0 + a01 * c.g1 + b0, g1 = a10 * c.g0 + a11 * c.g1 + b1;
        c.g0 = g0; c.g1 = g1;
        const double n00 = a00 * c.q00 + a01 * c.q10, n01 = a00 * c.q01 + a01 * c.q11;
        const double n10 = a10 * c.q00 + a11 * c.q10, n11 = a10 * c.q01 + a11 * c.q11;
        c.q00 = n00; c.q01 = n01; c.q10 = n10; c.q11 = n11;
    }
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
    const unsigned grid = static_cast<unsigned>((n + 255) / 256);
    if (fast) {
        hipLaunchKernelGGL(pll_serial_kernel<kFast>, dim3(1), dim3(64), 0, s, d_in, n, d_out, d_state, c);
        FMRX_LAUNCH_CHECK("pll_serial");
        if (n) hipLaunchKernelGGL(nco_out_kernel<kFast>, dim3(grid), dim3(256), 0, s, d_out, n, c);
    } else {
        hipLaunchKernelGGL(pll_serial_kernel<kExact>, dim3(1), dim3(64), 0, s, d_in, n, d_out, d_state, c);
        FMRX_LAUNCH_CHECK("pll_serial");
        if (n) hipLaunchKernelGGL(nco_out_kernel<kExact>, dim3(grid), dim3(256), 0, s, d_out, n, c);
    }
    FMRX_LAUNCH_CHECK("nco_out");
    return FMRX_OK;
}

int k_fm_pll_channels(const float *d_in, long pitch_in, size_t n, int n_ch, float *d_trig, long pitch_trig, float *d_state,
                      float *d_nco0, float freq, float Fs, float ncoScale, float phaseAdjust, float normBandwidth, hipStream_t s, bool flat,
                      bool exact, bool in8)
{
    if (n == 0 || n_ch <= 0) return FMRX_OK;
    if (reinterpret_cast<uintptr_t>(d_in) % 16 || reinterpret_cast<uintptr_t>(d_trig) % 16 || pitch_in % (in8 ? 16 : 4) || pitch_trig % 4 ||
        pitch_in < static_cast<long>(n) + 16 || (in8 && exact))
        return fail(FMRX_EINVAL, "fm_pll_channels: rows must be 16-byte aligned with 16 readable samples behind their own");
    const PllCoef c = make_coef(freq, Fs, ncoScale, phaseAdjust, normBandwidth);
    if (in8)
        hipLaunchKernelGGL((pll_channels_kernel<kFast, false, true>), dim3(static_cast<unsigned>((n_ch + 63) / 64)), dim3(64), 0, s, d_in,
                           pitch_in, static_cast<long>(n), static_cast<long>(n_ch), d_trig, pitch_trig, d_state, d_nco0, c);
    else if (!exact)
        hipLaunchKernelGGL((pll_channels_kernel<kFast, false>), dim3(static_cast<unsigned>((n_ch + 63) / 64)), dim3(64), 0, s, d_in, pitch_in,
                           static_cast<long>(n), static_cast<long>(n_ch), d_trig, pitch_trig, d_state, d_nco0, c);
    else if (flat)
        hipLaunchKernelGGL((pll_channels_kernel<kExact, true>), dim3(static_cast<unsigned>((n_ch + 63) / 64)), dim3(64), 0, s, d_in, pitch_in,
                           static_cast<long>(n), static_cast<long>(n_ch), d_trig, pitch_trig, d_state, d_nco0, c);
    else
        hipLaunchKernelGGL((pll_channels_kernel<kExact, false>), dim3(static_cast<unsigned>((n_ch + 63) / 64)), dim3(64), 0, s, d_in, pitch_in,
                           static_cast<long>(n), static_cast<long>(n_ch), d_trig, pitch_trig, d_state, d_nco0, c);
    FMRX_LAUNCH_CHECK("pll_channels");
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
    const size_t nseg = n / kPllSegmentMin + 2;
    const size_t nchunk = n / kLtiChunk + 2;
    return 8 + nseg * 16 + 2 * (nseg / 64 + 2) + 2 + (nchunk + 2) * 8 + 2 * (nchunk / 64 + 2);   // + chunk records (4 doubles), workgroup totals
}

size_t pll_parallel_lti_floats(size_t n)
{
    const size_t nchunk = n / kLtiChunk + 2;
    return (nchunk + 2) * 8 + 2 * (nchunk / 64 + 2) + 2;
}

int k_fm_pll_parallel(const float *d_in, size_t n, float *d_out, float *d_state, float freq, float Fs, float ncoScale,
                      float phaseAdjust, float normBandwidth, float *d_scratch, const Options &o, hipStream_t s, double off_hint,
                      int phases, float *d_lti)
{
    int L = kPllSegment, W = kPllWarmup;
    if (o.pll_warmup >= 0 && o.pll_warmup <= 65536) W = o.pll_warmup / 4 * 4;                    // tuning: warm-up samples per lane
    if (o.pll_segment >= kPllSegmentMin && o.pll_segment <= 65536) L = o.pll_segment / 4 * 4;   // tuning: samples per lane
    // The linear-system start holds while the float32 grid of trigArg is a perturbation of the loop (ulp <= 0.25 rad: the
    // first 2^22 rad = 8.4 M IF samples = 35 s of a stream, measured to 27.7 s); beyond that the grid IS the loop's
    // dynamics and only true steps reproduce it: the lanes then start the first way.  off_hint = the caller's count of the
    // stream's IF samples in front of this call (< 0: unknown).
    const double w = 2 * 3.14159265358979323846 * static_cast<double>(freq / Fs);
    const bool lti = o.pll_start == 1 && off_hint >= 0.0 && w * (off_hint + static_cast<double>(n)) < 4194304.0;
    if (lti) {
        // lanes start from the linear system's state (see pll_lti_chunks_kernel): W, L in whole 64-sample chunks
        if (o.pll_warmup < 0) W = kPllWarmupLti;
        W = (W + kLtiChunk - 1) / kLtiChunk * kLtiChunk;
        L = kLtiChunk;                                             // one lane per chunk (a wave's 64 lanes then start inside two chunk workgroups)
    }
    if (reinterpret_cast<uintptr_t>(d_in) % 16)
        return fail(FMRX_EINVAL, "fm_pll_parallel: input must be 16-byte aligned (the lanes fetch 16-byte groups)");
    if (d_lti && reinterpret_cast<uintptr_t>(d_lti) % 8) return fail(FMRX_EINVAL, "fm_pll_parallel: chunk records must be 8-byte aligned");
    if (n < static_cast<size_t>(4 * L)) {   // nothing to gain
        if (!(phases & 2)) return FMRX_OK;
        return k_fm_pll(d_in, n, d_out, d_state, freq, Fs, ncoScale, phaseAdjust, normBandwidth, 1, s);
    }
    PllCoef c = make_coef(freq, Fs, ncoScale, phaseAdjust, normBandwidth);
    // the loop's state repeats every P samples on an on-frequency pilot: Fs / gcd(Fs, freq), when both are whole Hz
    int P = 0;
    if (o.pll_align != 0 && !lti && Fs == static_cast<float>(static_cast<long>(Fs)) && freq == static_cast<float>(static_cast<long>(freq)) && freq > 0) {
        long x = static_cast<long>(Fs), y = static_cast<long>(freq);
        while (y) { const long t = x % y; x = y; y = t; }
        const long p = static_cast<long>(Fs) / x;
        if (p > 0 && p <= 4096 && p % 4 == 0) P = static_cast<int>(p);
    }
    const long nseg = static_cast<long>((n + L - 1) / L);
    // scratch: [2] repaired-segment counter (u32), [3],[4] largest accepted |dphase|,|dinteg| (diagnostics),
    // [5..7] previous call's start phase / length / valid; [8..] per-segment records, then the mismatch bitmask
    unsigned *n_repaired = reinterpret_cast<unsigned *>(d_scratch + 2);
    float *seg = d_scratch + 8;
    unsigned long long *badmask = reinterpret_cast<unsigned long long *>(seg + (nseg + 1) * 16);
    const unsigned grid = static_cast<unsigned>((nseg + kSegThreads - 1) / kSegThreads);
    const double *lti_rec = nullptr, *lti_wgtot = nullptr;
    if (lti) {
        c.integ_tol_ulps = kPllIntegTolUlpsLti;
        const long nchunk = static_cast<long>(n / kLtiChunk) + 1;
        const size_t mask_floats = 2 * (static_cast<size_t>(nseg) / 64 + 2);
        size_t off = 8 + (static_cast<size_t>(nseg) + 1) * 16 + mask_floats;
        off += off & 1;                                            // doubles
        double *rec = reinterpret_cast<double *>(d_lti ? d_lti : d_scratch + off);
        double *wgtot = rec + 4 * (nchunk + 1);
        if (phases & 1) {
            hipLaunchKernelGGL(pll_lti_chunks_kernel, dim3(static_cast<unsigned>((nchunk + 63) / 64)), dim3(64), 0, s, d_in,
                               static_cast<long>(n), c, nchunk, rec, wgtot);
            FMRX_LAUNCH_CHECK("pll_lti_chunks");
        }
        lti_rec = rec;
        lti_wgtot = wgtot;
    }
    if (!(phases & 2)) return FMRX_OK;
    hipLaunchKernelGGL(pll_segments_kernel, dim3(grid), dim3(kSegThreads), 0, s, d_in, static_cast<long>(n), d_out, d_state, c, L, W, P,
                       nseg, seg, d_scratch, lti_rec, lti_wgtot, badmask, kPllTolPhase, kPllTolInteg);
    FMRX_LAUNCH_CHECK("pll_segments");
    hipLaunchKernelGGL(pll_repair_kernel, dim3(1), dim3(kRepairThreads), 0, s, d_in, static_cast<long>(n), d_out, d_state, c, L, nseg, seg,
                       badmask, kPllTolPhase, kPllTolInteg, n_repaired, d_scratch);
    FMRX_LAUNCH_CHECK("pll_repair");
    return FMRX_OK;
}

}  // namespace fmrx

// device_math.hpp -- small __device__ helpers shared by the kernel files, so
// that a value computed by two different kernels (e.g. a discriminator sample
// computed inside the fused audio kernel and again by the history-tail kernel)
// is the same bit pattern.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Plain operators below must stay separate IEEE operations: no a*b+c -> fma.
// (The __fmul_rn/__fadd_rn intrinsics are not enough: they inline OCML bitcode
// whose multiplies carry the `contract` flag and get fused anyway.)  Files that
// want FMAs ask for them explicitly (__builtin_elementwise_fma / fmaf).
#pragma clang fp contract(off)

namespace fmrx {

// FM discriminator, reference evaluation order (src/filter.cpp:254-259):
// separate rounded products, one IEEE divide.  Compile the including file
// with contraction off for this to stay bit-compatible.
__device__ __forceinline__ float demod_exact(float i, float q, float pi, float pq)
{
    const float ii = i * i, qq = q * q;
    const float den = ii + qq;
    if (den == 0.0f) return 0.0f;
    const float a = i * (q - pq);
    const float b = q * (i - pi);
    return (a - b) / den;  // IEEE divide
}

// Throughput form used by the specialised pipeline: the reference's numerator
// and denominator (separately rounded products -- so degenerate cases such as
// I*Q - Q*I stay exactly 0, which the stereo PLL's atan2 is sensitive to), but
// a 1-ulp hardware reciprocal (v_rcp_f32) instead of the ~10-instruction IEEE
// divide: |error| <= ~1.5 ulp of the quotient.  Tiny denominators are scaled
// by 2^64 first (numerator too: same quotient) so v_rcp_f32 never sees a
// denormal or overflows; branch-free.
__device__ __forceinline__ float demod_fast(float i, float q, float pi, float pq)
{
    const float ii = i * i, qq = q * q;
    const float den = ii + qq;
    const float a = i * (q - pq);
    const float b = q * (i - pi);
    const float num = a - b;
    const float sc = den < 8.6736174e-19f ? 1.8446744e19f : 1.0f;   // den < 2^-60 ? 2^64 : 1
    const float quot = (num * sc) * __builtin_amdgcn_rcpf(den * sc);
    return den == 0.0f ? 0.0f : quot;                                // select, not a branch: every lane runs both anyway
}

// The same for operands known to be 0 or of ordinary size (the matrix-core kernels' straight-line tiles: an IF sample is an
// integer times the plan's scale 2^-(s+7), and those tiles run only when that scale is >= 2^-50): no denominator is tiny, so
// the 2^64 scaling of demod_fast (a compare, a select and two multiplies per output) selects 1 every time and is left out.
// Bit-identical to demod_fast on those operands.
__device__ __forceinline__ float demod_fast_bounded(float i, float q, float pi, float pq)
{
    const float den = i * i + q * q;
    const float num = i * (q - pq) - q * (i - pi);
    const float quot = num * __builtin_amdgcn_rcpf(den);
    return den == 0.0f ? 0.0f : quot;
}

// PCM pack of src/threadMonoOnly.cpp:185-191: NaN -> 0 else (short)(a*16384).
// wrap: what the compiled reference does out of range (cvttss2si, low 16 bits).
__device__ __forceinline__ int16_t pcm_pack(float a, int wrap)
{
    if (a != a) return 0;
    const float s = a * 16384.0f;
    if (wrap) {
        int v;
        if (s >= 2147483648.0f || s < -2147483648.0f) v = static_cast<int>(0x80000000u);
        else v = static_cast<int>(s);
        return static_cast<int16_t>(static_cast<uint16_t>(static_cast<uint32_t>(v)));
    }
    if (s >= 32767.0f) return 32767;
    if (s <= -32768.0f) return -32768;
    return static_cast<int16_t>(s);
}

// the same value without a branch (straight-line epilogues): out of range or NaN -> 0 either way under wrap (the low 16 bits of
// cvttss2si's 0x80000000); v_cvt_i32_f32 itself saturates, so it is only used in range
__device__ __forceinline__ int16_t pcm_pack_flat(float a, int wrap)
{
    const float s = a * 16384.0f;
    const float c = fminf(fmaxf(s, -32768.0f), 32767.0f);                    // saturating form (NaN -> -32768 here, 0 below)
    const bool keep = wrap ? (s < 2147483648.0f && s >= -2147483648.0f) : a == a;   // false for NaN in both forms
    const float arg = wrap ? s : c;
    int v;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(v) : "v"(arg));
    return static_cast<int16_t>(static_cast<uint16_t>(keep ? static_cast<uint32_t>(v) : 0u));
}

}  // namespace fmrx

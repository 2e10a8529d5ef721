// kernels_diag.hip -- measurement aids, not part of the path: what this box's memory system gives a pure
// streaming READ by the two access methods the front-end kernels use (bench.py reports it next to the 8 TB/s
// peak so that a kernel's fraction of the roofline can be read against what a read-only kernel reaches):
//   method 0: global_load_dwordx4, non-temporal, 4 loads in flight per wave, into registers
//   method 1: LDS-DMA (global_load_lds_dwordx4) into a wave-private 4-slot ring, counted vmcnt, read back
//   method m >= 2: as 1, chunks of m - 1 steps of 3 KiB dealt round-robin over the waves instead of one contiguous run per wave
// Persistent grid, every wave streams its own contiguous run; the data is XOR-reduced so nothing is dropped.
#include "fmrx_internal.hpp"

namespace fmrx {
namespace {

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read_regs_kernel(const u4 *__restrict__ x, long n16, unsigned *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long waves = static_cast<long>(gridDim.x) * 4, w = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const long pieces = n16 / 64;                                       // 1 KiB pieces
    const long per = (pieces + waves - 1) / waves;
    const long p0 = w * per, p1 = p0 + per < pieces ? p0 + per : pieces;
    u4 acc = {0, 0, 0, 0};
    for (long p = p0; p < p1; p += 4) {
        u4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const long q = p + i < p1 ? p + i : p1 - 1;
            v[i] = __builtin_nontemporal_load(x + q * 64 + lane);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;   // keeps the loads alive
}

// SWEEP: steps are dealt round-robin over the waves (the chip reads one contiguous stretch at any moment) instead of every
// wave owning one contiguous run
__global__ __launch_bounds__(256) void read_dma_kernel(const unsigned char *__restrict__ x, long n_bytes, unsigned *__restrict__ out, int SWEEP)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *ring = lds + wave * 4 * 3072;
    const long waves = static_cast<long>(gridDim.x) * 4, w = static_cast<long>(blockIdx.x) * 4 + wave;
    const long steps = n_bytes / 3072, per = (steps + waves - 1) / waves;
    const long s0 = w * per, s1 = s0 + per < steps ? s0 + per : steps;
    auto issue = [&](long s, int slot) {
        if (SWEEP) {   // chunks of SWEEP steps dealt round-robin over the waves
            const long i = s - s0, c = i / SWEEP, r = i - c * SWEEP;
            const long t = (c * waves + w) * SWEEP + r;
            s = t < steps ? t : steps - 1;
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + s * 3072 + k * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(ring + slot * 3072 + k * 1024), 16, 0, 0);
    };
    u4 acc = {0, 0, 0, 0};
    for (int i = 0; i < 3 && s0 + i < s1; i++) issue(s0 + i, i);
    int slot = 0, fill = 3;
    for (long s = s0; s < s1; s++) {
        const bool steady = s + 3 < s1;
        if (steady) issue(s + 3, fill);
        if (steady) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u4 *src = reinterpret_cast<const u4 *>(ring + slot * 3072);
        acc ^= src[lane] ^ src[64 + lane] ^ src[128 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot = (slot + 1) & 3;
        fill = (fill + 1) & 3;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

}  // namespace

int k_stream_read(const void *d_buf, size_t bytes, int method, unsigned *d_sink, hipStream_t s)
{
    if (bytes < 3072 * 1024) return fail(FMRX_EINVAL, "stream_read: buffer too small to mean anything");
    if (reinterpret_cast<uintptr_t>(d_buf) % 16) return fail(FMRX_EINVAL, "stream_read: buffer must be 16-byte aligned");
    if (method == 0)
        hipLaunchKernelGGL(read_regs_kernel, dim3(1024), dim3(256), 0, s, static_cast<const u4 *>(d_buf), static_cast<long>(bytes / 16), d_sink);
    else if (method == 1)
        hipLaunchKernelGGL(read_dma_kernel, dim3(512), dim3(256), 4 * 4 * 3072, s, static_cast<const unsigned char *>(d_buf),
                           static_cast<long>(bytes - bytes % 3072), d_sink, 0);
    else   // method m >= 2: chunks of m - 1 steps
        hipLaunchKernelGGL(read_dma_kernel, dim3(512), dim3(256), 4 * 4 * 3072, s, static_cast<const unsigned char *>(d_buf),
                           static_cast<long>(bytes - bytes % 3072), d_sink, method - 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FMRX_EHIP, "launch stream_read: %s", hipGetErrorString(e));
    return FMRX_OK;
}

}  // namespace fmrx

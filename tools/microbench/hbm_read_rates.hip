// hbm_read_rates.hip -- what does this box's memory system give a pure streaming READ, by access method?
// (the question behind kernels_fe_mfma.hip: is LDS-DMA the ceiling of the front end's 5.5 TB/s?)
//   hipcc --offload-arch=gfx950 -O3 hbm_read_rates.hip -o hbm_read_rates && ./hbm_read_rates
// Variants, all persistent grids of 256-thread workgroups, each wave streaming its own contiguous run:
//   0: global_load_dwordx4 into registers, 4 loads in flight per wave, XOR-reduced (so nothing is dropped)
//   1: same, 8 loads in flight
//   2: LDS-DMA (global_load_lds_dwordx4), 3 x 1 KiB pieces per step into a 4-slot ring, counted vmcnt, ds_read back
//   3: variant 0 with non-temporal loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int INFLIGHT, bool NT>
__global__ __launch_bounds__(256) void read_regs(const u4 *__restrict__ x, long n16, unsigned *out)
{
    const int lane = threadIdx.x & 63;
    const long waves = static_cast<long>(gridDim.x) * 4, w = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const long per = (n16 / 64 + waves - 1) / waves;                  // 1 KiB pieces per wave
    const long p0 = w * per, p1 = p0 + per < n16 / 64 ? p0 + per : n16 / 64;
    u4 acc = {0, 0, 0, 0};
    for (long p = p0; p < p1; p += INFLIGHT) {
        u4 v[INFLIGHT];
#pragma unroll
        for (int i = 0; i < INFLIGHT; i++) {
            const long q = p + i < p1 ? p + i : p1 - 1;
            v[i] = NT ? __builtin_nontemporal_load(x + q * 64 + lane) : x[q * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < INFLIGHT; i++) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;   // keeps the loads alive
}

__global__ __launch_bounds__(256) void read_dma(const unsigned char *__restrict__ x, long n_bytes, unsigned *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *ring = lds + wave * 4 * 3072;
    const long waves = static_cast<long>(gridDim.x) * 4, w = static_cast<long>(blockIdx.x) * 4 + wave;
    const long steps = n_bytes / 3072, per = (steps + waves - 1) / waves;
    const long s0 = w * per, s1 = s0 + per < steps ? s0 + per : steps;
    auto issue = [&](long s, int slot) {
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

int main()
{
    const long n_bytes = 512L * 1024 * 1024 - (512L * 1024 * 1024) % 3072;
    unsigned char *d; unsigned *o;
    CHECK(hipMalloc(&d, n_bytes)); CHECK(hipMalloc(&o, 4));
    std::vector<unsigned char> h(n_bytes);
    for (long i = 0; i < n_bytes; i++) h[i] = static_cast<unsigned char>(rand());
    CHECK(hipMemcpy(d, h.data(), n_bytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wgs : {256, 512, 1024}) {
        for (int v = 0; v < 4; v++) {
            auto launch = [&]() {
                if (v == 0) hipLaunchKernelGGL((read_regs<4, false>), dim3(wgs), dim3(256), 0, 0, (const u4 *)d, n_bytes / 16, o);
                if (v == 1) hipLaunchKernelGGL((read_regs<8, false>), dim3(wgs), dim3(256), 0, 0, (const u4 *)d, n_bytes / 16, o);
                if (v == 2) hipLaunchKernelGGL(read_dma, dim3(wgs), dim3(256), 4 * 4 * 3072, 0, d, n_bytes, o);
                if (v == 3) hipLaunchKernelGGL((read_regs<4, true>), dim3(wgs), dim3(256), 0, 0, (const u4 *)d, n_bytes / 16, o);
            };
            for (int i = 0; i < 200; i++) launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 100; i++) launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("variant %d (%s), %4d workgroups: %.4f ms per 512 MiB = %.0f GB/s\n", v,
                   v == 0 ? "regs x4" : v == 1 ? "regs x8" : v == 2 ? "LDS-DMA ring" : "regs x4 nt", wgs, ms / 100, n_bytes / (ms / 100 * 1e-3) / 1e9);
        }
    }
    return 0;
}

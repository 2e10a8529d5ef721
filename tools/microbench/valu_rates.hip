// valu_rates.hip -- VALU issue-rate microbenchmark for gfx950 (MI355X).
// Answers the design questions behind kernels_fe.hip: what do v_fma_f32,
// v_pk_fma_f32 (SGPR tap operand, op_sel broadcast), v_cvt_f32_ubyteN and
// v_fma_mix_f32 cost per wave-instruction, alone and mixed 808:342 like the
// front-end kernel, at 1..8 waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NACC = 8;
constexpr int UNROLL = 8;   // ops per acc per loop iteration

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float hs, long long *cyc)
{
    f2 acc[NACC];
    f2 x = {(float)threadIdx.x, (float)(threadIdx.x + 1)};
    unsigned raw = threadIdx.x * 0x01020304u;
    float tmp[4] = {0, 0, 0, 0};
    for (int i = 0; i < NACC; i++) acc[i] = (f2){0.f, (float)i};
    f2 h2 = {hs, hs * 0.5f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "s"(hs));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(x), "s"(h2));
                if (OP == 2) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(tmp[i & 3]) : "v"(raw));
                if (OP == 3) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i].x) : "v"(raw), "s"(hs));
                if (OP == 4) {  // the front-end mix: 8 pk_fma then ~3.4 cvt
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(x), "s"(h2));
                    if (i < 3 || (i == 3 && (u & 1))) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(tmp[i & 3]) : "v"(raw));
                }
                if (OP == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(h2));  // VGPR tap
                if (OP == 6) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(x), "v"(h2));
                if (OP == 7) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(tmp[i & 3]) : "v"(raw));
                if (OP == 8) {  // front-end mix with the signed (SDWA) conversion
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(x), "s"(h2));
                    if (i < 3 || (i == 3 && (u & 1))) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(tmp[i & 3]) : "v"(raw));
                }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = tmp[0] + tmp[1] + tmp[2] + tmp[3];
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
int run(const char *name, int waves_per_simd, double ops_per_inner)
{
    const int cus = 256, iters = 2000;
    const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = 1 wave per SIMD per block
    float *out; long long *cyc;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    CHECK(hipMalloc(&cyc, sizeof(long long) * blocks));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    const double inner = (double)iters * UNROLL * NACC;       // inner bodies per wave
    const double instr = inner * ops_per_inner;               // VALU instr per wave
    // s_memtime ticks at 100 MHz on gfx9 "REFCLK"? report both raw ticks and wall-derived
    const double wave_instr_total = instr * 4.0 * blocks;     // all waves
    const double per_simd_instr = wave_instr_total / (cus * 4.0);
    const double ns_per_instr_simd = ms * 1e6 / per_simd_instr;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f clk @2.4GHz)  memtime/instr=%.3f\n",
           name, waves_per_simd, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4, mean / instr);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
    return 0;
}

int main()
{
    for (int w : {1, 3, 8}) {
        if (run<0>("v_fma_f32 (sgpr src)", w, 1)) return 1;
        if (run<1>("v_pk_fma_f32 (sgpr pair, op_sel)", w, 1)) return 1;
        if (run<5>("v_pk_fma_f32 (vgpr tap)", w, 1)) return 1;
        if (run<6>("v_pk_mul_f32", w, 1)) return 1;
        if (run<2>("v_cvt_f32_ubyte1", w, 1)) return 1;
        if (run<3>("v_fma_mix_f32", w, 1)) return 1;
        if (run<4>("mix 8 pk_fma : 3.5 cvt", w, 1.0 + 3.5 / 8.0)) return 1;
        if (run<7>("v_cvt_f32_i32_sdwa sext byte", w, 1)) return 1;
        if (run<8>("mix 8 pk_fma : 3.5 cvt_sdwa", w, 1.0 + 3.5 / 8.0)) return 1;
        printf("\n");
    }
    return 0;
}

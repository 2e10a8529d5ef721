#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// taps laid out phase-major: hp[p*QP + q] = h'[T-1-(p+D*q)] (0 beyond), QP = ceil(T/D) padded to 12
template<int T, int D, int R>
__global__ __launch_bounds__(256) void fe_proto(const uint8_t* __restrict__ x, const float* __restrict__ hp, f2* __restrict__ y, int nout, float c0)
{
    constexpr int W = D*(R-1)+T;          // window samples
    constexpr int LEAD = 4;               // alignment lead samples (8 bytes)
    constexpr int NB = (2*(W+LEAD)+15)/16; // b128 reads
    constexpr int QT = (T + D - 1)/D;      // taps per phase (max)
    constexpr int QP = 12;
    extern __shared__ uint8_t lds[];
    const int t = threadIdx.x;
    const u4* lw = reinterpret_cast<const u4*>(lds + t*(2*D*R));
    uint32_t raw[NB*4];
#pragma unroll
    for (int i=0;i<NB;i++){ u4 v = lw[i]; raw[4*i]=v.x; raw[4*i+1]=v.y; raw[4*i+2]=v.z; raw[4*i+3]=v.w; }
    f2 acc[R];
#pragma unroll
    for (int r=0;r<R;r++) acc[r] = (f2){c0, c0};
#pragma unroll
    for (int p=0;p<D;p++){
        __builtin_amdgcn_sched_barrier(0);
        f8 ha; f4 hb;
        asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ha), "=&s"(hb) : "s"(hp), "i"(p*QP*4), "i"(p*QP*4+32));
        float hq[12] = {ha[0],ha[1],ha[2],ha[3],ha[4],ha[5],ha[6],ha[7],hb[0],hb[1],hb[2],hb[3]};
        // phase p: window samples j = p + D*i, i = 0..R-1+QT-1 ; output r uses i = r + q with tap (p + D*q) -> hq[q]
#pragma unroll
        for (int i=0; i<R+QT-1; i++){
            const int j = p + D*i;
            if (j < W) {
            const int bo = 2*(j+LEAD);
            const uint32_t wI = raw[bo/4];
            f2 xs;
            if ((bo%4)==0){ xs.x = (float)(wI&0xffu); xs.y = (float)((wI>>8)&0xffu);} 
            else { xs.x = (float)((wI>>16)&0xffu); xs.y = (float)(wI>>24);} 
#pragma unroll
            for (int r=0;r<R;r++){
                const int q = i - r;
                if (q>=0 && q<QT && (p + D*q) < T){
                    const float h = hq[q];
                    acc[r] = __builtin_elementwise_fma(xs, (f2){h,h}, acc[r]);
                }
            }
            }
        }
#pragma unroll
        for (int r=0;r<R;r++) asm volatile("" : "+v"(acc[r]));
    }
    const int k0 = (blockIdx.x*256 + t)*R;
#pragma unroll
    for (int r=0;r<R;r++) if (k0+r<nout) y[k0+r]=acc[r];
}
template __global__ void fe_proto<101,10,8>(const uint8_t*, const float*, f2*, int, float);

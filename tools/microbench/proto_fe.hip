// Prototype: codegen check for the per-thread register-window FIR (T=101, D=10, R=8)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template<int T, int D, int R>
__global__ __launch_bounds__(256) void fe_proto(const uint8_t* __restrict__ x, const float* __restrict__ hp, f2* __restrict__ y, int nout)
{
    constexpr int W = D*(R-1)+T;          // window samples
    constexpr int LEAD = 4;               // alignment lead samples (8 bytes)
    constexpr int NB = (2*(W+LEAD)+15)/16; // b128 reads
    extern __shared__ uint8_t lds[];
    // (staging elided in proto: pretend lds filled)
    const int t = threadIdx.x;
    const u4* lw = reinterpret_cast<const u4*>(lds + t*(2*D*R));
    uint32_t raw[NB*4];
#pragma unroll
    for (int i=0;i<NB;i++){ u4 v = lw[i]; raw[4*i]=v.x; raw[4*i+1]=v.y; raw[4*i+2]=v.z; raw[4*i+3]=v.w; }
    f2 acc[R];
#pragma unroll
    for (int r=0;r<R;r++) acc[r] = (f2){hp[T+1], hp[T+1]};
    // window sample j (0..W-1) is at byte offset 2*(j+LEAD); output r uses sample j=D*r+m with tap h[T-1-m]
#pragma unroll
    for (int p=0;p<D;p++){
        // samples j ≡ p mod D: j = p + D*q, q=0..; 
#pragma unroll
        for (int q=0; p + D*q < W; q++){
            const int j = p + D*q;
            const int bo = 2*(j+LEAD);
            const uint32_t wI = raw[bo/4];
            f2 xs;
            if ((bo%4)==0){ xs.x = (float)(wI&0xffu); xs.y = (float)((wI>>8)&0xffu);} 
            else { xs.x = (float)((wI>>16)&0xffu); xs.y = (float)(wI>>24);} 
#pragma unroll
            for (int r=0;r<R;r++){
                const int m = j - D*r;
                if (m>=0 && m<T){
                    const float h = hp[T-1-m];
                    acc[r] = __builtin_elementwise_fma(xs, (f2){h,h}, acc[r]);
                }
            }
        }
    }
    const int k0 = (blockIdx.x*256 + t)*R;
#pragma unroll
    for (int r=0;r<R;r++) if (k0+r<nout) y[k0+r]=acc[r];
}
template __global__ void fe_proto<101,10,8>(const uint8_t*, const float*, f2*, int);

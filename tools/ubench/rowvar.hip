// micro-benchmark: what paces the bare recurrence row of 2 columns per lane in ONE wave -- issue slots or latencies?
// Variants of the row body, a lone wave each, shader clock cycles per row.
// build: hipcc --offload-arch=gfx950 -O3 -o rowvar rowvar.hip ; ./rowvar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ double shr_z(double v){ int lo=__builtin_amdgcn_update_dpp(0,__double2loint(v),0x138,0xf,0xf,true), hi=__builtin_amdgcn_update_dpp(0,__double2hiint(v),0x138,0xf,0xf,true); return __hiloint2double(hi,lo); }
__device__ __forceinline__ double shr_add(double v, int dsh){ int lo=__builtin_amdgcn_update_dpp(0,__double2loint(v),0x138,0xf,0xf,true), hi=__builtin_amdgcn_update_dpp(0,__double2hiint(v),0x138,0xf,0xf,true)+dsh; return __hiloint2double(hi,lo); }
template<int V, int UN=8> __global__ __launch_bounds__(64) void k(double* out, int rows, double a, int dsh){
  const int lane=threadIdx.x&63;
  double v0=1.0+lane*1e-3, v1=2.0+lane*1e-3, c0=a, c1=a+1e-3, s=1.0+1e-12;
  double w0=1.5+lane*1e-3, w1=2.5+lane*1e-3, d0=a, d1=a+1e-3;
  long long t0=clock64();
  for(int r=0;r<rows;r+=UN){
#pragma unroll
    for(int u=0;u<UN;u++){
      if constexpr (V==0) { const double t=shr_z(v1)*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1e-9; c1+=1e-9; }
      if constexpr (V==1) { const double t=shr_add(v1,dsh); v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1e-9; c1+=1e-9; }
      if constexpr (V==2) { const double t=shr_z(v1)*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); }
      if constexpr (V==3) { const double t=v1*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1e-9; c1+=1e-9; }
      if constexpr (V==4) { const double t=v1; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1e-9; c1+=1e-9; }
      if constexpr (V==5) { const double t=shr_z(v1)*s; const double tw=shr_z(w1)*s; v1=fma(c1,v1,v0); w1=fma(d1,w1,w0); v0=fma(c0,v0,t); w0=fma(d0,w0,tw);
                            c0+=1e-9; c1+=1e-9; d0+=1e-9; d1+=1e-9; }
      if constexpr (V==6) { const double t=v1; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); }
      if constexpr (V==7) { const double t=shr_add(v1,dsh); const double tw=shr_add(w1,dsh); v1=fma(c1,v1,v0); w1=fma(d1,w1,w0); v0=fma(c0,v0,t); w0=fma(d0,w0,tw);
                            c0+=1e-9; c1+=1e-9; d0+=1e-9; d1+=1e-9; }
      if constexpr (V==8) { const double t=shr_add(v1,dsh); v1=fma(c1,v1,v0); v0=fma(c0,v0,t); }
    }
  }
  long long t1=clock64();
  out[lane]=v0+v1+c0+c1+w0+w1+d0+d1;
  if(lane==0) out[64]=(double)(t1-t0);
}
template<int V, int UN=8> void run(double* d, const char* what){
  const int rows=19200;
  hipLaunchKernelGGL((k<V,UN>), dim3(1), dim3(64), 0, 0, d, 960, 0.5, 0);
  CHK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<V,UN>), dim3(1), dim3(64), 0, 0, d, rows, 0.5, 0);
  CHK(hipDeviceSynchronize());
  double h[65]; CHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("variant %d unroll %2d: %6.1f cycles a row   %s\n", V, UN, h[64]/rows, what);
}
int main(){
  double* d; CHK(hipMalloc(&d, 4096));
  run<0>(d, "the row as it is: 2 dpp moves, multiply, 2 fma, 2 adds");
  run<1>(d, "exponent add in the dpp move instead of the multiply (6 instructions)");
  run<2>(d, "as it is without the coefficient adds (5)");
  run<3>(d, "no dpp: own lane's value, multiply, 2 fma, 2 adds (5)");
  run<4>(d, "no dpp, no multiply (4)");
  run<6>(d, "only the two fma (2)");
  run<8>(d, "dpp move + dpp add, 2 fma, no adds (4)");
  run<0,16>(d, "as it is");
  run<0,24>(d, "as it is");
  run<0,48>(d, "as it is");
  run<0,4>(d, "as it is");
  run<5>(d, "TWO independent rows interleaved, as they are (14)");
  run<7>(d, "TWO independent rows interleaved, exponent add (12)");
  return 0;
}

// micro-benchmark: sustained VALU rate with the chip full (fp32 and fp64 fma), to calibrate the
// per-instruction costs quoted in DESIGN.md against the clock the part actually runs at
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<typename T> __global__ __launch_bounds__(256) void k(T* out, int iters, T a, T b){
  T x[8]; for(int i=0;i<8;i++) x[i]=(T)(threadIdx.x+i);
  for(int it=0; it<iters; it++){
    #pragma unroll
    for(int r=0;r<4;r++)
      #pragma unroll
      for(int i=0;i<8;i++) x[i]=fma(x[i],a,b);
  }
  T s=0; for(int i=0;i<8;i++) s+=x[i];
  out[(size_t)blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<typename T> void run(const char* name, int blocks){
  T* d; CHK(hipMalloc(&d, sizeof(T)*blocks*256));
  const int iters=20000;
  hipLaunchKernelGGL((k<T>), dim3(blocks), dim3(256), 0, 0, d, 100, (T)1.0000001, (T)0.5);
  CHK(hipDeviceSynchronize());
  hipEvent_t e0,e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<T>), dim3(blocks), dim3(256), 0, 0, d, iters, (T)1.0000001, (T)0.5);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms,e0,e1));
  double fmas=(double)blocks*256*iters*32;
  // wave-instructions per SIMD: blocks*4 waves / (256 CUs * 4 SIMDs)
  double winstr_per_simd = (double)blocks*4*iters*32/(256.0*4.0);
  printf("%-6s blocks=%5d: %8.3f ms  %7.2f TFLOP/s  %.2f ns per wave-instruction per SIMD\n", name, blocks, ms, 2*fmas/ms/1e9, ms*1e6/winstr_per_simd);
  CHK(hipFree(d));
}
int main(){
  for(int blocks: {256, 1024, 2048, 4096}){ run<float>("fp32", blocks); run<double>("fp64", blocks); }
  return 0;
}

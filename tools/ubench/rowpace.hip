// micro-benchmark: how fast does a wave walk the bare recurrence row (C columns per lane: 2 DPP moves, 1 multiply,
// C fma, C adds) as a function of how many such waves run where?  Grid of `wgs` workgroups of `nw` waves (256 B ..
// 90 KB of dynamic LDS each, to steer how many share a compute unit); every wave walks `rows` rows.
// build: hipcc --offload-arch=gfx950 -O3 -o rowpace rowpace.hip ; ./rowpace
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ double shr_z(double v){ int lo=__builtin_amdgcn_update_dpp(0,__double2loint(v),0x138,0xf,0xf,true), hi=__builtin_amdgcn_update_dpp(0,__double2hiint(v),0x138,0xf,0xf,true); return __hiloint2double(hi,lo); }
template<int C> __global__ __launch_bounds__(512) void k(double* out, int rows, double a, int active_waves){
  extern __shared__ double dyn[];
  const int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  if (wave >= active_waves) return;
  double v[C], c[C], s=1.0+1e-12;
  for(int i=0;i<C;i++){ v[i]=1.0+lane*1e-3+i; c[i]=a+i*1e-3; }
  long long t0=wall_clock64(); long long c0=clock64();
  for(int r=0;r<rows;r+=8){
#pragma unroll
    for(int u=0;u<8;u++){
      const double t=shr_z(v[C-1])*s;
#pragma unroll
      for(int i=C-1;i>=1;i--) v[i]=fma(c[i],v[i],v[i-1]);
      v[0]=fma(c[0],v[0],t);
#pragma unroll
      for(int i=0;i<C;i++) c[i]+=1e-9;
    }
  }
  long long t1=wall_clock64(); long long c1=clock64();
  double acc=0; for(int i=0;i<C;i++) acc+=v[i]+c[i];
  if (acc==12345.678) dyn[lane]=acc;
  out[(size_t)(blockIdx.x*8+wave)*64+lane]=acc;
  if(lane==0){ out[(1<<20)+blockIdx.x*8+wave]=(double)(t1-t0); out[(2<<20)+blockIdx.x*8+wave]=(double)(c1-c0); }
}
template<int C> void run(double* d, int wgs, int nw, int active, size_t shm){
  const int rows=20000;
  CHK(hipFuncSetAttribute((const void*)k<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 1024));
  hipLaunchKernelGGL((k<C>), dim3(wgs), dim3(64*nw), shm, 0, d, 800, 0.5, active);
  CHK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<C>), dim3(wgs), dim3(64*nw), shm, 0, d, rows, 0.5, active);
  CHK(hipDeviceSynchronize());
  static double h[8192], hc[8192];
  CHK(hipMemcpy(h, d+(1<<20), sizeof(double)*wgs*8, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(hc, d+(2<<20), sizeof(double)*wgs*8, hipMemcpyDeviceToHost));
  double mx=0, mn=1e30, sum=0, cyc=0; int n=0;
  for(int b=0;b<wgs;b++) for(int w=0;w<active;w++){ double x=h[b*8+w]; mx=x>mx?x:mx; mn=x<mn?x:mn; sum+=x; cyc+=hc[b*8+w]; n++; }
  printf("C=%d wgs=%4d waves/wg=%d active=%d lds=%3zuKB : ns/row min %6.1f mean %6.1f max %6.1f ; cycles/row mean %6.1f\n", C, wgs, nw, active, shm/1024,
         mn*10.0/rows, sum/n*10.0/rows, mx*10.0/rows, cyc/n/rows);
}
int main(){
  double* d; CHK(hipMalloc(&d, sizeof(double)*(3<<20)+65536));
  const size_t big=84*1024, small=1024;
  for (int wgs : {1, 32, 64, 128, 256}) run<2>(d, wgs, 8, 4, big);
  for (int wgs : {64, 256}) run<2>(d, wgs, 4, 4, big);
  for (int wgs : {64, 256, 512}) run<2>(d, wgs, 8, 4, small);
  for (int wgs : {64, 256}) run<2>(d, wgs, 8, 8, big);
  for (int wgs : {64, 256}) run<2>(d, wgs, 8, 7, big);
  for (int wgs : {1, 64, 256}) run<4>(d, wgs, 8, 4, big);
  for (int wgs : {64, 256}) run<4>(d, wgs, 8, 8, big);
  for (int wgs : {512}) run<4>(d, wgs, 8, 7, 40*1024);
  // (round 5) 8 columns per lane: 464 own columns a strip, 22 strips a 10^4-column table
  for (int wgs : {1, 64, 256}) run<8>(d, wgs, 8, 4, big);
  for (int wgs : {256}) run<8>(d, wgs, 8, 8, big);
  for (int wgs : {512, 768}) run<8>(d, wgs, 3, 2, 50*1024);
  for (int wgs : {768}) run<4>(d, wgs, 5, 4, 50*1024);
  return 0;
}

// micro-benchmark: what it costs a lone spine wave (C = 2 columns per lane) to hand the last column of every
// row to another wave through LDS, by the form of the post.  One wave per workgroup, ns per row.
//   0  no post                      1  ds_write_b64 from all lanes, lane 63 to the ring, the others to scratch (one word each)
//   2  only lane 63 active (exec)   3  all lanes but 63 to ONE scratch word
//   4  readlane x2 + writelane x2 into a register pair, one ds_write_b64 per 8 rows
//   5  as 1 with ds_write_b32 x2    6  readlane x2, v_mov, ds_write from lane 0 only... (no: exec) -> skipped
// build: hipcc --offload-arch=gfx950 -O3 -o post post.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ int shr_i(int v, int fill){ return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double shr_d(double v, double f){ int lo=shr_i(__double2loint(v),__double2loint(f)), hi=shr_i(__double2hiint(v),__double2hiint(f)); return __hiloint2double(hi,lo); }
typedef __attribute__((address_space(3))) double lds_double;

template<int MODE> __global__ __launch_bounds__(64) void k(double* out, int iters, double a, double b){
  __shared__ double ring[256];
  __shared__ double pad[64 + 256 + 8];
  const int lane=threadIdx.x;
  double v0=1.0+lane*1e-3, v1=1.0+lane*2e-3, c0=a, c1=b, s=1.0+1e-9;
  const unsigned base = (lane == 63) ? (unsigned)(uintptr_t)(lds_double*)&ring[0] : (MODE == 3 ? (unsigned)(uintptr_t)(lds_double*)&pad[0] : (unsigned)(uintptr_t)(lds_double*)&pad[lane]);
  int acc_lo = 0, acc_hi = 0;
  long long t0=wall_clock64();
  for(int it=0; it<iters; it++){
    const unsigned wa = base + (unsigned)((it & 31) * 64);
    #pragma unroll
    for(int u=0;u<8;u++){
      double t=shr_d(v1, 0.0)*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1.0; c1+=1.0;
      if (MODE==1 || MODE==3) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(wa), "v"(v1), "n"(0) : "memory");
      if (MODE==2) { if (lane == 63) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(wa), "v"(v1), "n"(0) : "memory"); }
      if (MODE==4) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v1), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v1), 63);
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(acc_lo) : "s"(lo), "n"(u));
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(acc_hi) : "s"(hi), "n"(u));
        if (u == 7) { const double x = __hiloint2double(acc_hi, acc_lo); asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(uintptr_t)(lds_double*)&ring[0] + (unsigned)((it & 31) * 64 + (lane & 7) * 8)), "v"(x) : "memory"); }
      }
      if (MODE==5) { asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4" :: "v"(wa), "v"(__double2loint(v1)), "v"(__double2hiint(v1)) : "memory"); }
    }
  }
  long long t1=wall_clock64();
  double accd=v0+v1+c0+c1+ring[lane]+pad[lane]+acc_lo+acc_hi;
  out[blockIdx.x*64+lane]=accd;
  if(lane==0) out[4096+blockIdx.x]=(double)(t1-t0);
}
template<int MODE> void run(const char* name, double* d){
  const int iters=20000;
  hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, d, 100, 1.0000001, 0.5);
  CHK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, d, iters, 1.0000001, 0.5);
  CHK(hipDeviceSynchronize());
  double h; CHK(hipMemcpy(&h, d+4096, 8, hipMemcpyDeviceToHost));
  printf("%-64s %7.2f ns per row\n", name, h*10.0/(iters*8.0));
}
int main(){
  double* d; CHK(hipMalloc(&d, 8*(4096+1024)));
  run<0>("C=2 row, no post", d);
  run<1>("+ ds_write_b64 all lanes, scattered scratch", d);
  run<2>("+ ds_write_b64 lane 63 only (exec mask)", d);
  run<3>("+ ds_write_b64 all lanes, one scratch word", d);
  run<4>("+ readlane/writelane gather, one ds_write per 8 rows", d);
  run<5>("+ 2 x ds_write_b32 all lanes", d);
  return 0;
}

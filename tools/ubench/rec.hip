// micro-benchmark: what costs time in the recurrence row loop?  one wave per block, R rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__device__ __forceinline__ int shr_dpp(int v, int fill){ return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double shr_dpp(double v){ int lo=shr_dpp(__double2loint(v),0), hi=shr_dpp(__double2hiint(v),0); return __hiloint2double(hi,lo); }
__device__ __forceinline__ double shr_shfl(double v){ double r=__shfl_up(v,1,64); return (threadIdx.x&63)==0?0.0:r; }
// row_shr:1 within rows of 16 + fix lanes 16,32,48 through readlane
__device__ __forceinline__ int shr_row(int v){
  int r=__builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
  int l15=__builtin_amdgcn_readlane(v,15), l31=__builtin_amdgcn_readlane(v,31), l47=__builtin_amdgcn_readlane(v,47);
  int lane=threadIdx.x&63;
  r = lane==16? l15 : lane==32? l31 : lane==48? l47 : r;
  return r; }
__device__ __forceinline__ double shr_row(double v){ int lo=shr_row(__double2loint(v)), hi=shr_row(__double2hiint(v)); return __hiloint2double(hi,lo); }

template<int MODE, int STORE, int C>
__global__ __launch_bounds__(64) void k(double* out, int R, double a, size_t pitch){
  int lane=threadIdx.x; int c0=blockIdx.x*64*C+lane*C;
  double v[C], ca[C], s[C];
  for(int i=0;i<C;i++){ v[i]=1.0+1e-3*(c0+i); ca[i]=(c0+i)*a*1e-3; s[i]=1.0+1e-6*i; }
  double* row=out+ (size_t)blockIdx.y*pitch*R;
  for(int n=0;n<R;n++){
    double lfv = MODE==0? shr_dpp(v[C-1]) : MODE==1? shr_shfl(v[C-1]) : MODE==2? shr_row(v[C-1]) : v[C-1]*0.5;
    double nm1=(double)(n%7)*1e-3;
    #pragma unroll
    for(int i=C-1;i>=0;i--){ double lf=i>0? v[i-1]:lfv; v[i]=fma(nm1-ca[i], v[i], lf*s[i]); }
    if(STORE==1){
      if(C==1) row[c0]=v[0]; else for(int i=0;i<C;i+=2) *reinterpret_cast<double2*>(row+c0+i)=make_double2(v[i],v[i+1]);
    }
    row+=pitch;
  }
  if(STORE==0){ double acc=0; for(int i=0;i<C;i++) acc+=v[i]; if(acc==123.456) out[c0]=acc; }
}

template<int MODE,int STORE,int C> void run(const char* name, double* d, int blocks, int R, size_t pitch){
  hipEvent_t e0,e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE,STORE,C>), dim3(blocks,1), dim3(64), 0, 0, d, R, 0.5, pitch);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for(int it=0;it<20;it++) hipLaunchKernelGGL((k<MODE,STORE,C>), dim3(blocks,1), dim3(64), 0, 0, d, R, 0.5, pitch);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms,e0,e1));
  printf("%-28s C=%d blocks=%4d R=%d : %8.2f us/launch  %7.1f ns/row\n", name, C, blocks, R, ms*1e3/20, ms*1e6/20/R);
}
int main(){
  size_t pitch=10240; int R=512; double* d; CHK(hipMalloc(&d, pitch*R*sizeof(double)));
  for(int blocks: {1, 78, 156}){
    run<0,1,1>("dpp wave_shr + store", d, blocks, R, pitch);
    run<0,0,1>("dpp wave_shr, no store", d, blocks, R, pitch);
    run<1,1,1>("shfl_up + store", d, blocks, R, pitch);
    run<1,0,1>("shfl_up, no store", d, blocks, R, pitch);
    run<2,1,1>("row_shr+readlane + store", d, blocks, R, pitch);
    run<2,0,1>("row_shr+readlane, no store", d, blocks, R, pitch);
    run<3,1,1>("no shift + store", d, blocks, R, pitch);
    run<3,0,1>("no shift, no store", d, blocks, R, pitch);
    if(blocks<=78){ run<0,1,2>("dpp wave_shr + store", d, blocks, R, pitch); run<2,1,2>("row_shr+readlane + store", d, blocks, R, pitch); run<2,0,2>("row_shr+readlane, no store", d, blocks, R, pitch);}
  }
  return 0;
}

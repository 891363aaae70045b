// micro-benchmark: issue / latency cost of the instructions on the recurrence row chain (one wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ int shr_i(int v, int fill){ return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double shr_d(double v, double f){ int lo=shr_i(__double2loint(v),__double2loint(f)), hi=shr_i(__double2hiint(v),__double2hiint(f)); return __hiloint2double(hi,lo); }

template<int MODE> __global__ __launch_bounds__(64) void k(double* out, int iters, double a, double b){
  __shared__ double lds[8][128];
  const int lane=threadIdx.x;
  double v0=1.0+lane*1e-3, v1=1.0+lane*2e-3, c0=a, c1=b, s=1.0+1e-9;
  float f=1.0f+lane; int inc0 = (int)(a*0) + 1, inc1 = (int)(b*0) + 1;
  double w[8]; for(int i=0;i<8;i++) w[i]=1.0+i*1e-3+lane*1e-6;
  long long t0=wall_clock64();
  for(int it=0; it<iters; it++){
    #pragma unroll
    for(int u=0;u<8;u++){
      if(MODE==0){ f = f*1.0000001f + 0.5f; }                         // dependent f32 fma
      if(MODE==1){ v0 = fma(c0, v0, c1); }                             // dependent f64 fma
      if(MODE==2){ w[u] = fma(c0, w[u], c1); }                         // 8 independent f64 fma chains
      if(MODE==3){ double t=shr_d(v0, 0.0)*s; v0=fma(c0,v0,t); }       // C=1 row: dpp, mul, fma
      if(MODE==4){ double t=shr_d(v0, 0.0)*s; v0=fma(c0,v0,t); c0+=1.0; lds[u][lane]=v0; } // + coef add + lds write
      if(MODE==5){ double t=shr_d(v1, 0.0)*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1.0; c1+=1.0;
                   *reinterpret_cast<double2*>(&lds[u][2*lane])=make_double2(v0,v1); } // C=2 row
      if(MODE==6){ // C=2 row with the scale folded into the shifted exponent (integer add on the high word)
                   int lo=shr_i(__double2loint(v1),0), hi=shr_i(__double2hiint(v1),0)+0x00100000; double t=__hiloint2double(hi,lo);
                   v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1.0; c1+=1.0;
                   *reinterpret_cast<double2*>(&lds[u][2*lane])=make_double2(v0,v1); }
      if(MODE==7){ v0 = v0 + 1.0; }                                    // dependent f64 add
      if(MODE==8){ int x=__double2loint(v0); x=shr_i(x,0); v0=__hiloint2double(__double2hiint(v0), x); } // dependent dpp
      if(MODE==9){ int x=__double2loint(v0); x=__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); v0=__hiloint2double(__double2hiint(v0), x); } // row_shr:1
      if(MODE==10){ int x=__double2loint(v0); x=__builtin_amdgcn_update_dpp(x, x, 0x142, 0xe, 0xf, false); v0=__hiloint2double(__double2hiint(v0), x); } // row_bcast:15
      if(MODE==11){ // C=2 row with the shift as row_shr:1 + row_bcast:15 (4 cheap DPP moves instead of 2 wave_shr)
                   int lo=__double2loint(v1), hi=__double2hiint(v1);
                   int tl=__builtin_amdgcn_update_dpp(0, lo, 0x142, 0xe, 0xf, false), th=__builtin_amdgcn_update_dpp(0, hi, 0x142, 0xe, 0xf, false);
                   tl=__builtin_amdgcn_update_dpp(tl, lo, 0x111, 0xf, 0xf, false); th=__builtin_amdgcn_update_dpp(th, hi, 0x111, 0xf, 0xf, false);
                   double t=__hiloint2double(th,tl)*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1.0; c1+=1.0; }
      if(MODE==13){ // C=2 row, coefficients advanced by an integer add on the high word (same binade)
                   double t=shr_d(v1, 0.0)*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t);
                   c0=__hiloint2double(__double2hiint(c0)+inc0, __double2loint(c0)); c1=__hiloint2double(__double2hiint(c1)+inc1, __double2loint(c1)); }
      if(MODE==14){ // ... and the scale as an exponent add
                   int lo=shr_i(__double2loint(v1),0), hi=shr_i(__double2hiint(v1),0)+0x00100000; double t=__hiloint2double(hi,lo);
                   v1=fma(c1,v1,v0); v0=fma(c0,v0,t);
                   c0=__hiloint2double(__double2hiint(c0)+inc0, __double2loint(c0)); c1=__hiloint2double(__double2hiint(c1)+inc1, __double2loint(c1)); }
      if(MODE==15){ // C=4 row, fp64 coefficient adds
                   double t=shr_d(w[3], 0.0)*s; w[3]=fma(w[7],w[3],w[2]); w[2]=fma(w[6],w[2],w[1]); w[1]=fma(w[5],w[1],w[0]); w[0]=fma(w[4],w[0],t);
                   w[4]+=1.0; w[5]+=1.0; w[6]+=1.0; w[7]+=1.0; }
      if(MODE==16){ // C=4 row, integer coefficient adds
                   double t=shr_d(w[3], 0.0)*s; w[3]=fma(w[7],w[3],w[2]); w[2]=fma(w[6],w[2],w[1]); w[1]=fma(w[5],w[1],w[0]); w[0]=fma(w[4],w[0],t);
                   for(int q=4;q<8;q++) w[q]=__hiloint2double(__double2hiint(w[q])+inc0, __double2loint(w[q])); }
      if(MODE==12){ double t=shr_d(v1, 0.0)*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1.0; c1+=1.0; } // C=2 row, no store
    }
  }
  long long t1=wall_clock64();
  double acc=v0+v1+f+c0+c1; for(int i=0;i<8;i++) acc+=w[i]; acc+=lds[3][lane];
  out[blockIdx.x*64+lane]=acc; 
  if(lane==0) out[4096+blockIdx.x]=(double)(t1-t0);
}
template<int MODE> void run(const char* name, double* d, int blocks){
  const int iters=20000;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, 100, 1.0000001, 0.5);
  CHK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.0000001, 0.5);
  CHK(hipDeviceSynchronize());
  double h; CHK(hipMemcpy(&h, d+4096, 8, hipMemcpyDeviceToHost));
  printf("%-44s blocks=%4d: %7.2f ns per step (wall_clock64 ticks %.0f)\n", name, blocks, h*10.0/(iters*8.0), h);
}
int main(){
  double* d; CHK(hipMalloc(&d, 8*(4096+1024)));
  for(int blocks: {1}){
    run<0>("dependent f32 fma", d, blocks);
    run<1>("dependent f64 fma", d, blocks);
    run<7>("dependent f64 add", d, blocks);
    run<2>("8 independent f64 fma chains (per fma)", d, blocks);
    run<8>("dependent dpp mov", d, blocks);
    run<3>("C=1 row: dpp x2, mul, fma", d, blocks);
    run<4>("C=1 row + coef add + ds_write", d, blocks);
    run<5>("C=2 row: dpp x2, mul, 2 fma, 2 add, ds_write", d, blocks);
    run<6>("C=2 row, exponent add instead of mul", d, blocks);
    run<9>("dependent dpp row_shr:1", d, blocks);
    run<10>("dependent dpp row_bcast:15", d, blocks);
    run<12>("C=2 row, wave_shr, no store", d, blocks);
    run<11>("C=2 row, row_shr + row_bcast15, no store", d, blocks);
    run<13>("C=2 row, integer coefficient adds", d, blocks);
    run<14>("C=2 row, integer coefficient adds + exponent add", d, blocks);
    run<15>("C=4 row", d, blocks);
    run<16>("C=4 row, integer coefficient adds", d, blocks);
  }
  return 0;
}

// micro-benchmark: what a producer trip of the chain form costs beyond its 8 bare rows (one wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ int shr_i(int v, int fill){ return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double shr_d(double v, double f){ int lo=shr_i(__double2loint(v),__double2loint(f)), hi=shr_i(__double2hiint(v),__double2hiint(f)); return __hiloint2double(hi,lo); }
__device__ __forceinline__ int lds_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_post(int *p, int v) { asm volatile("" ::: "memory"); if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory"); }

template<int MODE> __global__ __launch_bounds__(64) void k(double* out, double* tab, int trips, double a, size_t pitch){
  __shared__ double vbuf[4][8][128];
  __shared__ double edge[32*8];
  __shared__ int cnt[4];
  const int lane=threadIdx.x;
  if(lane<4) cnt[lane]=0x7fffffff;
  for(int i=lane;i<256;i+=64) edge[i]=1e-30;
  __syncthreads();
  double v0=1.0+lane*1e-3, v1=1.0+lane*2e-3, c0=a, c1=a+0.1, s=1.0+1e-9;
  double* p = tab + 2*lane;
  double ne[8];
  for(int u=0;u<8;u++) ne[u]=edge[u];
  int nl=lds_peek(&cnt[0]), nn=lds_peek(&cnt[1]);
  long long t0=wall_clock64();
  for(int g=0; g<trips; g++){
    double e[8];
    #pragma unroll
    for(int u=0;u<8;u++) e[u] = (MODE>=1) ? ne[u] : 0.0;
    if(MODE>=3){
      if(nl < g+1 || nn < g-2) { while(lds_peek(&cnt[0]) < g+1) __builtin_amdgcn_s_sleep(1); }
      nl=lds_peek(&cnt[0]); nn=lds_peek(&cnt[1]);
      asm volatile("" ::: "memory");
    }
    if(MODE>=1){
      #pragma unroll
      for(int u=0;u<8;u++) ne[u]=edge[((g+1)&31)*8+u];
    }
    #pragma unroll
    for(int u=0;u<8;u++){
      double t=shr_d(v1, e[u])*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1.0; c1+=1.0;
      *reinterpret_cast<double2*>(&vbuf[g&3][u][2*lane])=make_double2(v0,v1);
      if(MODE>=2){ *reinterpret_cast<double2*>(p)=make_double2(v0,v1); p+=pitch; }
    }
    if(MODE>=2 && (g&63)==63) p -= 512*pitch;   // stay inside the buffer
    if(MODE>=3) lds_post(&cnt[2], g+1);
    if(MODE>=4){ asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); lds_post(&cnt[3], g-5); }
  }
  long long t1=wall_clock64();
  out[blockIdx.x*64+lane]=v0+v1+c0+c1+vbuf[1][2][lane];
  if(lane==0) out[4096+blockIdx.x]=(double)(t1-t0);
}
template<int MODE> void run(const char* name, double* d, double* tab, int blocks){
  const int trips=20000;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, tab, 100, 1.0000001, (size_t)10240);
  CHK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, tab, trips, 1.0000001, (size_t)10240);
  CHK(hipDeviceSynchronize());
  double h; CHK(hipMemcpy(&h, d+4096, 8, hipMemcpyDeviceToHost));
  printf("%-60s: %7.1f ns per trip, %6.2f ns per row\n", name, h*10.0/trips, h*10.0/(trips*8.0));
}
int main(){
  double* d; CHK(hipMalloc(&d, 8*(4096+1024)));
  double* tab; CHK(hipMalloc(&tab, 8ull*10240*520));
  run<0>("8 rows (C=2: dpp, mul, 2 fma, 2 add, ds_write_b128)", d, tab, 1);
  run<1>("+ left inputs: 8 ds_read_b64 a trip ahead, fill into lane 0", d, tab, 1);
  run<2>("+ global_store_dwordx4 per row, pointer += pitch", d, tab, 1);
  run<3>("+ two counters peeked a trip ahead, one posted", d, tab, 1);
  run<4>("+ s_waitcnt vmcnt(48), second post", d, tab, 1);
  run<0>("8 rows, 256 blocks", d, tab, 256);
  run<4>("everything, 256 blocks (stores collide: same rows)", d, tab, 256);
  run<4>("everything, 20 blocks", d, tab, 20);
  return 0;
}

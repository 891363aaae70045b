// micro-benchmark: one lone wave running the trip of k_fill_ck's spine (C = 2), pieces switched on one by one:
//   1 edge post (ds_write_b64 per row)      2 look-ahead loads (4 x ds_read_b128 at row 4, used next trip)
//   4 left counter read + check branch      8 progress post (ds_write_b32)
//  16 period branch (renormalise every 6th trip)   32 ring guard every 8th trip (two counter reads)
// build: hipcc --offload-arch=gfx950 -O3 -o spine spine.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ int shr_i(int v, int fill){ return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double shr_d(double v, double f){ int lo=shr_i(__double2loint(v),__double2loint(f)), hi=shr_i(__double2hiint(v),__double2hiint(f)); return __hiloint2double(hi,lo); }
typedef __attribute__((address_space(3))) double lds_double;
typedef double d2 __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ void st1(unsigned addr, double x) { asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(x), "n"(OFF) : "memory"); }

template<int MODE> __global__ __launch_bounds__(64) void k(double* out, int trips, double a, double b){
  __shared__ __attribute__((aligned(16))) double ring[256], left[256];
  __shared__ double pad[64 + 256 + 8];
  __shared__ int cnt[8], ppad[64];
  const int lane=threadIdx.x;
  for (int i = lane; i < 256; i += 64) { left[i] = 1e-300 * (i + 1); ring[i] = 0; }
  if (lane < 8) cnt[lane] = 1 << 30;
  __syncthreads();
  double v0=1.0+lane*1e-3, v1=1.0+lane*2e-3, c0=a, c1=b, s=1.0+1e-9;
  int ep = 700;
  const unsigned base = (lane == 63) ? (unsigned)(uintptr_t)(lds_double*)&ring[0] : (unsigned)(uintptr_t)(lds_double*)&pad[lane];
  int *post_addr = (lane == 0) ? &cnt[1] : &ppad[lane];
  int n_left = 1 << 30, tin = 0;
  double ea[8], eb[8];
  for (int u = 0; u < 8; u++) { ea[u] = left[u]; eb[u] = left[8 + u]; }
  auto trip = [&](int g, double (&e)[8], double (&en)[8]) {
    if (MODE & 4) { if (__builtin_expect(n_left < g + 1, 0)) { while (__hip_atomic_load(&cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < g + 1) __builtin_amdgcn_s_sleep(1); } }
    if (MODE & 32) { if (__builtin_expect((g & 7) == 0, 0)) { while (__hip_atomic_load(&cnt[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < g - 23 || __hip_atomic_load(&cnt[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < g - 24) __builtin_amdgcn_s_sleep(1); } }
    if (MODE & 16) {
      if (__builtin_expect(tin == 0, 0)) {
        int kmax = max(__builtin_amdgcn_frexp_exp(v0), __builtin_amdgcn_frexp_exp(v1));
        v0 = ldexp(v0, -kmax - 700); v1 = ldexp(v1, -kmax - 700); ep += kmax + 700;
        int dl = shr_i(ep, ep) - ep;
        s = ldexp(1.0, min(max(dl, -1100), 220));
      }
    }
    const unsigned wa = base + (unsigned)((g & 31) * 64), wan = base + (unsigned)(((g + 1) & 31) * 64);
    auto row = [&](auto uc) {
      constexpr int u = decltype(uc)::value;
      double t=shr_d(v1, e[u])*s; v1=fma(c1,v1,v0); v0=fma(c0,v0,t); c0+=1.0; c1+=1.0;
      if (MODE & 1) { if constexpr (u < 7) st1<(u + 1) * 8>(wa, v1); else st1<0>(wan, v1); }
      if constexpr (u == 4) {
        if (MODE & 4) n_left = __hip_atomic_load(&cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        if (MODE & 2) {
          const d2 *src = reinterpret_cast<const d2 *>(left + ((g + 1) & 31) * 8);
          #pragma unroll
          for (int q = 0; q < 4; q++) { const d2 t2 = src[q]; en[2*q] = t2.x; en[2*q+1] = t2.y; }
        }
      }
    };
    row(std::integral_constant<int,0>{}); row(std::integral_constant<int,1>{}); row(std::integral_constant<int,2>{}); row(std::integral_constant<int,3>{});
    row(std::integral_constant<int,4>{}); row(std::integral_constant<int,5>{}); row(std::integral_constant<int,6>{}); row(std::integral_constant<int,7>{});
    asm volatile("" ::: "memory");
    if (MODE & 8) __hip_atomic_store(post_addr, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    if (++tin == 6) tin = 0;
  };
  long long t0=wall_clock64();
  int g = 0;
  for (; g + 1 < trips; g += 2) { trip(g, ea, eb); trip(g + 1, eb, ea); }
  long long t1=wall_clock64();
  double accd=v0+v1+c0+c1+ring[lane]+pad[lane]+ea[3]+eb[5]+cnt[1]+ep;
  out[blockIdx.x*64+lane]=accd;
  if(lane==0) out[4096+blockIdx.x]=(double)(t1-t0);
}
template<int MODE> void run(const char* name, double* d){
  const int trips=20000;
  hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, d, 100, 1.0000001, 0.5);
  CHK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, d, trips, 1.0000001, 0.5);
  CHK(hipDeviceSynchronize());
  double h; CHK(hipMemcpy(&h, d+4096, 8, hipMemcpyDeviceToHost));
  printf("%-72s %7.2f ns per row\n", name, h*10.0/(trips*8.0));
}
int main(){
  double* d; CHK(hipMalloc(&d, 8*(4096+1024)));
  run<0>("bare rows (left inputs from registers)", d);
  run<1>("+ edge post", d);
  run<2>("+ look-ahead loads only", d);
  run<3>("+ edge post + look-ahead loads", d);
  run<7>("+ edge post + loads + left counter", d);
  run<15>("+ edge post + loads + counter + progress post", d);
  run<31>("+ ... + period branch", d);
  run<63>("+ ... + ring guard every 8th trip   (= the kernel's trip)", d);
  run<62>("the kernel's trip without the edge post", d);
  return 0;
}

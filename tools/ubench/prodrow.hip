// diagnostic: what one producer wave's row costs, alone on its compute unit -- the chain kernel's row
// loop (C = 2 columns a lane) with pieces removed.  build: hipcc --offload-arch=gfx950 -O3 -o prodrow prodrow.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ int shr_i(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ unsigned lds_addr(double *p) { return (unsigned)(uintptr_t)(lds_double *)p; }
template <int OFF>
__device__ __forceinline__ void lds_store2(unsigned addr, double x, double y) {
  asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(x), "v"(y), "n"(OFF), "n"(OFF + 1) : "memory");
}
// bits of V: 1 no LDS store, 2 no DPP shift (left input = e only), 4 no coefficient update, 8 no scale multiply,
// 16 no left-input loads, 32 post a progress word per trip
template <int V>
__global__ void k_rows(double *out, const double *edge, int trips, double a, unsigned long long *ticks) {
  __shared__ double vbuf[4][8][128];
  __shared__ double ein[256];
  __shared__ int post[64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) ein[i] = edge[i];
  __syncthreads();
  double v0 = 1.0 + lane, v1 = 2.0 + lane, c0 = a, c1 = a + 0.1, s = 1.0000001;
  double e[8];
  for (int u = 0; u < 8; u++) e[u] = ein[u];
  const unsigned long long t0 = wall_clock64();
  for (int g = 0; g < trips; g++) {
    if (!(V & 16)) {
      const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&ein[(g & 31) * 8]);
#pragma unroll
      for (int u = 0; u < 8; u++)
        e[u] = __longlong_as_double((long long)__hip_atomic_load(src + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    const unsigned sb = lds_addr(&vbuf[g & 3][0][2 * lane]);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      double t;
      if (V & 2) t = e[u] + v1;
      else {
        int lo = shr_i(__double2loint(v1), __double2loint(e[u])), hi = shr_i(__double2hiint(v1), __double2hiint(e[u]));
        t = __hiloint2double(hi, lo);
      }
      if (!(V & 8)) t *= s;
      v1 = fma(c1, v1, v0);
      v0 = fma(c0, v0, t);
      if (!(V & 4)) { c0 += 1.0; c1 += 1.0; }
      if (!(V & 1)) {
        if (u & 1) lds_store2<128>(sb + (u >> 1) * 2048, v0, v1);
        else lds_store2<0>(sb + (u >> 1) * 2048, v0, v1);
      }
    }
    if (V & 32) {
      asm volatile("" ::: "memory");
      __hip_atomic_store(&post[lane], g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
    }
    // keep the values bounded
    if ((g & 15) == 15) { v0 = ldexp(v0, -__builtin_amdgcn_frexp_exp(v0)); v1 = ldexp(v1, -__builtin_amdgcn_frexp_exp(v1)); c0 = a; c1 = a + 0.1; }
  }
  const unsigned long long t1 = wall_clock64();
  __syncthreads();
  out[lane] = v0 + v1 + vbuf[1][2][lane] + post[lane];
  if (lane == 0) ticks[0] = t1 - t0;
}
template <int V>
void run(const char *what, double *out, double *edge, unsigned long long *ticks) {
  const int trips = 20000;
  hipLaunchKernelGGL(k_rows<V>, dim3(1), dim3(64), 0, 0, out, edge, trips, 0.5, ticks);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_rows<V>, dim3(1), dim3(64), 0, 0, out, edge, trips, 0.5, ticks);
  CHECK(hipDeviceSynchronize());
  unsigned long long h;
  CHECK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
  printf("%-58s %6.1f ns/row  %6.3f us/trip\n", what, h * 10.0 / trips / 8, h / 100.0 / trips);
}
int main() {
  double *out, *edge;
  unsigned long long *ticks;
  CHECK(hipMalloc(&out, 64 * 8));
  CHECK(hipMalloc(&edge, 256 * 8));
  CHECK(hipMalloc(&ticks, 8));
  double h[256];
  for (int i = 0; i < 256; i++) h[i] = 1.0 + i * 1e-3;
  CHECK(hipMemcpy(edge, h, sizeof(h), hipMemcpyHostToDevice));
  run<32>("the row loop as in k_fill_chain (C=2), progress post", out, edge, ticks);
  run<0>("... without the progress post", out, edge, ticks);
  run<1>("... without the LDS stores", out, edge, ticks);
  run<2>("... without the DPP shift", out, edge, ticks);
  run<4>("... without the coefficient updates", out, edge, ticks);
  run<8>("... without the scale multiply", out, edge, ticks);
  run<16>("... without the left-input loads", out, edge, ticks);
  run<1 | 2 | 4 | 8 | 16>("only the two fma per row", out, edge, ticks);
  run<1 | 4 | 8 | 16>("DPP + two fma", out, edge, ticks);
  return 0;
}

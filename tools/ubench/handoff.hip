// diagnostic: one-way latency of an 8-byte flag hand-off between two workgroups through global
// memory, by placement (same XCD / different XCDs) and by the store's scope bits.
// build: hipcc --offload-arch=gfx950 -O3 -o handoff handoff.hip ; run: ./handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}
template <int WIDE>
__device__ __forceinline__ void put(unsigned long long *p, unsigned long long v) {
  if (WIDE) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned long long get(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// blocks A and B play ping-pong on two flags 4 KB apart; everybody else leaves at once
template <int WIDE>
__global__ void k_pingpong(unsigned long long *flags, int A, int B, int iters, unsigned long long *out, unsigned *xcd) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) xcd[b] = xcc_id();
  if (b != A && b != B) return;
  if (threadIdx.x != 0) return;
  unsigned long long *ab = flags, *ba = flags + 512;
  const unsigned long long limit = 200000000ull;  // 2 s of the 100 MHz clock: never hang
  const unsigned long long t0 = wall_clock64();
  if (b == A) {
    for (int i = 1; i <= iters; i++) {
      put<WIDE>(ab, (unsigned long long)i);
      while (get(ba) != (unsigned long long)i)
        if (wall_clock64() - t0 > limit) { out[1] = 1; return; }
    }
    out[0] = wall_clock64() - t0;
  } else {
    for (int i = 1; i <= iters; i++) {
      while (get(ab) != (unsigned long long)i)
        if (wall_clock64() - t0 > limit) { out[1] = 2; return; }
      put<WIDE>(ba, (unsigned long long)i);
    }
  }
}

int main() {
  unsigned long long *flags, *out;
  unsigned *xcd;
  CHECK(hipMalloc(&flags, 8192));
  CHECK(hipMalloc(&out, 16));
  CHECK(hipMalloc(&xcd, 64 * 4));
  const int iters = 2000;
  for (int wide = 0; wide < 2; wide++)
    for (int B : {8, 16, 1, 2, 5}) {
      CHECK(hipMemset(flags, 0, 8192));
      CHECK(hipMemset(out, 0, 16));
      if (wide) hipLaunchKernelGGL(k_pingpong<1>, dim3(32), dim3(64), 0, 0, flags, 0, B, iters, out, xcd);
      else hipLaunchKernelGGL(k_pingpong<0>, dim3(32), dim3(64), 0, 0, flags, 0, B, iters, out, xcd);
      CHECK(hipDeviceSynchronize());
      unsigned long long h[2];
      unsigned hx[32];
      CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hx, xcd, 32 * 4, hipMemcpyDeviceToHost));
      printf("store %s  blocks 0 (xcd %u) <-> %d (xcd %u): %s one-way %.3f us\n", wide ? "sc1 (agent)" : "sc0 (workgroup)", hx[0], B,
             hx[B], h[1] ? "TIMED OUT" : "ok", h[0] / 100.0 / iters / 2.0);
    }
  return 0;
}

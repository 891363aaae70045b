// diagnostic: one-way latency of an 8-byte flag hand-off between two workgroups through global memory, by
// placement (same XCD / different XCDs) and by the scope bits of BOTH the store and the polling load
// (inline asm: sc1 = device scope, sc0 = group scope / L1 bypass, none = wave scope).
// build: hipcc --offload-arch=gfx950 -O3 -o handoff2 handoff2.hip ; run: ./handoff2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}
template <int S>
__device__ __forceinline__ void put(unsigned long long *p, unsigned long long v) {
  if (S == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else if (S == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int L>
__device__ __forceinline__ unsigned long long get(const unsigned long long *p) {
  unsigned long long v;
  if (L == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (L == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int S, int L>
__global__ void k_pingpong(unsigned long long *flags, int A, int B, int iters, unsigned long long *out, unsigned *xcd) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) xcd[b] = xcc_id();
  if (b != A && b != B) return;
  if (threadIdx.x != 0) return;
  unsigned long long *ab = flags, *ba = flags + 512;
  const unsigned long long limit = 20000000ull;  // 0.2 s of the 100 MHz clock: never hang
  const unsigned long long t0 = wall_clock64();
  if (b == A) {
    for (int i = 1; i <= iters; i++) {
      put<S>(ab, (unsigned long long)i);
      while (get<L>(ba) != (unsigned long long)i)
        if (wall_clock64() - t0 > limit) { out[1] = 1; return; }
    }
    out[0] = wall_clock64() - t0;
  } else {
    for (int i = 1; i <= iters; i++) {
      while (get<L>(ab) != (unsigned long long)i)
        if (wall_clock64() - t0 > limit) { out[1] = 2; return; }
      put<S>(ba, (unsigned long long)i);
    }
  }
}

template <int S, int L>
void run(unsigned long long *flags, unsigned long long *out, unsigned *xcd, int B) {
  const int iters = 2000;
  CHECK(hipMemset(flags, 0, 8192));
  CHECK(hipMemset(out, 0, 16));
  hipLaunchKernelGGL((k_pingpong<S, L>), dim3(32), dim3(64), 0, 0, flags, 0, B, iters, out, xcd);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[2];
  unsigned hx[32];
  CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hx, xcd, 32 * 4, hipMemcpyDeviceToHost));
  const char *sn[] = {"plain", "sc0", "sc1"};
  printf("store %-5s load %-5s  blocks 0 (xcd %u) <-> %2d (xcd %u): %s one-way %.3f us\n", sn[S], sn[L], hx[0], B, hx[B],
         h[1] ? "TIMED OUT" : "ok", h[0] / 100.0 / iters / 2.0);
}

int main() {
  unsigned long long *flags, *out;
  unsigned *xcd;
  CHECK(hipMalloc(&flags, 8192));
  CHECK(hipMalloc(&out, 16));
  CHECK(hipMalloc(&xcd, 64 * 4));
  for (int B : {8, 16, 1, 3}) {
    run<2, 2>(flags, out, xcd, B);
    run<0, 1>(flags, out, xcd, B);
    run<1, 1>(flags, out, xcd, B);
    run<0, 2>(flags, out, xcd, B);
    run<2, 1>(flags, out, xcd, B);
    run<0, 0>(flags, out, xcd, B);
  }
  return 0;
}

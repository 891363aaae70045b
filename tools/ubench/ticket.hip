// micro-benchmark: how many tickets per second one atomic counter in global memory hands out when every wave
// of the chip asks (lane 0 of each wave: atomicAdd, wait for the value, ask again), and how batching helps.
// build: hipcc --offload-arch=gfx950 -O3 -o ticket ticket.hip ; run: ./ticket
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int BATCH, int WORK>
__global__ __launch_bounds__(512) void k(unsigned *counter, unsigned total, double *sink) {
  const int lane = threadIdx.x & 63;
  double acc = lane;
  for (;;) {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(counter, (unsigned)BATCH);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= total) break;
    for (int b = 0; b < BATCH; b++)
      for (int i = 0; i < WORK; i++) acc = fma(acc, 1.0000001, 1e-9);   // (what a tile costs: WORK dependent fma)
  }
  if (acc == 12345.678) sink[0] = acc;
}

template <int BATCH, int WORK>
void run(unsigned *counter, double *sink, int grid, unsigned total) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  double best = 1e9;
  for (int r = 0; r < 3; r++) {
    CHK(hipMemsetAsync(counter, 0, 4, 0));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<BATCH, WORK>), dim3(grid), dim3(512), 0, 0, counter, total, sink);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("grid %3d  batch %d  work %5d fma per ticket: %7.3f ms for %u tickets = %6.1f M tickets/s (%6.1f M atomics/s)\n", grid, BATCH, WORK, best, total,
         total / best / 1e3, total / (double)BATCH / best / 1e3);
}

int main() {
  unsigned *counter; double *sink;
  CHK(hipMalloc(&counter, 256)); CHK(hipMalloc(&sink, 64));
  const unsigned total = 400000;
  for (int grid : {64, 152, 256}) {
    run<1, 0>(counter, sink, grid, total);
    run<1, 1000>(counter, sink, grid, total);
    run<1, 4000>(counter, sink, grid, total);
    run<4, 1000>(counter, sink, grid, total);
  }
  return 0;
}

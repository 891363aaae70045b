// micro-benchmark: how the HBM write bandwidth of the tile workers' store pattern scales with the number of
// compute units that write, and what a lane's two 16-byte stores 32 bytes apart (4 columns per lane) cost
// against two stores that each cover a contiguous kilobyte.
//   pattern 0  lane l writes 16 B at row + 32 l and at row + 32 l + 16      (k_fill_ck / k_fill_hb, C = 4)
//   pattern 1  lane l writes 16 B at row + 16 l and at row + 1024 + 16 l    (every instruction: 8 whole lines)
//   pattern 2  lane l writes 16 B at row + 16 l only (1 KB rows: C = 2)
// build: hipcc --offload-arch=gfx950 -O3 -o wcap wcap.hip ; run: ./wcap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int PAT>
__global__ __launch_bounds__(512) void k(char *base, unsigned *ticket, size_t pitch, int strips, int rows, unsigned total) {
  const int lane = threadIdx.x & 63;
  const d2 val = {1.0 + lane, 2.0};
  for (;;) {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= total) break;
    const unsigned jw = t % strips, b = t / strips;
    char *p = base + (size_t)b * rows * pitch + (size_t)jw * 2048;
    for (int r = 0; r < rows; r++) {
      char *q = p + (size_t)r * pitch;
      if (PAT == 0) {
        *(d2 *)(q + lane * 32) = val;
        *(d2 *)(q + lane * 32 + 16) = val;
      } else if (PAT == 1) {
        *(d2 *)(q + lane * 16) = val;
        *(d2 *)(q + 1024 + lane * 16) = val;
      } else {
        *(d2 *)(q + lane * 16) = val;
      }
    }
  }
}

template <int PAT>
double run(char *buf, unsigned *ticket, size_t pitch, int strips, int rows, unsigned total, int grid) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  double best = 1e9;
  for (int it = 0; it < 4; it++) {
    CHK(hipMemsetAsync(ticket, 0, 4, 0));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<PAT>), dim3(grid), dim3(512), 0, 0, buf, ticket, pitch, strips, rows, total);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const int rows = 48, strips = 40, blocks = 832;   // 40 x 2 KB per row, 832 blocks of 48 rows: 3.27 GB
  const size_t pitch = 80 * 1024;
  char *buf; unsigned *ticket;
  CHK(hipMalloc(&buf, (size_t)blocks * rows * pitch + (1 << 20)));
  CHK(hipMalloc(&ticket, 256));
  const unsigned total = (unsigned)strips * blocks;
  for (int grid : {32, 64, 128, 152, 192, 256, 512}) {
    const double gb2 = (double)total * rows * 2048 / 1e9, gb1 = gb2 / 2;
    double t0 = run<0>(buf, ticket, pitch, strips, rows, total, grid);
    double t1 = run<1>(buf, ticket, pitch, strips, rows, total, grid);
    double t2 = run<2>(buf, ticket, pitch, strips, rows, total, grid);
    printf("grid %3d: 2x16B 32 apart %.3f ms %.2f TB/s (%.1f GB/s per WG) | 2 x contiguous KB %.3f ms %.2f TB/s (%.1f) | 1 KB rows %.3f ms %.2f TB/s (%.1f)\n",
           grid, t0, gb2 / t0, gb2 / t0 * 1000 / grid, t1, gb2 / t1, gb2 / t1 * 1000 / grid, t2, gb1 / t2, gb1 / t2 * 1000 / grid);
  }
  return 0;
}

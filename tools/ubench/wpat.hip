// micro-benchmark: what HBM write bandwidth a table-fill store pattern gets.  Every wave writes tiles of
// ROWS rows x (16*LPW bytes per lane) with the rows `pitch` bytes apart; what varies is which tiles the
// waves that run at the same time are writing:
//   0  every wave takes tiles from one global ticket (neighbouring tiles go to unrelated waves)
//   1  the 8 waves of a workgroup take 8 horizontally adjacent tiles (8 KB contiguous per row), no sync
//   2  ... with a barrier every 8 rows
//   3  linear: a workgroup writes one contiguous region front to back
// build: hipcc --offload-arch=gfx950 -O3 -o wpat wpat.hip ; run: ./wpat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE, int NT>
__global__ __launch_bounds__(512) void k(char *base, unsigned *ticket, size_t pitch, int strips, int blocks_v, int rows, size_t table_bytes, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ unsigned s_t;
  const d2 val = {1.0 + lane, 2.0};
  if (MODE == 3) {
    // linear: workgroup b writes [b, b+1) * chunk, 8 KB per step (8 waves x 1 KB)
    const size_t total = table_bytes * D;
    const size_t chunk = total / gridDim.x / 8192 * 8192;
    char *p = base + (size_t)blockIdx.x * chunk + wave * 1024 + lane * 16;
    for (size_t o = 0; o < chunk; o += 8192) {
      if (NT) __builtin_nontemporal_store(val, (d2 *)(p + o));
      else *(d2 *)(p + o) = val;
    }
    return;
  }
  const unsigned tiles_per_table = (unsigned)strips * blocks_v;
  const unsigned total = tiles_per_table * D;
  for (;;) {
    unsigned t;
    if (MODE == 0) {
      t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1u);
      t = __builtin_amdgcn_readfirstlane(t);
    } else {
      __syncthreads();
      if (threadIdx.x == 0) s_t = atomicAdd(ticket, 8u);
      __syncthreads();
      t = s_t + wave;
    }
    if (t >= total) break;
    unsigned d, q;
    if (MODE == 0) { d = t % D; q = t / D; }                               // tables interleaved, tile by tile
    else { const unsigned G = t / 8; d = G % D; q = (G / D) * 8 + wave; }  // ... group of 8 adjacent tiles by group
    const unsigned jw = q % strips, b = q / strips;
    char *p = base + (size_t)d * table_bytes + (size_t)b * rows * pitch + (size_t)jw * 1024 + lane * 16;
    for (int r = 0; r < rows; r++) {
      if (NT) __builtin_nontemporal_store(val, (d2 *)(p + (size_t)r * pitch));
      else *(d2 *)(p + (size_t)r * pitch) = val;
      if (MODE == 2 && (r & 7) == 7) __syncthreads();
    }
  }
}

template <int MODE, int NT>
double run(char *buf, unsigned *ticket, size_t pitch, int strips, int blocks_v, int rows, size_t table_bytes, int D, int grid) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  double best = 1e9;
  for (int it = 0; it < 4; it++) {
    CHK(hipMemsetAsync(ticket, 0, 4, 0));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(512), 0, 0, buf, ticket, pitch, strips, blocks_v, rows, table_bytes, D);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const int D = 8, rows = 96, strips = 80, blocks_v = 52;           // 80 x 1 KB per row (a full row), 52 blocks of 96 rows
  const size_t pitch = 80 * 1024, table_bytes = (size_t)blocks_v * rows * pitch;   // ~0.8 GB span per table, 1/... written
  char *buf; unsigned *ticket;
  CHK(hipMalloc(&buf, table_bytes * D + (1 << 20)));
  CHK(hipMalloc(&ticket, 256));
  const double gb = (double)D * strips * blocks_v * rows * 1024 / 1e9;
  printf("bytes written per run: %.2f GB (tiles of %d rows x 1 KB, row pitch %zu, %d tables)\n", gb, rows, pitch, D);
  for (int grid : {256, 512}) {
    double t;
    t = run<0, 0>(buf, ticket, pitch, strips, blocks_v, rows, table_bytes, D, grid); printf("grid %d  per-wave tiles            : %.3f ms  %.2f TB/s\n", grid, t, gb / t);
    t = run<0, 1>(buf, ticket, pitch, strips, blocks_v, rows, table_bytes, D, grid); printf("grid %d  per-wave tiles, nt        : %.3f ms  %.2f TB/s\n", grid, t, gb / t);
    t = run<1, 0>(buf, ticket, pitch, strips, blocks_v, rows, table_bytes, D, grid); printf("grid %d  8 adjacent tiles per WG    : %.3f ms  %.2f TB/s\n", grid, t, gb / t);
    t = run<2, 0>(buf, ticket, pitch, strips, blocks_v, rows, table_bytes, D, grid); printf("grid %d  8 adjacent + barrier / 8 rows: %.3f ms  %.2f TB/s\n", grid, t, gb / t);
    t = run<3, 0>(buf, ticket, pitch, strips, blocks_v, rows, table_bytes, D, grid); printf("grid %d  linear                    : %.3f ms  %.2f TB/s (of %.2f GB)\n", grid, t, (double)table_bytes * D / 1e9 / t, (double)table_bytes * D / 1e9);
    t = run<3, 1>(buf, ticket, pitch, strips, blocks_v, rows, table_bytes, D, grid); printf("grid %d  linear, nt                : %.3f ms  %.2f TB/s\n", grid, t, (double)table_bytes * D / 1e9 / t);
  }
  return 0;
}

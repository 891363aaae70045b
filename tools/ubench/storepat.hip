// micro-benchmark (round 6): the store stream of the multi-table fills, flavour by flavour.
// A wave "owns" a tile of R = 48 consecutive table rows x one kilobyte (or two) of columns and stores it row by row,
// 16 bytes a lane -- exactly the tile workers' pattern (k_fill_hb: hb_store16; k_fill_pc's consumers): rows of a
// 10^4-column table are 80 KB apart (row pitch = roundup(len, 64) + 256 elements: 512-byte aligned bases with slack),
// so a tile is 48 pieces of 1 KB, 80 KB apart.  Flavours of global_store_dwordx4: plain, nt, sc1, sc0 sc1, sc0 nt.
// Shapes: A  one contiguous KB per instruction (C = 2 strips; each of the two instructions of a C = 4 row)
//         B  two rows' KB per lane pair of instructions interleaved (what a wave with two row groups in flight issues)
//         C  a full 2 KB of one row per wave (two instructions back to back on the same row)
//         D  4 KB per row per wave (4 instructions: what a 512-column tile would store)
// and the same with the waves' rows visited in order or the tiles taken from a ticket (as the kernels do).
// build: hipcc --offload-arch=gfx950 -O3 -o storepat storepat.hip ; run: ./storepat [tables=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int FL>
__device__ __forceinline__ void st16(char *p, d2 v) {
  if (FL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (FL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
}

// KBR: kilobytes of a row a wave stores (1, 2 or 4); WORK: dependent fp64 fma per row between the stores (0: pure stores)
template <int FL, int KBR, int WORK>
__global__ __launch_bounds__(512, 2) void k(char *base, unsigned *ticket, size_t pitch, int strips, int rows, unsigned total, size_t table_bytes, int tables) {
  const int lane = threadIdx.x & 63;
  d2 val = {1.0 + lane, 2.0};
  double w = 1.0 + lane * 1e-9;
  for (;;) {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= total) break;
    const unsigned d = t % (unsigned)tables, u = t / (unsigned)tables;
    const unsigned jw = u % (unsigned)strips, b = u / (unsigned)strips;
    char *p = base + (size_t)d * table_bytes + (size_t)b * rows * pitch + (size_t)jw * (1024 * KBR) + lane * 16;
    for (int r = 0; r < rows; r++) {
      char *q = p + (size_t)r * pitch;
#pragma unroll
      for (int i = 0; i < WORK; i++) w = fma(w, 1.0000001, 1e-9);
      if (WORK) val.x = w;
#pragma unroll
      for (int kb = 0; kb < KBR; kb++) st16<FL>(q + kb * 1024, val);
    }
  }
}

template <int FL, int KBR, int WORK>
double run(char *buf, unsigned *ticket, size_t pitch, int strips, int rows, unsigned total, int grid, size_t table_bytes, int tables) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  double best = 1e9;
  for (int it = 0; it < 4; it++) {
    CHK(hipMemsetAsync(ticket, 0, 4, 0));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<FL, KBR, WORK>), dim3(grid), dim3(512), 0, 0, buf, ticket, pitch, strips, rows, total, table_bytes, tables);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 0 && ms < best) best = ms;
  }
  return best;
}

template <int KBR, int WORK>
void sweep(char *buf, unsigned *ticket, int tables, int grid) {
  const int rows = 48, blocks = 104;                 // 104 blocks of 48 rows: 4992 rows of 80 KB per table
  const size_t pitch = 80 * 1024;
  const int strips = 80 / KBR;                       // the whole row is written: 80 KB
  const size_t table_bytes = (size_t)blocks * rows * pitch;
  const unsigned total = (unsigned)strips * blocks * tables;
  const double gb = (double)total * rows * 1024 * KBR / 1e9;
  const char *name[5] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 nt"};
  double t[5];
  t[0] = run<0, KBR, WORK>(buf, ticket, pitch, strips, rows, total, grid, table_bytes, tables);
  t[1] = run<1, KBR, WORK>(buf, ticket, pitch, strips, rows, total, grid, table_bytes, tables);
  t[2] = run<2, KBR, WORK>(buf, ticket, pitch, strips, rows, total, grid, table_bytes, tables);
  t[3] = run<3, KBR, WORK>(buf, ticket, pitch, strips, rows, total, grid, table_bytes, tables);
  t[4] = run<4, KBR, WORK>(buf, ticket, pitch, strips, rows, total, grid, table_bytes, tables);
  printf("%d tables (%.2f GB), grid %3d x 8 waves, %d KB of a row per wave, %2d fma a row:", tables, gb, grid, KBR, WORK);
  for (int f = 0; f < 5; f++) printf("  %s %.3f ms %.2f TB/s", name[f], t[f], gb / t[f]);
  printf("\n");
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int tables = argc > 1 ? atoi(argv[1]) : 8;
  const size_t table_bytes = (size_t)104 * 48 * 80 * 1024;
  char *buf; unsigned *ticket;
  CHK(hipMalloc(&buf, table_bytes * tables + (1 << 20)));
  CHK(hipMalloc(&ticket, 256));
  for (int grid : {256, 512}) {
    sweep<1, 0>(buf, ticket, tables, grid);
    sweep<2, 0>(buf, ticket, tables, grid);
    sweep<4, 0>(buf, ticket, tables, grid);
    sweep<1, 40>(buf, ticket, tables, grid);   // (about what a worker computes per 1 KB of a row: 128 cells x ~25 instructions / 64 lanes)
    sweep<2, 80>(buf, ticket, tables, grid);
  }
  return 0;
}

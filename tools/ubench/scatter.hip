// micro-benchmark: 10^6 scattered 4-byte updates over a slab of 33 MB / 205 MB (what k_count_cells of lists.hip does
// with the pairs of a group set): returning atomics, non-returning atomics, plain stores, plain loads, and the same
// with a second atomic per element on ~10^4 item counters.
// build: hipcc --offload-arch=gfx950 -O3 -o scatter scatter.hip ; run: ./scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// MODE 0 returning atomic (+ item atomic when the word was 0)   1 non-returning atomic   2 plain store   3 plain load
// 4 non-returning atomic + non-returning item atomic   5 returning atomic only   6 atomic OR of a bit (1 bit per cell)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *slab, unsigned long long elems, unsigned *icnt, unsigned nitems, unsigned long long G, unsigned *sink) {
  const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  const unsigned long long r = mix(g * 0x9E3779B97F4A7C15ull + 12345);
  const unsigned long long idx = r % elems;
  const unsigned item = (unsigned)(idx * nitems / elems);
  if (MODE == 0) {
    if (atomicAdd(&slab[idx], 1u) == 0u) atomicAdd(&icnt[item], 1u);
  } else if (MODE == 1) {
    __hip_atomic_fetch_add(&slab[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (MODE == 2) {
    slab[idx] = (unsigned)g + 1u;
  } else if (MODE == 3) {
    if (slab[idx] == 0xdeadbeefu) sink[0] = 1;
  } else if (MODE == 4) {
    __hip_atomic_fetch_add(&slab[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&icnt[item], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (MODE == 5) {
    if (atomicAdd(&slab[idx], 1u) == 0xdeadbeefu) sink[0] = 1;
  } else if (MODE == 6) {
    __hip_atomic_fetch_or(&slab[idx >> 5], 1u << (idx & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MODE>
void run(const char *what, unsigned *slab, unsigned long long elems, unsigned *icnt, unsigned nitems, unsigned long long G, unsigned *sink) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    CHK(hipMemsetAsync(slab, 0, elems * 4, 0));
    CHK(hipMemsetAsync(icnt, 0, (size_t)nitems * 4, 0));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE>), dim3((unsigned)((G + 255) / 256)), dim3(256), 0, 0, slab, elems, icnt, nitems, G, sink);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("  %-62s %8.1f us  %7.2f G updates/s\n", what, best * 1e3, G / best / 1e6);
}

int main() {
  unsigned *slab, *icnt, *sink;
  const unsigned long long G = 1000000;
  CHK(hipMalloc(&sink, 64));
  for (unsigned long long mb : {33ull, 205ull}) {
    const unsigned long long elems = mb * 1000000ull / 4;
    const unsigned nitems = mb == 33 ? 12600 : 10200;
    CHK(hipMalloc(&slab, elems * 4)); CHK(hipMalloc(&icnt, (size_t)nitems * 4));
    printf("slab of %llu MB, %llu scattered updates, %u item counters\n", mb, G, nitems);
    run<0>("returning atomic, + item atomic when the word was zero", slab, elems, icnt, nitems, G, sink);
    run<5>("returning atomic only", slab, elems, icnt, nitems, G, sink);
    run<1>("non-returning atomic", slab, elems, icnt, nitems, G, sink);
    run<4>("non-returning atomic + non-returning item atomic", slab, elems, icnt, nitems, G, sink);
    run<6>("non-returning atomic OR of one bit (a bit per cell)", slab, elems, icnt, nitems, G, sink);
    run<2>("plain store", slab, elems, icnt, nitems, G, sink);
    run<3>("plain load", slab, elems, icnt, nitems, G, sink);
    CHK(hipFree(slab)); CHK(hipFree(icnt));
  }
  return 0;
}

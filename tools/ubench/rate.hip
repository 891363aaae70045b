// micro-benchmark: issue rate of the instructions the tile workers are made of, with all four SIMDs of every
// compute unit busy (8 waves per workgroup, 256 workgroups) and enough independent chains per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o rate rate.hip ; run: ./rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int OP>
__global__ __launch_bounds__(512) void k(double *out, int iters) {
  double a[8];
  int n[8];
  for (int i = 0; i < 8; i++) { a[i] = 1.0 + threadIdx.x * 1e-3 + i; n[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = fma(a[i], 1.0000001, 1e-9);
      else if (OP == 1) { a[i] += (double)n[i]; n[i] += 3; }                     // cvt + add_f64 + add_i32
      else if (OP == 2) a[i] = ldexp(a[i], (it & 1) ? 1 : -1);
      else if (OP == 3) { n[i] = (n[i] >> 3) + 7; }                               // two integer ops
      else if (OP == 4) { a[i] = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(a[i]), 0x138, 0xf, 0xf, true),
                                                 __builtin_amdgcn_update_dpp(0, __double2loint(a[i]), 0x138, 0xf, 0xf, true)) + 1.0; }
      else if (OP == 5) { n[i] = __builtin_amdgcn_frexp_exp(a[i]) + n[i]; a[i] += 1.0; }
      else if (OP == 6) { a[i] = a[i] + 1.0; }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + n[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(double *out, int iters) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  double best = 1e9;
  for (int r = 0; r < 3; r++) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(512), 0, 0, out, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  double *out; CHK(hipMalloc(&out, 256 * 512 * 8));
  const int iters = 20000;
  const double ops = (double)iters * 8 * 2;   // wave-instructions of the group per SIMD (2 waves per SIMD)
  const char *names[] = {"fma_f64", "cvt_f64_i32 + add_f64 + add_i32", "ldexp_f64 (+cndmask)", "lshr + add (int)", "2 dpp movs + add_f64", "frexp_exp + add_i32 + add_f64", "add_f64"};
  double t[7];
  t[0] = run<0>(out, iters); t[1] = run<1>(out, iters); t[2] = run<2>(out, iters); t[3] = run<3>(out, iters);
  t[4] = run<4>(out, iters); t[5] = run<5>(out, iters); t[6] = run<6>(out, iters);
  for (int i = 0; i < 7; i++) printf("%-36s %.3f ms: %.1f ns per group per SIMD\n", names[i], t[i], t[i] * 1e6 / ops);
  return 0;
}

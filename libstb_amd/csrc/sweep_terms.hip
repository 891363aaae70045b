// sweep_terms.hip -- the reductions of the hyper-parameter posteriors and the small table utilities.
//
//   K3  k_sweep_partial   sum of S_S(n,t) over the (n,t) pairs of aterms    lib/samplea.c:68-80
//   K4  k_terms_partial   restaurant terms of aterms / lgamma sum of bterms  lib/samplea.c:65-67,
//                                                                            lib/sampleb.c:33-41
//       k_lookup          S_S semantics for a list of (n,m)                  lib/stable.c:941-974
//       k_to_float        S_FLOAT storage: narrow a slab                     lib/stable.h:31-33
// Sums are double-double per thread, then a fixed-shape tree: the same bits on every run.

#include "stb_common.h"

// ------------------------------------------------------------------------------------------------
// S_FLOAT storage: narrow a slab

__global__ __launch_bounds__(256) void k_to_float(const double *src, float *dst, uint64_t n2) {
  // two elements per thread: one 16-byte load, one 8-byte store
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n2; i += step) {
    const double2 v = reinterpret_cast<const double2 *>(src)[i];
    reinterpret_cast<float2 *>(dst)[i] = make_float2((float)v.x, (float)v.y);
  }
}

extern "C" int stb_table_to_float(const double *d_src, float *d_dst, uint64_t elems, void *stream) {
  STB_ENTRY;
  if (elems == 0) return 0;
  if (elems & 1) return stb_fail("stb_table_to_float: element count must be even (slabs are)");
  const uint64_t n2 = elems / 2;
  uint64_t blocks = (n2 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_to_float, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_src, d_dst, n2);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// lookups with S_S semantics

__device__ __forceinline__ double dev_S_S(const double *table, const double *S1, unsigned N,
                                          unsigned M, unsigned n, unsigned m) {
  // test order of lib/stable.c:941-974 for a table that cannot grow
  if (n == m) return 0.0;
  if (m == 1) return (n >= 1 && n <= N) ? S1[n - 1] : -HUGE_VAL;
  if (n < m || m == 0) return -HUGE_VAL;
  if (m > M || n > N) return -HUGE_VAL;
  return table[stb_row_offset(n, M) + (m - 2)];
}

__global__ void k_lookup(const double *table, const double *S1, unsigned N, unsigned M,
                         const uint32_t *n, const uint32_t *m, uint64_t G, double *out) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; g < G; g += step) out[g] = dev_S_S(table, S1, N, M, n[g], m[g]);
}

extern "C" int stb_lookup_S(const double *d_table, const double *d_S1, unsigned N, unsigned M,
                            const uint32_t *d_n, const uint32_t *d_m, uint64_t G, double *d_out,
                            void *stream) {
  STB_ENTRY;
  if (G == 0) return 0;
  uint64_t blocks = (G + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_lookup, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_table,
                     d_S1, N, M, d_n, d_m, G, d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- the ratio table's accessors on the device (lib/stable.c:875-939, no growth): what the table-indicator sampling step
// either side of the path reads for every customer (test/demo.c:405-445: S_V / S_U / S_UV after every S_remake).
// The V slab holds rows n = 2 .. N with m = 2 .. min(n, M) (stb_vrow_offset).  which: 0 V, 1 U, 2 UV.
__device__ __forceinline__ double dev_S_V(const double *vtable, unsigned N, unsigned M, unsigned n, unsigned m) {
  if (m < 2 || n < m || n > N || m > M || n < 2) return 0.0;  // (lib/stable.c:922, :929: "return 0")
  return vtable[stb_vrow_offset(n, M) + (m - 2)];
}
__global__ void k_lookup_V(const double *vtable, unsigned N, unsigned M, double a, int which, const uint32_t *n, const uint32_t *m, uint64_t G,
                           double *out) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; g < G; g += step) {
    const unsigned nn = n[g], mm = m[g];
    double r;
    if (which == 0) {
      r = dev_S_V(vtable, N, M, nn, mm);
    } else if (which == 1) {  // S_U, lib/stable.c:875-883 (m = 0 is the caller's error there: a NaN here)
      r = (mm == 1) ? (double)nn - a : (mm == 0 ? __longlong_as_double(0x7ff8000000000000ll) : (double)nn - (double)mm * a + 1.0 / dev_S_V(vtable, N, M, nn, mm));
    } else {                  // S_UV, lib/stable.c:885-897
      if (mm == 1) r = -HUGE_VAL;
      else if (mm == nn + 1) r = 1.0;
      else if (mm == nn) r = ((double)nn + 1.0) / ((double)nn - 1.0);
      else r = ((double)nn - (double)mm * a) * dev_S_V(vtable, N, M, nn, mm) + 1.0;
    }
    out[g] = r;
  }
}

static int lookup_v(const double *d_vtable, unsigned N, unsigned M, double a, int which, const uint32_t *d_n, const uint32_t *d_m, uint64_t G, double *d_out,
                    void *stream) {
  if (G == 0) return 0;
  if (!d_vtable || !d_n || !d_m || !d_out) return stb_fail("stb_lookup_V: null pointer");
  uint64_t blocks = (G + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_lookup_V, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_vtable, N, M, a, which, d_n, d_m, G, d_out);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int stb_lookup_V(const double *d_vtable, unsigned N, unsigned M, const uint32_t *d_n, const uint32_t *d_m, uint64_t G, double *d_out, void *stream) {
  STB_ENTRY;
  return lookup_v(d_vtable, N, M, 0.0, 0, d_n, d_m, G, d_out, stream);
}
extern "C" int stb_lookup_U(const double *d_vtable, unsigned N, unsigned M, double a, const uint32_t *d_n, const uint32_t *d_m, uint64_t G, double *d_out,
                            void *stream) {
  STB_ENTRY;
  return lookup_v(d_vtable, N, M, a, 1, d_n, d_m, G, d_out, stream);
}
extern "C" int stb_lookup_UV(const double *d_vtable, unsigned N, unsigned M, double a, const uint32_t *d_n, const uint32_t *d_m, uint64_t G, double *d_out,
                             void *stream) {
  STB_ENTRY;
  return lookup_v(d_vtable, N, M, a, 2, d_n, d_m, G, d_out, stream);
}

// second stage: out[d] = base[d] + sum_b partial[d][b], one block per d, fixed order
__global__ __launch_bounds__(256) void k_reduce_final(const dd_t *partial, int nb, double *out,
                                                      const double *base) {
  __shared__ dd_t lds[4];
  const int d = blockIdx.x;
  dd_t v{0.0, 0.0};
  for (int b = threadIdx.x; b < nb; b += 256) dd_merge(v, partial[(size_t)d * nb + b]);
  v = block_reduce_dd(v, lds);
  if (threadIdx.x == 0) {
    if (base) dd_add(v, base[d]);
    out[d] = v.hi + v.lo;
  }
}

// the same with the base values in the kernel arguments and the result written to pinned host memory as well: an
// evaluation is then two launches and ONE wait, no copy in either direction (sampleb's bterms)
struct base_args {
  double base[64];
};
__global__ __launch_bounds__(256) void k_reduce_final_host(const dd_t *partial, int nb, double *out, base_args B, double *out_host) {
  __shared__ dd_t lds[4];
  const int d = blockIdx.x;
  dd_t v{0.0, 0.0};
  for (int b = threadIdx.x; b < nb; b += 256) dd_merge(v, partial[(size_t)d * nb + b]);
  v = block_reduce_dd(v, lds);
  if (threadIdx.x == 0) {
    dd_add(v, B.base[d]);
    const double r = v.hi + v.lo;
    out[d] = r;
    out_host[d] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// K3: sweep.  Each block takes a contiguous chunk of pairs and DT discounts; the (n,t) pair is read
// once per DT tables, the row offset computed once, and DT gathers issued.

#define STB_SWEEP_DT 8
#define STB_SWEEP_CHUNK 4096
// pairs per workgroup: a few thousand pairs (what a fused evaluation leaves to the gather) go 256 to a
// workgroup, one look-up per thread, instead of three workgroups walking 16 look-ups each one after the other
__host__ __device__ static inline unsigned stb_sweep_chunk(uint64_t G) { return (G <= 65536) ? 256u : (unsigned)STB_SWEEP_CHUNK; }

__global__ __launch_bounds__(256) void k_sweep_partial(const double *tables, uint64_t tstride,
                                                       const double *S1, uint64_t s1stride, int D,
                                                       unsigned N, unsigned M, const uint32_t *n,
                                                       const uint16_t *t, uint64_t G, dd_t *partial,
                                                       int nb) {
  __shared__ dd_t lds[4];
  const int d0 = blockIdx.y * STB_SWEEP_DT;
  const uint64_t chunk = (uint64_t)stb_sweep_chunk(G);
  const uint64_t g0 = (uint64_t)blockIdx.x * chunk;
  const uint64_t g1 = (g0 + chunk < G) ? g0 + chunk : G;
  dd_t acc[STB_SWEEP_DT];
#pragma unroll
  for (int q = 0; q < STB_SWEEP_DT; q++) acc[q] = dd_t{0.0, 0.0};
  for (uint64_t g = g0 + threadIdx.x; g < g1; g += 256) {
    const unsigned nn = n[g], tt = t[g];
    if (nn <= 1) continue;  // lib/samplea.c:78: only n>1 contributes
    // classify once (lib/stable.c:944-949), then the gather differs per table only by base
    int kind;  // 0: zero, 1: S1, 2: -inf, 3: table
    uint64_t off = 0;
    if (nn == tt) kind = 0;
    else if (tt == 1) kind = (nn <= N) ? 1 : 2;
    else if (nn < tt || tt == 0) kind = 2;
    else if (tt > M || nn > N) kind = 2;
    else {
      kind = 3;
      off = stb_row_offset(nn, M) + (tt - 2);
    }
#pragma unroll
    for (int q = 0; q < STB_SWEEP_DT; q++) {
      const int d = d0 + q;
      if (d < D) {
        double v;
        if (kind == 3) v = tables[(uint64_t)d * tstride + off];
        else if (kind == 1) v = S1[(uint64_t)d * s1stride + nn - 1];
        else v = (kind == 0) ? 0.0 : -HUGE_VAL;
        dd_add(acc[q], v);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < STB_SWEEP_DT; q++) {
    dd_t r = block_reduce_dd(acc[q], lds);
    if (threadIdx.x == 0 && d0 + q < D) partial[(size_t)(d0 + q) * nb + blockIdx.x] = r;
  }
}

static int sweep_blocks(uint64_t G) { return (int)((G + stb_sweep_chunk(G) - 1) / stb_sweep_chunk(G)); }

extern "C" size_t stb_sweep_workspace_bytes(uint64_t G, int D) {
  int nb = sweep_blocks(G);
  if (nb < 256) nb = 256;  // (a subset of up to 65536 of the G pairs may be swept in the same workspace: 256 to a workgroup)
  return (size_t)D * nb * sizeof(dd_t) + 256;
}

extern "C" int stb_sweep_S(const double *d_tables, uint64_t table_stride, const double *d_S1,
                           uint64_t s1_stride, int D, unsigned N, unsigned M, const uint32_t *d_n,
                           const uint16_t *d_t, uint64_t G, double *d_out, void *d_ws,
                           size_t ws_bytes, void *stream) {
  STB_ENTRY;
  hipStream_t st = (hipStream_t)stream;
  if (D < 1) return stb_fail("stb_sweep_S: D=%d", D);
  if (ws_bytes < stb_sweep_workspace_bytes(G, D)) return stb_fail("stb_sweep_S: workspace too small");
  int nb = sweep_blocks(G);
  dd_t *partial = (dd_t *)d_ws;
  if (nb == 0) {
    HIPCHK(hipMemsetAsync(d_out, 0, sizeof(double) * D, st));
    return 0;
  }
  dim3 grid(nb, (D + STB_SWEEP_DT - 1) / STB_SWEEP_DT);
  hipLaunchKernelGGL(k_sweep_partial, grid, dim3(256), 0, st, d_tables, table_stride, d_S1,
                     s1_stride, D, N, M, d_n, d_t, G, partial, nb);
  hipLaunchKernelGGL(k_reduce_final, dim3(D), dim3(256), 0, st, partial, nb, d_out,
                     (const double *)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// K4: per-restaurant terms

#define STB_TERMS_CHUNK 2048
#define STB_TERMS_DMAX 64

struct terms_args {
  double x[STB_TERMS_DMAX];  // abscissae
  double p[STB_TERMS_DMAX];  // restaurant: log(x) ; bterms: lgamma(x/apar)
  double q[STB_TERMS_DMAX];  // bterms: x/apar
  int D;
  int mode;  // 0 restaurant (aterms head), 1 bterms
};

__global__ __launch_bounds__(256) void k_terms_partial(terms_args A, const uint32_t *T,
                                                       const double *bpar, uint64_t I, dd_t *partial,
                                                       int nb, double *a_out) {
  __shared__ dd_t lds[4];
  // (a fused evaluation's first launch also puts the abscissae where the table walk reads its discounts)
  if (a_out && blockIdx.x == 0 && threadIdx.x == 0) a_out[blockIdx.y] = A.x[blockIdx.y];
  const uint64_t i0 = (uint64_t)blockIdx.x * STB_TERMS_CHUNK;
  const uint64_t i1 = (i0 + STB_TERMS_CHUNK < I) ? i0 + STB_TERMS_CHUNK : I;
  const int d = blockIdx.y;
  dd_t acc{0.0, 0.0};
  if (A.mode == 0) {
    const double x = A.x[d], lx = A.p[d];
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += 256) {
#pragma clang fp contract(off)
      // lib/samplea.c:66-67: T*log(x) + lgamma(T + b/x) - lgamma(b/x), association as written
      const double Ti = (double)T[i];
      const double bx = bpar[i] / x;
      const double term = (Ti * lx + lgamma(Ti + bx)) - lgamma(bx);
      dd_add(acc, term);
    }
  } else {
    const double lg = A.p[d], xa = A.q[d];
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += 256)
      // lib/sampleb.c:38-39: lgamma(T + x/a) - lgamma(x/a)
      dd_add(acc, lgamma((double)T[i] + xa) - lg);
  }
  dd_t r = block_reduce_dd(acc, lds);
  if (threadIdx.x == 0) partial[(size_t)d * nb + blockIdx.x] = r;
}

// bterms over at most STB_TERMS_CHUNK restaurants (sampleb's usual case: a restaurant per document group): ONE launch --
// the partial sum of k_terms_partial's only block and the final step of k_reduce_final_host, same operations in the same
// order, so the same bits as the two-launch form; the value goes straight to pinned host memory
// (flag_host: a word of the same pinned memory that takes `seq` once the value is there -- the host of a one-abscissa
// evaluation spins on it instead of waiting for the stream: the launch's completion signal reaches it microseconds later)
__global__ __launch_bounds__(256) void k_bterms_one(terms_args A, base_args B, const uint32_t *T, uint64_t I, double *out, double *out_host,
                                                    double *flag_host, double seq) {
  __shared__ dd_t lds[4];
  const int d = blockIdx.x;
  const double lg = A.p[d], xa = A.q[d];
  dd_t acc{0.0, 0.0};
  for (uint64_t i = threadIdx.x; i < I; i += 256) dd_add(acc, lgamma((double)T[i] + xa) - lg);  // lib/sampleb.c:38-39
  const dd_t r = block_reduce_dd(acc, lds);
  if (threadIdx.x == 0) {
    dd_t v{0.0, 0.0};
    dd_merge(v, r);
    dd_add(v, B.base[d]);
    const double res = v.hi + v.lo;
    out[d] = res;
    out_host[d] = res;
    if (flag_host) {
      __threadfence_system();
      __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static int terms_blocks(uint64_t I) { return (int)((I + STB_TERMS_CHUNK - 1) / STB_TERMS_CHUNK); }

extern "C" size_t stb_terms_workspace_bytes(uint64_t I, int D) {
  int nb = terms_blocks(I);
  if (nb < 1) nb = 1;
  return (size_t)D * nb * sizeof(dd_t) + (size_t)D * sizeof(double) + 512;
}

static int run_terms(terms_args &A, const double *base_host, const uint32_t *d_T,
                     const double *d_bpar, uint64_t I, double *d_out, void *d_ws, size_t ws_bytes,
                     hipStream_t st) {
  const int D = A.D;
  if (ws_bytes < stb_terms_workspace_bytes(I, D)) return stb_fail("terms: workspace too small");
  int nb = terms_blocks(I);
  double *d_base = (double *)d_ws;
  dd_t *partial = (dd_t *)((char *)d_ws + stb_align_up((size_t)D * sizeof(double), 256));
  if (base_host)
    HIPCHK(hipMemcpyAsync(d_base, base_host, sizeof(double) * D, hipMemcpyHostToDevice, st));
  if (nb == 0) {
    if (base_host)
      HIPCHK(hipMemcpyAsync(d_out, d_base, sizeof(double) * D, hipMemcpyDeviceToDevice, st));
    else
      HIPCHK(hipMemsetAsync(d_out, 0, sizeof(double) * D, st));
    return 0;
  }
  hipLaunchKernelGGL(k_terms_partial, dim3(nb, D), dim3(256), 0, st, A, d_T, d_bpar, I, partial, nb, (double *)nullptr);
  hipLaunchKernelGGL(k_reduce_final, dim3(D), dim3(256), 0, st, partial, nb, d_out,
                     base_host ? (const double *)d_base : (const double *)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int stb_restaurant_terms(const double *x_host, int D, const uint32_t *d_T,
                                    const double *d_bpar, uint64_t I, double *d_out, void *d_ws,
                                    size_t ws_bytes, void *stream) {
  STB_ENTRY;
  if (D < 1 || D > STB_TERMS_DMAX) return stb_fail("stb_restaurant_terms: D=%d (max %d)", D, STB_TERMS_DMAX);
  terms_args A;
  memset(&A, 0, sizeof(A));
  A.D = D;
  A.mode = 0;
  for (int d = 0; d < D; d++) {
    if (!(x_host[d] > 0)) return stb_fail("stb_restaurant_terms: x=%g", x_host[d]);
    A.x[d] = x_host[d];
    A.p[d] = log(x_host[d]);  // host libm: one scalar per abscissa, same call as samplea.c:66
  }
  return run_terms(A, nullptr, d_T, d_bpar, I, d_out, d_ws, ws_bytes, (hipStream_t)stream);
}

// first launch of a fused aterms evaluation (groups.hip): the restaurant terms' partial sums -- reduced by that
// evaluation's last launch -- and the abscissae written to a_out[0..D) for the table walk; returns the partials
// and their number per abscissa
int stb_restaurant_partials(const double *x_host, int D, const uint32_t *d_T, const double *d_bpar, uint64_t I, void *d_ws,
                            size_t ws_bytes, double *a_out, const dd_t **partial_out, int *nb_out, hipStream_t st) {
  if (D < 1 || D > STB_TERMS_DMAX) return stb_fail("stb_groups_aterms: D=%d (max %d)", D, STB_TERMS_DMAX);
  if (ws_bytes < stb_terms_workspace_bytes(I, D)) return stb_fail("terms: workspace too small");
  terms_args A;
  memset(&A, 0, sizeof(A));
  A.D = D;
  A.mode = 0;
  for (int d = 0; d < D; d++) {
    if (!(x_host[d] > 0)) return stb_fail("stb_groups_aterms: x=%g", x_host[d]);
    A.x[d] = x_host[d];
    A.p[d] = log(x_host[d]);  // host libm: one scalar per abscissa, same call as samplea.c:66
  }
  int nb = terms_blocks(I);
  if (nb < 1) nb = 1;  // (no restaurants: one empty partial per abscissa, and the abscissae still go out)
  dd_t *partial = (dd_t *)((char *)d_ws + stb_align_up((size_t)D * sizeof(double), 256));
  hipLaunchKernelGGL(k_terms_partial, dim3(nb, D), dim3(256), 0, st, A, d_T, d_bpar, I, partial, nb, a_out);
  HIPCHK(hipGetLastError());
  *partial_out = partial;
  *nb_out = nb;
  return 0;
}

// ---- bterms for sampleb: T[] resident, an evaluation = two launches and one wait ----
struct stb_bctx {
  int dev;
  uint64_t I, cap;  // restaurants now, and at most
  uint32_t *d_T;
  double *d_out, *h_out, *h_out_dev;  // (h_out: STB_TERMS_DMAX values, then the flag word of k_bterms_one)
  double seq;
  void *d_ws;
  size_t ws_bytes;
  hipStream_t st;
};

extern "C" void stb_bterms_free(stb_bctx_t *c) {
  STB_ENTRY;
  if (!c) return;
  const int prev = stb_device_enter(c->dev);
  (void)hipStreamSynchronize(c->st);
  stb_pool_free(c->d_T);
  stb_pool_free(c->d_out);
  stb_pool_free(c->d_ws);
  stb_pool_free(c->h_out);
  stb_device_leave(prev);
  free(c);
}

extern "C" stb_bctx_t *stb_bterms_create(const uint32_t *T, int I) {
  STB_ENTRY;
  if (stb_device_count() < 1) {
    stb_fail("stb_bterms_create: no HIP device (libstb_amd has no CPU path)");
    return nullptr;
  }
  stb_bctx_t *c = (stb_bctx_t *)calloc(1, sizeof(*c));
  if (!c) return nullptr;
  const int prev = stb_device_enter(stb_get_device());
  bool ok = hipGetDevice(&c->dev) == hipSuccess;
  c->I = (uint64_t)(I > 0 ? I : 0);
  c->cap = c->I;
  c->ws_bytes = stb_terms_workspace_bytes(c->I, STB_TERMS_DMAX);
  // (the null stream, as the reference-shaped host samplers always used here: making a stream costs milliseconds, a
  // sampleb call a fraction of one)
  c->st = nullptr;
  ok = ok && stb_pool_malloc((void **)&c->d_T, sizeof(uint32_t) * (c->I ? c->I : 1)) == hipSuccess;
  ok = ok && stb_pool_malloc((void **)&c->d_out, sizeof(double) * STB_TERMS_DMAX) == hipSuccess;
  ok = ok && stb_pool_malloc(&c->d_ws, c->ws_bytes) == hipSuccess;
  ok = ok && stb_pool_malloc((void **)&c->h_out, sizeof(double) * (STB_TERMS_DMAX + 8), 1) == hipSuccess;
  ok = ok && hipHostGetDevicePointer((void **)&c->h_out_dev, c->h_out, 0) == hipSuccess;
  if (ok && c->I) ok = hipMemcpyAsync(c->d_T, T, sizeof(uint32_t) * c->I, hipMemcpyHostToDevice, c->st) == hipSuccess &&
                       hipStreamSynchronize(c->st) == hipSuccess;
  if (!ok) {
    const hipError_t e = hipGetLastError();
    stb_fail("stb_bterms_create: %s", e != hipSuccess ? hipGetErrorString(e) : "out of memory");
    stb_device_leave(prev);
    stb_bterms_free(c);
    return nullptr;
  }
  stb_device_leave(prev);
  return c;
}

// new totals for a context made for at least as many restaurants (a sampler keeps one context from call to call: a
// stream, pinned memory and device buffers cost more to make than twenty evaluations take); non-zero when I does not fit
extern "C" int stb_bterms_update(stb_bctx_t *c, const uint32_t *T, int I) {
  STB_ENTRY;
  if (!c) return stb_fail("stb_bterms_update: null context");
  if (I < 0 || (uint64_t)I > c->cap) return 1;
  const int prev = stb_device_enter(c->dev);
  int rc = 0;
  c->I = (uint64_t)I;
  if (I > 0 && (hipMemcpyAsync(c->d_T, T, sizeof(uint32_t) * (size_t)I, hipMemcpyHostToDevice, c->st) != hipSuccess ||
                hipStreamSynchronize(c->st) != hipSuccess))
    rc = stb_fail("stb_bterms_update: %s", hipGetErrorString(hipGetLastError()));
  stb_device_leave(prev);
  return rc;
}

// out_host[j] = bterms(x_j) (lib/sampleb.c:33-41), j < J <= 64; blocks until the values are there
extern "C" int stb_bterms_eval(stb_bctx_t *c, const double *x_host, int J, double Q, double shape, double apar, double *out_host) {
  STB_ENTRY;
  if (!c) return stb_fail("stb_bterms_eval: null context");
  if (J < 1 || J > STB_TERMS_DMAX) return stb_fail("stb_bterms_eval: J=%d (max %d)", J, STB_TERMS_DMAX);
  if (!(apar > 0)) return stb_fail("stb_bterms_eval: apar=%g", apar);
  terms_args A;
  base_args B;
  memset(&A, 0, sizeof(A));
  memset(&B, 0, sizeof(B));
  A.D = J;
  A.mode = 1;
  for (int j = 0; j < J; j++) {
    if (!(x_host[j] > 0)) return stb_fail("stb_bterms_eval: x=%g", x_host[j]);
    A.x[j] = x_host[j];
    A.q[j] = x_host[j] / apar;
    A.p[j] = lgamma(A.q[j]);                                     // lib/sampleb.c:36
    B.base[j] = -Q * x_host[j] + (shape - 1) * log(x_host[j]);   // lib/sampleb.c:37
  }
  const int prev = stb_device_enter(c->dev);
  int rc = 0;
  int nb = terms_blocks(c->I);
  if (nb < 1) nb = 1;
  dd_t *partial = (dd_t *)((char *)c->d_ws + stb_align_up((size_t)STB_TERMS_DMAX * sizeof(double), 256));
  bool spun = false;
  if (nb == 1 && stb_env_int("STB_BTERMS_ONE", 1)) {
    const bool spin = J == 1 && stb_env_int("STB_SPIN_WAIT", 1);
    volatile double *flag = c->h_out + STB_TERMS_DMAX;
    if (spin) {
      c->seq += 1.0;
      *flag = 0.0;
    }
    hipLaunchKernelGGL(k_bterms_one, dim3(J), dim3(256), 0, c->st, A, B, c->d_T, c->I, c->d_out, c->h_out_dev,
                       spin ? c->h_out_dev + STB_TERMS_DMAX : (double *)nullptr, c->seq);
    if (spin && hipGetLastError() == hipSuccess) {
      // (bounded: ~a millisecond of looks; whatever happens, the stream is waited for below when the word has not come)
      for (unsigned n = 0; n < 400000u && *flag != c->seq; n++) __builtin_ia32_pause();
      spun = *flag == c->seq;
    }
  } else {
    hipLaunchKernelGGL(k_terms_partial, dim3(nb, J), dim3(256), 0, c->st, A, c->d_T, (const double *)nullptr, c->I, partial, nb,
                       (double *)nullptr);
    hipLaunchKernelGGL(k_reduce_final_host, dim3(J), dim3(256), 0, c->st, partial, nb, c->d_out, B, c->h_out_dev);
  }
  if (spun) {
    out_host[0] = ((volatile double *)c->h_out)[0];
  } else if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->st) != hipSuccess)
    rc = stb_fail("stb_bterms_eval: %s", hipGetErrorString(hipGetLastError()));
  else
    for (int j = 0; j < J; j++) out_host[j] = c->h_out[j];
  stb_device_leave(prev);
  return rc;
}

extern "C" int stb_bterms(const double *x_host, int J, double Q, double shape, double apar,
                          const uint32_t *d_T, uint64_t I, double *d_out, void *d_ws,
                          size_t ws_bytes, void *stream) {
  STB_ENTRY;
  if (J < 1 || J > STB_TERMS_DMAX) return stb_fail("stb_bterms: J=%d (max %d)", J, STB_TERMS_DMAX);
  if (!(apar > 0)) return stb_fail("stb_bterms: apar=%g", apar);
  terms_args A;
  double base[STB_TERMS_DMAX];
  memset(&A, 0, sizeof(A));
  A.D = J;
  A.mode = 1;
  for (int j = 0; j < J; j++) {
    if (!(x_host[j] > 0)) return stb_fail("stb_bterms: x=%g", x_host[j]);
    A.x[j] = x_host[j];
    A.q[j] = x_host[j] / apar;
    A.p[j] = lgamma(A.q[j]);                                   // lib/sampleb.c:36
    base[j] = -Q * x_host[j] + (shape - 1) * log(x_host[j]);   // lib/sampleb.c:37
  }
  return run_terms(A, base, d_T, nullptr, I, d_out, d_ws, ws_bytes, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// aterms2: the S-free discount posterior (reference lib/samplea.c:85-150).  For a sampled partition of
// every restaurant's customers into tables, the posterior's table part is a sum over tables of
// log Gamma(size - x) / Gamma(1 - x) -- the reference's gcache_value(size - 1) with par = 1 - x
// (lib/lgamma.c:36-52) -- which depends on the partition only through HOW MANY tables have each
// size.  samplea2 builds that histogram once per call on the host; each evaluation is then one small
// kernel over the sizes that occur plus the restaurant terms of aterms.

// lib/lgamma.c:36-52 for par = 1 - x: log of the rising factorial par (par+1) ... (par+j-1)
__device__ __forceinline__ double gcache_term(int j, double par, double lgpar) {
#pragma clang fp contract(off)
  if (j <= 0) return 0.0;
  if (j == 1) return log(par);
  if (j == 2) return log(par * (par + 1));
  if (j == 3) return log(par * (par + 1) * (par + 2));
  return lgamma((double)j + par) - lgpar;
}

struct hist_args {
  double par[STB_TERMS_DMAX];    // 1 - x
  double lgpar[STB_TERMS_DMAX];  // lgamma(1 - x)
};

// partial[d][block] = sum over sizes s of cnt[s] * gcache_term(s - 1)
__global__ __launch_bounds__(256) void k_hist_partial(hist_args A, const uint32_t *cnt, unsigned S, dd_t *partial, int nb) {
  __shared__ dd_t lds[4];
  const int d = blockIdx.y;
  const double par = A.par[d], lgpar = A.lgpar[d];
  dd_t acc{0.0, 0.0};
  for (unsigned s = 2 + blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
    const uint32_t c = cnt[s];
    if (c) dd_add(acc, (double)c * gcache_term((int)s - 1, par, lgpar));
  }
  const dd_t r = block_reduce_dd(acc, lds);
  if (threadIdx.x == 0) partial[(size_t)d * nb + blockIdx.x] = r;
}

struct stb_hist {
  int dev;
  unsigned S;
  int I;
  uint32_t *d_cnt, *d_T;
  double *d_bpar, *d_out;  // d_out: [2][DMAX]
  void *d_ws;
  size_t ws_bytes;
  dd_t *d_partial;
  int nb;
  hipStream_t st;
};

extern "C" void stb_hist_free(stb_hist_t *h) {
  STB_ENTRY;
  if (!h) return;
  const int prev = stb_device_enter(h->dev);
  if (h->st) (void)hipStreamSynchronize(h->st);
  void *ptrs[] = {h->d_cnt, h->d_T, h->d_bpar, h->d_out, h->d_ws, h->d_partial};
  for (void *p : ptrs) stb_pool_free(p);
  if (h->st) (void)hipStreamDestroy(h->st);
  stb_device_leave(prev);
  free(h);
}

extern "C" stb_hist_t *stb_hist_create(const uint32_t *cnt, unsigned S, int I, const uint32_t *T, const double *bpar) {
  STB_ENTRY;
  if (stb_device_count() < 1) {
    stb_fail("stb_hist_create: no HIP device (libstb_amd has no CPU path)");
    return nullptr;
  }
  stb_hist_t *h = (stb_hist_t *)calloc(1, sizeof(*h));
  if (!h) return nullptr;
  const int prev = stb_device_enter(stb_get_device());
  bool ok = hipGetDevice(&h->dev) == hipSuccess;
  h->S = S;
  h->I = I;
  h->nb = (int)((S + 255) / 256);
  if (h->nb < 1) h->nb = 1;
  if (h->nb > 64) h->nb = 64;
  h->ws_bytes = stb_terms_workspace_bytes((uint64_t)(I > 0 ? I : 0), STB_TERMS_DMAX);
  ok = ok && hipStreamCreate(&h->st) == hipSuccess;
  ok = ok && stb_pool_malloc((void **)&h->d_cnt, sizeof(uint32_t) * (S ? S : 1)) == hipSuccess;
  ok = ok && stb_pool_malloc((void **)&h->d_T, sizeof(uint32_t) * (I > 0 ? I : 1)) == hipSuccess;
  ok = ok && stb_pool_malloc((void **)&h->d_bpar, sizeof(double) * (I > 0 ? I : 1)) == hipSuccess;
  ok = ok && stb_pool_malloc((void **)&h->d_out, sizeof(double) * 2 * STB_TERMS_DMAX) == hipSuccess;
  ok = ok && stb_pool_malloc(&h->d_ws, h->ws_bytes) == hipSuccess;
  ok = ok && stb_pool_malloc((void **)&h->d_partial, sizeof(dd_t) * STB_TERMS_DMAX * h->nb) == hipSuccess;
  if (ok && S) ok = hipMemcpy(h->d_cnt, cnt, sizeof(uint32_t) * S, hipMemcpyHostToDevice) == hipSuccess;
  if (ok && I > 0)
    ok = hipMemcpy(h->d_T, T, sizeof(uint32_t) * I, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(h->d_bpar, bpar, sizeof(double) * I, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    const hipError_t e = hipGetLastError();
    stb_fail("stb_hist_create: %s", e != hipSuccess ? hipGetErrorString(e) : "out of memory");
    stb_device_leave(prev);
    stb_hist_free(h);
    return nullptr;
  }
  stb_device_leave(prev);
  return h;
}

// out_host[d] = aterms2(x_d): restaurant terms + sum over table sizes of count * log rising factorial
extern "C" int stb_hist_aterms2(stb_hist_t *h, const double *x_host, int D, double *out_host) {
  STB_ENTRY;
  if (!h) return stb_fail("stb_hist_aterms2: null histogram");
  if (D < 1 || D > STB_TERMS_DMAX) return stb_fail("stb_hist_aterms2: D=%d (max %d)", D, STB_TERMS_DMAX);
  hist_args A;
  for (int d = 0; d < D; d++) {
    if (!(x_host[d] > 0 && x_host[d] < 1)) return stb_fail("stb_hist_aterms2: x=%g outside (0,1)", x_host[d]);
    A.par[d] = 1.0 - x_host[d];
    A.lgpar[d] = lgamma(A.par[d]);  // lib/lgamma.c:32
  }
  const int prev = stb_device_enter(h->dev);
  int rc = 0;
  double hh[2 * STB_TERMS_DMAX];
  do {
    hipLaunchKernelGGL(k_hist_partial, dim3(h->nb, D), dim3(256), 0, h->st, A, h->d_cnt, h->S, h->d_partial, h->nb);
    hipLaunchKernelGGL(k_reduce_final, dim3(D), dim3(256), 0, h->st, h->d_partial, h->nb, h->d_out, (const double *)nullptr);
    if ((rc = stb_restaurant_terms(x_host, D, h->d_T, h->d_bpar, (uint64_t)(h->I > 0 ? h->I : 0), h->d_out + STB_TERMS_DMAX,
                                   h->d_ws, h->ws_bytes, h->st)))
      break;
    if (hipMemcpyAsync(hh, h->d_out, sizeof(hh), hipMemcpyDeviceToHost, h->st) != hipSuccess ||
        hipStreamSynchronize(h->st) != hipSuccess) {
      rc = stb_fail("stb_hist_aterms2: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    for (int d = 0; d < D; d++) out_host[d] = hh[STB_TERMS_DMAX + d] + hh[d];
  } while (0);
  stb_device_leave(prev);
  return rc;
}

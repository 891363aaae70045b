// stb_kernels.hip -- gfx950 kernels and the additive C ABI (include/stb_hip.h) of libstb_amd.
//
// What runs here, and the reference code it replaces (paths relative to the reference tree):
//   K1/K2  table fill of S_remake_part, double S branch                    lib/stable.c:321-388
//            k_fill_chain   default: one launch, column blocks chained through edge granules
//            k_fill_pc      many tables: producer wave + consumer waves per block, per 128 rows
//            k_rec+k_logconv / k_fill_bfp / k_fill_rows   earlier forms, kept for ablation
//            k_fill_chainx  experimental: chain blocks + converter blocks in one launch
//          k_fillv_chain    V-table fill                                   lib/stable.c:451-482
//   K3     k_sweep_partial  the S_S gather-sum inside aterms               lib/samplea.c:68-80
//          k_fill_chain<.., DOT>  the same sum taken inside the fill (grids of discounts)
//   K4     k_terms_partial  restaurant terms of aterms / lgamma sum of bterms
//                                                       lib/samplea.c:65-67, lib/sampleb.c:33-41
//          k_lookup         S_S semantics for a list of (n,m)              lib/stable.c:941-974
//
// Design notes (DESIGN.md has the long form).  A table row depends only on the row above it, and
// inside a row information moves one column to the right per row (cell (n,m) reads (n-1,m) and
// (n-1,m-1)).  So the table is cut into column strips, one 64-lane wavefront per strip, every
// lane holding 1-4 adjacent columns in registers; inside a wave the left neighbour comes through
// one DPP wave shift per row.  The forms differ in how a strip gets the column to its left: the
// launch-per-row-block forms recompute a halo and publish the last row (the "frontier") at the
// kernel boundary; the chain form hands the edge column on, wave to wave, inside one launch.
//
// No CPU fallback exists in this file: without a device every entry point fails with a message.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

#include "stb_layout.h"
#include "../../include/stb_hip.h"

// ------------------------------------------------------------------------------------------------
// error plumbing

static thread_local char g_err[512] = "";

static int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" const char *stb_last_error(void) { return g_err; }

// The HIP runtime draws from libc's rand() while it initialises (first stream, first module load:
// observed on ROCm 7.2), which would shift the rand() stream the caller's ARMS sampler is about to
// consume (reference lib/arms.c:913-918) and make samplea/sampleb depend on whether the GPU had
// been touched before.  Every entry point that can reach the runtime therefore runs with rand()'s
// state swapped to a private buffer (glibc: rand() and random() share the state that
// initstate/setstate switch) and restores the caller's state on exit.
struct rand_guard {
  char buf[128];
  char *old;
  rand_guard() { old = initstate(0x5eedu, buf, sizeof buf); }
  ~rand_guard() { setstate(old); }
  rand_guard(const rand_guard &) = delete;
  rand_guard &operator=(const rand_guard &) = delete;
};
#define STB_ENTRY rand_guard stb_rand_guard_

extern "C" int stb_device_count(void) {
  STB_ENTRY;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int stb_device_name(char *buf, int len) {
  STB_ENTRY;
  hipDeviceProp_t p;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipGetDeviceProperties(&p, dev));
  snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// A small cache of device (and pinned host) buffers.  samplea builds and frees a group set (a dozen buffers, 70 MB at
// 10^6 pairs) on every call, as the reference builds and frees its table (lib/samplea.c:57-60,223);
// hipMalloc / hipFree cost ~1 ms each way there.  Freed buffers are kept (up to STB_POOL_MB,
// default 4096) and handed out again to requests of about the same size on the same device.
// Callers return a buffer only after the work that used it has completed.
struct pool_block {
  void *p;
  size_t bytes;
  int dev;
  int kind;  // 0: device memory, 1: pinned host memory
  bool used;
};
static std::mutex g_pool_mu;
static std::vector<pool_block> g_pool;
static size_t g_pool_idle = 0;

static hipError_t pool_raw_alloc(void **p, size_t bytes, int kind) {
  return kind ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
}
static void pool_raw_free(void *p, int kind) {
  if (kind) (void)hipHostFree(p);
  else (void)hipFree(p);
}

static hipError_t pool_malloc(void **out, size_t bytes, int kind = 0) {
  if (bytes == 0) bytes = 1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (size_t i = 0; i < g_pool.size(); i++) {
      const pool_block &b = g_pool[i];
      if (!b.used && b.dev == dev && b.kind == kind && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + 4096 &&
          (best < 0 || b.bytes < g_pool[best].bytes))
        best = (int)i;
    }
    if (best >= 0) {
      g_pool[best].used = true;
      g_pool_idle -= g_pool[best].bytes;
      *out = g_pool[best].p;
      return hipSuccess;
    }
  }
  void *p = nullptr;
  hipError_t e = pool_raw_alloc(&p, bytes, kind);
  if (e != hipSuccess) {
    // out of memory: give the idle buffers back and try once more
    std::vector<pool_block> drop;
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      for (size_t i = 0; i < g_pool.size();) {
        if (!g_pool[i].used) {
          drop.push_back(g_pool[i]);
          g_pool_idle -= g_pool[i].bytes;
          g_pool.erase(g_pool.begin() + i);
        } else {
          i++;
        }
      }
    }
    for (const pool_block &q : drop) pool_raw_free(q.p, q.kind);
    (void)hipGetLastError();
    e = pool_raw_alloc(&p, bytes, kind);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pool.push_back(pool_block{p, bytes, dev, kind, true});
  *out = p;
  return hipSuccess;
}

// returns false if p did not come from the pool
static bool pool_free(void *p) {
  if (!p) return true;
  static const size_t cap = (size_t)(getenv("STB_POOL_MB") ? atol(getenv("STB_POOL_MB")) : 4096) << 20;
  int release = -1;  // kind to release with, -1: kept or unknown
  bool known = false;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++) {
      if (g_pool[i].p == p) {
        known = true;
        if (g_pool_idle + g_pool[i].bytes <= cap) {
          g_pool[i].used = false;
          g_pool_idle += g_pool[i].bytes;
        } else {
          release = g_pool[i].kind;
          g_pool.erase(g_pool.begin() + i);
        }
        break;
      }
    }
  }
  if (release >= 0) pool_raw_free(p, release);
  return known;
}

extern "C" void stb_pool_trim(void) {
  std::vector<pool_block> drop;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size();) {
      if (!g_pool[i].used) {
        drop.push_back(g_pool[i]);
        g_pool.erase(g_pool.begin() + i);
      } else {
        i++;
      }
    }
    g_pool_idle = 0;
  }
  for (const pool_block &q : drop) pool_raw_free(q.p, q.kind);
}

extern "C" void *stb_device_malloc(size_t bytes) {
  STB_ENTRY;
  void *p = nullptr;
  hipError_t e = pool_malloc(&p, bytes ? bytes : 1, 0);
  if (e != hipSuccess) {
    fail("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
extern "C" void stb_device_free(void *p) {
  STB_ENTRY;
  if (p && !pool_free(p)) (void)hipFree(p);
}
extern "C" void *stb_host_malloc(size_t bytes) {
  STB_ENTRY;
  void *p = nullptr;
  hipError_t e = pool_malloc(&p, bytes ? bytes : 1, 1);
  if (e != hipSuccess) {
    fail("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
extern "C" void stb_host_free(void *p) {
  STB_ENTRY;
  if (p && !pool_free(p)) (void)hipHostFree(p);
}
extern "C" int stb_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream) {
  STB_ENTRY;
  HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}
extern "C" int stb_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) {
  STB_ENTRY;
  HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}
extern "C" int stb_stream_sync(void *stream) {
  STB_ENTRY;
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

extern "C" uint64_t stb_cells(unsigned N, unsigned M) { return stb_table_cells(N, M); }
extern "C" uint64_t stb_elems(unsigned N, unsigned M) { return stb_table_elems(N, M); }
extern "C" uint64_t stb_rowoff(unsigned n, unsigned M) { return stb_row_offset(n, M); }
extern "C" uint64_t stb_vcells(unsigned N, unsigned M) { return stb_vtable_cells(N, M); }
extern "C" uint64_t stb_velems(unsigned N, unsigned M) { return stb_vtable_elems(N, M); }
extern "C" uint64_t stb_vrowoff(unsigned n, unsigned M) { return stb_vrow_offset(n, M); }

// ------------------------------------------------------------------------------------------------
// cell arithmetic

// A table value S (not its log) as mant * 2^expo, mant in [0.5,1), or exact zero (mant 0, expo EZ).
// The recurrence  S^n_m = (n-1-m a) S^{n-1}_m + S^{n-1}_{m-1}  is then one fma plus exponent
// bookkeeping -- no transcendental on the dependent chain, and every step rounds once at 2^-53
// relative, which is tighter than the reference's log-domain step (one rounding at ulp(log S)).
#define STB_EZ (-(1 << 28))

struct cell_t {
  double m;
  int e;
};

__device__ __forceinline__ cell_t cell_zero() { return cell_t{0.0, STB_EZ}; }
__device__ __forceinline__ cell_t cell_one() { return cell_t{0.5, 1}; }

// coef * up + left
__device__ __forceinline__ cell_t cell_step(double coef, cell_t up, cell_t left) {
  int E = max(up.e, left.e);
  double x = ldexp(up.m, up.e - E);
  double y = ldexp(left.m, left.e - E);
  double r = fma(coef, x, y);
  cell_t o;
  o.m = __builtin_amdgcn_frexp_mant(r);
  o.e = E + __builtin_amdgcn_frexp_exp(r);
  return o;
}

__device__ __forceinline__ double cell_log(cell_t c) {
  // log(m 2^e) = e ln2 + log m ; ln2 split so that e*LN2_HI is exact for |e| < 2^20
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  double de = (double)c.e;
  return fma(de, LN2_HI, fma(de, LN2_LO, log(c.m)));
}

// log-domain variant: the reference's own cell update, same association order
// (lib/stable.c:95-103 logadd; :381-386 the two call sites).  rn intrinsics keep hipcc from
// contracting n - m*a into an fma the reference does not have.
__device__ __forceinline__ double ld_logadd(double V, double lp) {
#pragma clang fp contract(off)
  double hi = V, lo = lp;
  if (lp > V) {
    hi = lp;
    lo = V;
  }
  return hi + log(1.0 + exp(lo - hi));
}
// lib/stable.c:384-385: logadd(log(N-M*a-1.0) + S[N-1][M], S[N-1][M-1])
__device__ __forceinline__ double ld_cell(int n, int c, double a, double up, double left) {
#pragma clang fp contract(off)
  const double coef = ((double)n - (double)c * a) - 1.0;
  return ld_logadd(log(coef) + up, left);
}

// one lane shifted up by one across the whole 64-lane wave (lane l receives lane l-1's value,
// lane 0 receives `fill`): DPP wave_shr:1, no LDS traffic
__device__ __forceinline__ int wave_shr1(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}
// same with 0 shifted into lane 0 (bound_ctrl: no register has to be preset with the fill value)
__device__ __forceinline__ double wave_shr1_zero(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shr1(double v, double fill) {
  int lo = wave_shr1(__double2loint(v), __double2loint(fill));
  int hi = wave_shr1(__double2hiint(v), __double2hiint(fill));
  return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------
// K1/K2: fill

struct fill_args {
  const double *a;    // [D] discounts (device)
  double *tables;     // D slabs (S or V layout)
  uint64_t tstride;   // elements between slabs
  double *S1;         // D vectors of N (S modes only)
  uint64_t s1stride;
  double *fm;         // frontier mantissas / plain values: [D][2][W]
  int *fe;            // frontier exponents (scaled mode):  [D][2][W]
  unsigned W;         // frontier row pitch (>= M+2)
  unsigned N, M;
  int R;              // rows advanced per launch
  int H;              // halo columns (>= R, multiple of C)
  int Wv;             // owned columns per strip = 64*C - H
};

#define STB_MODE_SCALED 0  // S table, (mantissa, exponent) cells
#define STB_MODE_LOGDOM 1  // S table, log-domain cells in the reference's operation order
#define STB_MODE_VRATIO 2  // V table, plain doubles in the reference's operation order

// lib/stable.c:475-480: V^n_m = (1 + (m<n ? (n-1-m a) V^{n-1}_m : 0)) / (1/V^{n-1}_{m-1} + (n-1-(m-1)a)).
// Column 1 is carried as +inf so that 1/V^{n-1}_1 = 0 turns this into the m=2 form of :475.
__device__ __forceinline__ double v_cell(int n, int c, double a, double up, double left) {
#pragma clang fp contract(off)
  const double nm1 = (double)(n - 1);
  const double num = 1.0 + ((c < n) ? ((nm1 - (double)c * a) * up) : 0.0);
  const double den = 1.0 / left + (nm1 - (double)(c - 1) * a);
  return num / den;
}

// Columns are numbered from 1 (column 1 is S^n_1, the S1 vector; columns <= 0 are identically 0).
// Strip j owns columns [2 + j*Wv, 2 + (j+1)*Wv); its wave also carries H halo columns to the left,
// so lane l holds columns cs + l*C .. cs + l*C + C-1 with cs = 2 + j*Wv - H.  For strip 0 the
// "halo" is columns <= 1, which are exact (zeros and S1), so nothing is ever approximate.
// Launch k advances rows n0 = 2 + k*R .. n0 + R - 1 from the frontier (row n0 - 1).
template <int C, int MODE>
__global__ __launch_bounds__(64) void k_fill_rows(fill_args A, int k) {
  const int lane = threadIdx.x;
  const int j = blockIdx.x;
  const int d = blockIdx.y;
  const double a = A.a[d];
  const unsigned N = A.N, M = A.M;
  const int n0 = 2 + k * A.R;                                      // first row of this launch
  const int n1 = min((int)N, n0 + A.R - 1);                        // last row
  const int nf = n0 - 1;                                           // frontier row (already done)
  const int c0 = 2 + j * A.Wv - A.H + lane * C;                    // lane's first column
  const bool owned = lane * C >= A.H;                              // lane's columns are stored
  double *table = A.tables + (uint64_t)d * A.tstride;
  double *S1 = (MODE == STB_MODE_VRATIO) ? nullptr : A.S1 + (uint64_t)d * A.s1stride;
  const uint64_t fbase = ((uint64_t)d * 2) * A.W;
  const double *fm_in = A.fm + fbase + (uint64_t)(k & 1) * A.W;
  const int *fe_in = A.fe + fbase + (uint64_t)(k & 1) * A.W;
  double *fm_out = A.fm + fbase + (uint64_t)((k + 1) & 1) * A.W;
  int *fe_out = A.fe + fbase + (uint64_t)((k + 1) & 1) * A.W;

  // ---- state of row nf for my C columns ----
  cell_t st[C];
  const int cmax_f = min(nf, (int)M);  // columns above the diagonal of row nf are zero
#pragma unroll
  for (int i = 0; i < C; i++) {
    const int c = c0 + i;
    if (MODE == STB_MODE_LOGDOM) {
      st[i].e = 0;
      if (k == 0)
        st[i].m = (c == 1) ? 0.0 : -HUGE_VAL;  // row 1: log S^1_1 = 0
      else
        st[i].m = (c >= 1 && c <= cmax_f) ? fm_in[c] : -HUGE_VAL;
    } else if (MODE == STB_MODE_VRATIO) {
      st[i].e = 0;
      if (c == 1)
        st[i].m = HUGE_VAL;
      else if (k == 0)
        st[i].m = 0.0;
      else
        st[i].m = (c >= 2 && c <= cmax_f) ? fm_in[c] : 0.0;
    } else {
      if (k == 0)
        st[i] = (c == 1) ? cell_one() : cell_zero();
      else if (c >= 1 && c <= cmax_f)
        st[i] = cell_t{fm_in[c], fe_in[c]};
      else
        st[i] = cell_zero();
    }
  }
  if (MODE != STB_MODE_VRATIO && k == 0 && j == 0 && lane == 0) S1[0] = 0.0;  // log S^1_1

  for (int n = n0; n <= n1; n++) {
    // value of my left neighbour's last column in row n-1
    cell_t left;
    if (MODE == STB_MODE_SCALED) {
      left.m = wave_shr1(st[C - 1].m, 0.0);
      left.e = wave_shr1(st[C - 1].e, STB_EZ);
    } else {
      left.m = wave_shr1(st[C - 1].m, (MODE == STB_MODE_LOGDOM) ? -HUGE_VAL : 0.0);
      left.e = 0;
    }
    const double nm1 = (double)(n - 1);
#pragma unroll
    for (int i = C - 1; i >= 0; i--) {
      const int c = c0 + i;
      const cell_t lf = (i > 0) ? st[i - 1] : left;
      if (MODE == STB_MODE_LOGDOM) {
        double v;
        if (c >= n || c < 1)
          v = (c == n) ? 0.0 : -HUGE_VAL;  // S^n_n = 1; above the diagonal / left of column 1: 0
        else
          // the (M<N-1)?:0 case of the reference is covered: the diagonal state is exactly 0.0
          v = ld_cell(n, c, a, st[i].m, lf.m);
        st[i].m = v;
      } else if (MODE == STB_MODE_VRATIO) {
        double v;
        if (c == 1) v = HUGE_VAL;
        else if (c < 1 || c > n) v = 0.0;
        else v = v_cell(n, c, a, st[i].m, lf.m);
        st[i].m = v;
      } else {
        st[i] = cell_step(fma(-(double)c, a, nm1), st[i], lf);
      }
    }
    // ---- write row n ----
    // last stored column of row n: S keeps m<=n-1 (the diagonal is implicit), V keeps m<=n
    const int cmax = min((MODE == STB_MODE_VRATIO) ? n : n - 1, (int)M);
    if (owned) {
      const uint64_t roff = (MODE == STB_MODE_VRATIO) ? stb_vrow_offset((unsigned)n, M)
                                                      : stb_row_offset((unsigned)n, M);
      double *row = table + roff - 2;  // row[c] is column c
      double y[C];
#pragma unroll
      for (int i = 0; i < C; i++) y[i] = (MODE == STB_MODE_SCALED) ? cell_log(st[i]) : st[i].m;
      if (c0 + C - 1 <= cmax) {
        // whole lane inside the row: 16-byte stores (c0-2 is even and the row base is 16B aligned)
        if (C == 1) {
          row[c0] = y[0];
        } else {
#pragma unroll
          for (int i = 0; i < C; i += 2)
            *reinterpret_cast<double2 *>(row + c0 + i) = make_double2(y[i], y[i + 1]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < C; i++)
          if (c0 + i <= cmax) row[c0 + i] = y[i];
      }
    } else if (MODE != STB_MODE_VRATIO && j == 0) {
      // strip 0 only: the lane whose last column is column 1 emits S1[n-1] = log S^n_1
      if (c0 + C - 1 == 1)
        S1[n - 1] = (MODE == STB_MODE_SCALED) ? cell_log(st[C - 1]) : st[C - 1].m;
    }
  }

  // ---- publish row n1 for the next launch: owned columns, plus column 1 from strip 0 ----
  if (n1 < (int)N) {
#pragma unroll
    for (int i = 0; i < C; i++) {
      const int c = c0 + i;
      const bool mine = owned || (j == 0 && c == 1);
      if (mine && c >= 1 && c <= (int)M) {
        fm_out[c] = st[i].m;
        if (MODE == STB_MODE_SCALED) fe_out[c] = st[i].e;
      }
    }
  }
}

// ---- block-floating variant of the scaled fill (the default) --------------------------------
//
// Same recurrence in the linear domain, but a cell is (v, ep) with true value v * 2^ep where ep is
// FROZEN for P consecutive rows: inside a period the update is  v <- (n-1 - c a) v + v_left * s,
// s = 2^(ep_left - ep) fixed per period, i.e. one add, one multiply and one fma per cell and row.
// v starts each period at 2^-BFP_BIAS * [0.5,1) and grows by at most N^2 per row (the left
// neighbour can be that much larger next to the diagonal), so P rows with P * (2 log2 N + 1) <= 1700
// bits of the ~1900 available never overflow; at the period end every cell is renormalised.
// The log that is stored is taken from the bits of v: exponent field + 7 leading mantissa bits
// index a 128-entry table {1/c, -log(1/c)} held in LDS, then a degree-5 polynomial in
// r = z/c - 1, |r| < 2^-8 (the construction used by table-driven libm logs).  Absolute error of
// the log is a few 1e-16, far inside the 1e-10 parity bound.
#define BFP_BIAS 900

__device__ double2 g_logtab[128];  // {invc, logc}; written by the host once per device

__device__ __forceinline__ double bfp_log(double v, int ep, const double2 *lt) {
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const int hi = __double2hiint(v), lo = __double2loint(v);
  const int kexp = ((hi >> 20) & 0x7ff) - 1023;
  const int idx = (hi >> 13) & 127;
  const double z = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);  // [1,2)
  const double2 t = lt[idx];
  const double r = fma(z, t.x, -1.0);
  double p = fma(r, 0.2, -0.25);  // r^6/6 <= 2^-48/6 = 6e-16 is dropped
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -0.5);
  p = fma(r, p, 1.0);
  const double kf = (double)(kexp + ep);
  return fma(kf, LN2_HI, fma(kf, LN2_LO, fma(r, p, t.y)));
}

// keep a value live at this point of the instruction stream (stops hipcc from sinking the log
// into the divergent store branches, which would serialise it behind the recurrence step)
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }

// emit one finished row: S1 from the lane that holds column 1 of strip 0, table values from owned
// lanes, 16 bytes per lane when the whole lane lies inside the row
template <int C>
__device__ __forceinline__ void bfp_store_row(const double (&y)[C], double *row, double *S1, int n,
                                              int c0, int cmax, bool owned, bool s1lane) {
  if (owned) {
    if (c0 + C - 1 <= cmax) {
      if (C == 1) {
        row[c0] = y[0];
      } else {
#pragma unroll
        for (int i = 0; i < C; i += 2)
          *reinterpret_cast<double2 *>(row + c0 + i) = make_double2(y[i], y[i + 1]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < C; i++)
        if (c0 + i <= cmax) row[c0 + i] = y[i];
    }
  } else if (s1lane) {
    S1[n - 1] = y[C - 1];
  }
}

// U consecutive rows for this lane's C columns.  The U recurrence steps are a short dependent
// chain (DPP shift, multiply, fma).  The U*C logs that follow are written stage-major -- all table
// reads, then stage 1 of every polynomial, then stage 2, ... -- so that the in-order wave always has
// an independent instruction to issue while a previous fma or LDS read is still in flight (there
// are only one or two waves per SIMD when few tables are being filled, so instruction-level
// parallelism is what hides latency here, not occupancy).
template <int C, int U>
__device__ __forceinline__ void bfp_rows(double (&v)[C], const double (&ca)[C], const double (&s)[C],
                                         const int (&ep)[C], int n, const double2 *lt, double *table,
                                         uint64_t &roff, double *S1, unsigned M, int c0, int clast,
                                         bool owned, bool s1lane) {
  constexpr int Q = U * C;
  double x[Q];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const double lfv = wave_shr1(v[C - 1], 0.0);
    const double nm1 = (double)(n + u - 1);
#pragma unroll
    for (int i = C - 1; i >= 0; i--) {
      const double lf = (i > 0) ? v[i - 1] : lfv;
      v[i] = fma(nm1 - ca[i], v[i], lf * s[i]);
    }
#pragma unroll
    for (int i = 0; i < C; i++) x[u * C + i] = v[i];
  }
  // ---- y = log(x 2^ep): exponent field + 7 mantissa bits -> table, degree-5 polynomial in r ----
  double2 t[Q];
  double z[Q], kf[Q], r[Q], pl[Q], y[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) t[q] = lt[(__double2hiint(x[q]) >> 13) & 127];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const int hi = __double2hiint(x[q]);
    z[q] = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(x[q]));
    kf[q] = (double)((int)((hi >> 20) & 0x7ff) - 1023 + ep[q % C]);
  }
#pragma unroll
  for (int q = 0; q < Q; q++) r[q] = fma(z[q], t[q].x, -1.0);
#pragma unroll
  for (int q = 0; q < Q; q++) pl[q] = fma(r[q], 0.2, -0.25);  // r^6/6 <= 6e-16 is dropped
#pragma unroll
  for (int q = 0; q < Q; q++) pl[q] = fma(r[q], pl[q], 1.0 / 3.0);
#pragma unroll
  for (int q = 0; q < Q; q++) pl[q] = fma(r[q], pl[q], -0.5);
#pragma unroll
  for (int q = 0; q < Q; q++) pl[q] = fma(r[q], pl[q], 1.0);
#pragma unroll
  for (int q = 0; q < Q; q++) y[q] = fma(kf[q], 0.693147180559945309417, fma(r[q], pl[q], t[q].y));
#pragma unroll
  for (int q = 0; q < Q; q++) pin(y[q]);

  // ---- emit the U rows: no column tests, the row slack of the slab layout (stb_layout.h) absorbs
  // whatever a wave holds beyond the diagonal or beyond column M ----
  (void)clast;
#pragma unroll
  for (int u = 0; u < U; u++) {
    if (owned) {
      double *row = table + roff - 2;
      if (C == 1) {
        row[c0] = y[u];
      } else {
#pragma unroll
        for (int i = 0; i < C; i += 2)
          *reinterpret_cast<double2 *>(row + c0 + i) = make_double2(y[u * C + i], y[u * C + i + 1]);
      }
    } else if (s1lane) {
      S1[n + u - 1] = y[u * C + C - 1];
    }
    roff += stb_row_pitch((unsigned)(n + u), M);
  }
}

template <int C>
__global__ __launch_bounds__(64) void k_fill_bfp(fill_args A, int k, int P) {
  __shared__ double2 lt[128];
  const int lane = threadIdx.x;
  lt[lane] = g_logtab[lane];
  lt[lane + 64] = g_logtab[lane + 64];
  __syncthreads();

  const int j = blockIdx.x;
  const int d = blockIdx.y;
  const double a = A.a[d];
  const unsigned N = A.N, M = A.M;
  const int n0 = 2 + k * A.R;
  const int n1 = min((int)N, n0 + A.R - 1);
  const int nf = n0 - 1;
  const int c0 = 2 + j * A.Wv - A.H + lane * C;
  const bool owned = lane * C >= A.H;
  const bool s1lane = (j == 0) && (c0 + C - 1 == 1);
  const int clast = 2 + j * A.Wv - A.H + 64 * C - 1;  // last column carried by this wave
  const int cmin = 2 + j * A.Wv - A.H;                // first column carried by this wave
  double *table = A.tables + (uint64_t)d * A.tstride;
  double *S1 = A.S1 + (uint64_t)d * A.s1stride;
  const uint64_t fbase = ((uint64_t)d * 2) * A.W;
  const double *fm_in = A.fm + fbase + (uint64_t)(k & 1) * A.W;
  const int *fe_in = A.fe + fbase + (uint64_t)(k & 1) * A.W;
  double *fm_out = A.fm + fbase + (uint64_t)((k + 1) & 1) * A.W;
  int *fe_out = A.fe + fbase + (uint64_t)((k + 1) & 1) * A.W;

  // ---- row nf: (mantissa, exponent) from the frontier -> (v, ep) ----
  double v[C], ca[C];
  int ep[C];
  const int cmax_f = min(nf, (int)M);
#pragma unroll
  for (int i = 0; i < C; i++) {
    const int c = c0 + i;
    double m = 0.0;
    int e = 1;
    if (k == 0) {
      if (c == 1) m = 0.5;  // S^1_1 = 1 = 0.5 * 2^1
    } else if (c >= 1 && c <= cmax_f) {
      m = fm_in[c];
      e = fe_in[c];
    }
    v[i] = ldexp(m, -BFP_BIAS);
    ep[i] = e + BFP_BIAS;
    ca[i] = (double)c * a;
  }
  if (k == 0 && j == 0 && lane == 0) S1[0] = 0.0;  // log S^1_1

  uint64_t roff = stb_row_offset((unsigned)n0, M);  // element offset of the next row to emit

  for (int nb = n0; nb <= n1; nb += P) {
    const int ne = min(n1, nb + P - 1);
    // ---- period set-up: freeze exponents, derive the per-cell scale of the left input ----
    double s[C];
    {
      // a cell far below its left neighbour (or an exact zero) adopts the neighbour's exponent so
      // that s stays <= 2^64
      int epl = wave_shr1(ep[C - 1], ep[0]);
#pragma unroll
      for (int i = 0; i < C; i++) {
        const int el = (i > 0) ? ep[i - 1] : epl;
        const int mine = ep[i];
        if (el > mine + 64 || v[i] == 0.0) {
          v[i] = ldexp(v[i], mine - el);
          ep[i] = el;
        }
      }
      epl = wave_shr1(ep[C - 1], ep[0]);
#pragma unroll
      for (int i = 0; i < C; i++) {
        const int el = (i > 0) ? ep[i - 1] : epl;
        s[i] = ldexp(1.0, min(max(el - ep[i], -1100), 100));
      }
    }
    // ---- the rows of this period, four at a time.  Rows above this wave's first column are
    // identically zero (and row 2 stores nothing): skip them, which also keeps every store inside
    // its own row's slack ----
    int n = max(nb, max(cmin, 3));
    if (nb == 2 && cmin < 3) {
      // row 2 of strip 0: advance the state, emit only S1
      const double lfv = wave_shr1(v[C - 1], 0.0);
#pragma unroll
      for (int i = C - 1; i >= 0; i--) {
        const double lf = (i > 0) ? v[i - 1] : lfv;
        v[i] = fma(1.0 - ca[i], v[i], lf * s[i]);
      }
      if (s1lane) S1[1] = bfp_log(v[C - 1], ep[C - 1], lt);
    }
    roff = stb_row_offset((unsigned)n, M);
    for (; n + 3 <= ne; n += 4)
      bfp_rows<C, 4>(v, ca, s, ep, n, lt, table, roff, S1, M, c0, clast, owned, s1lane);
    for (; n <= ne; n++)
      bfp_rows<C, 1>(v, ca, s, ep, n, lt, table, roff, S1, M, c0, clast, owned, s1lane);
    // ---- renormalise ----
#pragma unroll
    for (int i = 0; i < C; i++) {
      const int kx = __builtin_amdgcn_frexp_exp(v[i]);
      const double m = __builtin_amdgcn_frexp_mant(v[i]);
      if (v[i] != 0.0) {
        v[i] = ldexp(m, -BFP_BIAS);
        ep[i] += kx + BFP_BIAS;
      }
    }
  }

  if (n1 < (int)N) {
#pragma unroll
    for (int i = 0; i < C; i++) {
      const int c = c0 + i;
      const bool mine = owned || (j == 0 && c == 1);
      if (mine && c >= 1 && c <= (int)M) {
        // back to (mantissa in [0.5,1), exponent): v = m 2^-BIAS, value = v 2^ep
        fm_out[c] = ldexp(v[i], BFP_BIAS);
        fe_out[c] = (v[i] != 0.0) ? ep[i] - BFP_BIAS : STB_EZ;
      }
    }
  }
}

// ---- split form of the block-floating fill -----------------------------------------------------
//
// With one or a few tables per GPU the fused kernel above is bound by the length of one row's
// dependent chain (shift, fma, LDS lookup, nine fmas, store) times N rows.  Here the chain is cut:
// k_rec carries ONLY the recurrence (per cell and row: one add, one multiply, one fma) and stores
// the raw block-floating significand v where the log will eventually live, plus one exponent per
// cell and renormalisation period in a small ring; k_logconv then turns v into log(v 2^e) in place,
// one thread per two cells, on other CUs and on another stream while the recurrence moves on.
// The serial part per row is three dependent fp64 operations; the logs are embarrassingly parallel.
#define STB_EP_RING 32  // row-blocks of exponents kept alive for the conversion kernels

struct split_args {
  int *epbuf;      // [D][STB_EP_RING][PPL][W] exponents frozen per period
  int PPL;         // periods per launch
  int P;           // rows per period
#ifdef STB_STAMPS
  unsigned long long *stamps;  // diagnostic build only: [launch][strip][8] s_memtime stamps
  int stamp_strips;
#endif
};

#ifdef STB_STAMPS
#define STAMP(slot)                                                                              \
  do {                                                                                           \
    if (X.stamps && lane == 0 && d == 0 && j < X.stamp_strips)                                   \
      X.stamps[((size_t)k * X.stamp_strips + j) * 8 + (slot)] = __builtin_amdgcn_s_memtime();  \
  } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

template <int C>
__global__ __launch_bounds__(64) void k_rec(fill_args A, split_args X, int k) {
  const int lane = threadIdx.x;
  const int j = blockIdx.x;
  const int d = blockIdx.y;
  const double a = A.a[d];
  const unsigned N = A.N, M = A.M;
  const int P = X.P;
  const int n0 = 2 + k * A.R;
  const int n1 = min((int)N, n0 + A.R - 1);
  const int nf = n0 - 1;
  const int c0 = 2 + j * A.Wv - A.H + lane * C;
  const bool owned = lane * C >= A.H;
  const int cmin = 2 + j * A.Wv - A.H;  // first column carried by this wave
  double *table = A.tables + (uint64_t)d * A.tstride;
  const uint64_t fbase = ((uint64_t)d * 2) * A.W;
  const double *fm_in = A.fm + fbase + (uint64_t)(k & 1) * A.W;
  const int *fe_in = A.fe + fbase + (uint64_t)(k & 1) * A.W;
  double *fm_out = A.fm + fbase + (uint64_t)((k + 1) & 1) * A.W;
  int *fe_out = A.fe + fbase + (uint64_t)((k + 1) & 1) * A.W;
  int *epslot = X.epbuf + ((uint64_t)d * STB_EP_RING + (uint64_t)(k % STB_EP_RING)) * X.PPL * A.W;

  STAMP(0);
  double v[C], ca[C];
  int ep[C];
  const int cmax_f = min(nf, (int)M);
#pragma unroll
  for (int i = 0; i < C; i++) {
    const int c = c0 + i;
    double m = 0.0;
    int e = 1;
    if (k == 0) {
      if (c == 1) m = 0.5;
    } else if (c >= 1 && c <= cmax_f) {
      m = fm_in[c];
      e = fe_in[c];
    }
    v[i] = ldexp(m, -BFP_BIAS);
    ep[i] = e + BFP_BIAS;
    ca[i] = (double)c * a;
  }
  STAMP(1);

  int pidx = 0;
  for (int nb = n0; nb <= n1; nb += P, pidx++) {
    const int ne = min(n1, nb + P - 1);
    double s[C];
    {
      int epl = wave_shr1(ep[C - 1], ep[0]);
#pragma unroll
      for (int i = 0; i < C; i++) {
        const int el = (i > 0) ? ep[i - 1] : epl;
        const int mine = ep[i];
        if (el > mine + 64 || v[i] == 0.0) {
          v[i] = ldexp(v[i], mine - el);
          ep[i] = el;
        }
      }
      epl = wave_shr1(ep[C - 1], ep[0]);
#pragma unroll
      for (int i = 0; i < C; i++) {
        const int el = (i > 0) ? ep[i - 1] : epl;
        s[i] = ldexp(1.0, min(max(el - ep[i], -1100), 100));
      }
      // exponents of this period, for the conversion kernel
      int *epp = epslot + (uint64_t)pidx * A.W;
#pragma unroll
      for (int i = 0; i < C; i++) {
        const int c = c0 + i;
        if (owned && c <= (int)M) epp[c] = ep[i];
      }
    }
    // A lone wave issues roughly one instruction per 5 cycles whatever its kind, and a launch
    // lasts as long as its slowest wave, so the row loop is kept to the bare recurrence: shift,
    // C x (multiply, fma, add), one store, one pointer bump.  No column tests: the slab's row
    // slack (stb_layout.h) absorbs what the wave holds beyond the diagonal or beyond column M.
    STAMP(2);
    // Rows above this wave's first column are identically zero and row 2 stores nothing: start at
    // ns (this also keeps every store inside its own row's slack).
    const int ns = max(nb, max(cmin, 3));
    if (nb == 2 && cmin < 3) {
      double t[C];
      t[0] = wave_shr1_zero(v[C - 1]) * s[0];
#pragma unroll
      for (int i = 1; i < C; i++) t[i] = v[i - 1] * s[i];
#pragma unroll
      for (int i = 0; i < C; i++) v[i] = fma(1.0 - ca[i], v[i], t[i]);
    }
    double coef[C];
#pragma unroll
    for (int i = 0; i < C; i++) coef[i] = (double)(ns - 1) - ca[i];
    double *rowp = table + stb_row_offset((unsigned)ns, M) - 2 + c0;  // my first column in row ns
    const unsigned pitch = stb_row_pitch((unsigned)ns, M);
    if (ns > ne) {
      // nothing to do in this period
    } else if (stb_row_pitch((unsigned)ne, M) == pitch) {
      for (int n = ns; n <= ne; n++) {
        double t[C];
        t[0] = wave_shr1_zero(v[C - 1]) * s[0];
#pragma unroll
        for (int i = 1; i < C; i++) t[i] = v[i - 1] * s[i];
#pragma unroll
        for (int i = 0; i < C; i++) {
          v[i] = fma(coef[i], v[i], t[i]);
          coef[i] += 1.0;
        }
        if (owned) {
          if (C == 1) {
            rowp[0] = v[0];
          } else {
#pragma unroll
            for (int i = 0; i < C; i += 2)
              *reinterpret_cast<double2 *>(rowp + i) = make_double2(v[i], v[i + 1]);
          }
        }
        rowp += pitch;
      }
    } else {
      // the row pitch steps up inside this period (once per 64 rows in the triangular part)
      for (int n = ns; n <= ne; n++) {
        double t[C];
        t[0] = wave_shr1_zero(v[C - 1]) * s[0];
#pragma unroll
        for (int i = 1; i < C; i++) t[i] = v[i - 1] * s[i];
#pragma unroll
        for (int i = 0; i < C; i++) {
          v[i] = fma(coef[i], v[i], t[i]);
          coef[i] += 1.0;
        }
        if (owned) {
          if (C == 1) {
            rowp[0] = v[0];
          } else {
#pragma unroll
            for (int i = 0; i < C; i += 2)
              *reinterpret_cast<double2 *>(rowp + i) = make_double2(v[i], v[i + 1]);
          }
        }
        rowp += stb_row_pitch((unsigned)n, M);
      }
    }
    STAMP(3);
#pragma unroll
    for (int i = 0; i < C; i++) {
      const int kx = __builtin_amdgcn_frexp_exp(v[i]);
      const double m = __builtin_amdgcn_frexp_mant(v[i]);
      if (v[i] != 0.0) {
        v[i] = ldexp(m, -BFP_BIAS);
        ep[i] += kx + BFP_BIAS;
      }
    }
  }

  if (n1 < (int)N) {
#pragma unroll
    for (int i = 0; i < C; i++) {
      const int c = c0 + i;
      const bool mine = owned || (j == 0 && c == 1);
      if (mine && c >= 1 && c <= (int)M) {
        fm_out[c] = ldexp(v[i], BFP_BIAS);
        fe_out[c] = (v[i] != 0.0) ? ep[i] - BFP_BIAS : STB_EZ;
      }
    }
  }
  STAMP(4);
}

// rows [ra, rb] of every table: raw significands -> logs, in place.  grid = (column chunks of 512,
// rb-ra+1 rows, D tables); a thread converts two adjacent columns (one 16-byte load and store).
__global__ __launch_bounds__(256) void k_logconv(fill_args A, split_args X, int ra, int rb) {
  __shared__ double2 lt[128];
  if (threadIdx.x < 128) lt[threadIdx.x] = g_logtab[threadIdx.x];
  __syncthreads();
  const int n = ra + blockIdx.y;
  const int d = blockIdx.z;
  if (n > rb) return;
  const unsigned M = A.M;
  const int k = (n - 2) / A.R;
  const int pidx = ((n - 2) % A.R) / X.P;
  const int *epp = X.epbuf + (((uint64_t)d * STB_EP_RING + (uint64_t)(k % STB_EP_RING)) * X.PPL + pidx) * A.W;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // log S^n_1 = log Gamma(n-a)/Gamma(1-a) in closed form (the recurrence kernel carries column 1
    // only as input to column 2); row 1 is written here too when the group starts at row 2
    double *S1 = A.S1 + (uint64_t)d * A.s1stride;
    const double a = A.a[d];
    S1[n - 1] = lgamma((double)n - a) - lgamma(1.0 - a);
    if (n == 2) S1[0] = 0.0;
  }
  const int cmax = min(n - 1, (int)M);
  const int c = 2 + 2 * (blockIdx.x * 256 + threadIdx.x);
  if (c > cmax) return;
  double *row = A.tables + (uint64_t)d * A.tstride + stb_row_offset((unsigned)n, M) - 2;
  if (c + 1 <= cmax) {
    double2 x = *reinterpret_cast<double2 *>(row + c);
    const int2 e = *reinterpret_cast<const int2 *>(epp + c);
    x.x = bfp_log(x.x, e.x, lt);
    x.y = bfp_log(x.y, e.y, lt);
    *reinterpret_cast<double2 *>(row + c) = x;
  } else {
    row[c] = bfp_log(row[c], epp[c], lt);
  }
}

// ---- producer/consumer form of the block-floating fill ---------------------------------------
//
// One workgroup = one column block of 256 columns = 1 producer wave + NCW consumer waves.
//  * The producer (wave 0) carries ONLY the recurrence, four columns per lane, and hands the raw
//    significands of its owned columns (the right-most 64*NCW; the rest is halo) to LDS.
//  * Each consumer wave turns 64 of them per row into logs and stores them -- 512 contiguous bytes
//    per row and wave -- so the log work (17 of ~25 instructions per cell) is spread over NCW other
//    SIMDs of the same CU, costs nothing on halo columns, and the table still moves 8 B per cell.
// Producer and consumers run PC_U rows apart through a two-slot LDS ring, one barrier per PC_U rows;
// the PC_U logs a consumer lane owns per trip are evaluated stage-major for ILP.
// log S^n_1 (the S1 vector) is not produced here: k_s1 evaluates lgamma(n-a) - lgamma(1-a).
#define PC_U 8
#define PC_BIAS 700  // per-lane shared exponent: leave room below for the smallest cell of a lane
#ifdef STB_STAMPS
__device__ unsigned long long *g_dbg2;  // hand-off timeline: [block<160][trip<1280][4] wall-clock stamps
__device__ unsigned long long *g_dbg;  // diagnostic build: [launch][block<512][wave<4][4]
#endif

#ifdef STB_STAMPS
#define PC_DUMP()                                                                           \
  do {                                                                                      \
    if (g_dbg && lane == 0 && d == 0 && j < 512) {                                          \
      unsigned long long *q_ = g_dbg + (((size_t)k * 512 + j) * 4 + wave) * 4;              \
      q_[0] = t_work;                                                                       \
      q_[1] = __builtin_amdgcn_s_memtime() - t_begin;                                       \
      q_[2] = n_trips;                                                                      \
    }                                                                                       \
  } while (0)
#endif
// LDS-only barrier: waits for this wave's LDS traffic, NOT for its global stores (a __syncthreads()
// would add s_waitcnt vmcnt(0) and stall every consumer on its stores' round trip once per trip)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NCW>
__global__ __launch_bounds__(64 * (1 + NCW)) void k_fill_pc(fill_args A, int k, int P) {
  constexpr int C = 4;
  constexpr int OW = 64 * NCW;      // owned (stored) columns per block
  constexpr int H = 256 - OW;       // halo columns recomputed per block
  __shared__ double2 lt[128];
  __shared__ __attribute__((aligned(32))) double vbuf[2][PC_U][OW];
  __shared__ int ebuf[2][OW];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < 128) lt[tid] = g_logtab[tid];

  const int j = blockIdx.x;
  const int d = blockIdx.y;
  const unsigned N = A.N, M = A.M;
  const int n0 = 2 + k * A.R;
  const int n1 = min((int)N, n0 + A.R - 1);
  const int nf = n0 - 1;
  const int cmin = 2 + j * OW - H;  // first column of the block (halo included)
  double *table = A.tables + (uint64_t)d * A.tstride;

  // ---- producer state: four columns per lane sharing ONE exponent, so that inside a lane the
  // left-neighbour term needs no rescaling (value of cell i = v[i] * 2^ep) ----
  double v[C], ca[C], s = 1.0;
  int ep = 1 + PC_BIAS;
  const int c0 = cmin + lane * C;
  const bool owned = lane * C >= H;
  if (wave == 0) {
    const double a = A.a[d];
    const uint64_t fbase = ((uint64_t)d * 2) * A.W;
    const double *fm_in = A.fm + fbase + (uint64_t)(k & 1) * A.W;
    const int *fe_in = A.fe + fbase + (uint64_t)(k & 1) * A.W;
    const int cmax_f = min(nf, (int)M);
    double m[C];
    int e[C], E = STB_EZ;
#pragma unroll
    for (int i = 0; i < C; i++) {
      const int c = c0 + i;
      m[i] = 0.0;
      e[i] = STB_EZ;
      if (k == 0) {
        if (c == 1) {
          m[i] = 0.5;
          e[i] = 1;
        }
      } else if (c >= 1 && c <= cmax_f) {
        m[i] = fm_in[c];
        e[i] = fe_in[c];
      }
      if (m[i] != 0.0) E = max(E, e[i]);
      ca[i] = (double)c * a;
    }
    if (E == STB_EZ) E = 1;  // an all-zero lane: exponent of the 1 the diagonal will bring
#pragma unroll
    for (int i = 0; i < C; i++) v[i] = (m[i] != 0.0) ? ldexp(m[i], max(e[i] - E, -1000) - PC_BIAS) : 0.0;
    ep = E + PC_BIAS;
  }
  // ---- consumer state ----
  const int ridx = (wave - 1) * 64 + lane;     // my slot in vbuf / ebuf (consumers only)
  const int cc = 2 + j * OW + ridx;            // my column

#ifdef STB_STAMPS
  unsigned long long t_work = 0, t_begin = __builtin_amdgcn_s_memtime(), t_mark = 0;
  int n_trips = 0;
#define PC_MARK() (t_mark = __builtin_amdgcn_s_memtime())
#define PC_ACC() (t_work += __builtin_amdgcn_s_memtime() - t_mark, n_trips++)
#else
#define PC_MARK() do {} while (0)
#define PC_ACC() do {} while (0)
#endif
  int pidx = 0;
  for (int nb = n0; nb <= n1; nb += P, pidx++) {
    const int ne = min(n1, nb + P - 1);
    const int ns = max(nb, max(cmin, 3));  // rows above the block's first column are all zero
    if (wave == 0) {
      // period set-up: freeze the scale of the cross-lane input, s = 2^(ep_left - ep).  Adjacent
      // lanes (4 columns apart) differ by at most (N^2)^4, i.e. 8 log2 N <= 216 bits for N < 2^27,
      // and v_left <= 2^(-700 + 1450), so v_left * s stays below 2^970: no exponent adoption needed.
      const int epl = wave_shr1(ep, ep);
      s = ldexp(1.0, min(max(epl - ep, -1100), 220));
      if (owned) {
#pragma unroll
        for (int i = 0; i < C; i++) ebuf[pidx & 1][lane * C - H + i] = ep;
      }
      if (nb == 2 && cmin < 3) {  // row 2 (nothing is stored for it)
        const double t0 = wave_shr1_zero(v[3]) * s;
        v[3] = fma(1.0 - ca[3], v[3], v[2]);
        v[2] = fma(1.0 - ca[2], v[2], v[1]);
        v[1] = fma(1.0 - ca[1], v[1], v[0]);
        v[0] = fma(1.0 - ca[0], v[0], t0);
      }
    }
    __syncthreads();  // exponents (and, first time, the log table) visible to the consumers
    int myep = 0;
    if (wave > 0) myep = ebuf[pidx & 1][ridx];

    if (ns <= ne) {
      double coef[C];
      if (wave == 0) {
#pragma unroll
        for (int i = 0; i < C; i++) coef[i] = (double)(ns - 1) - ca[i];
      }
      // consumers address row r as rowbase(r)[coff]: a wave-uniform row base advanced by the row
      // pitch, plus a per-lane constant column offset
      double *rowbase = table + stb_row_offset((unsigned)ns, M);
      const int coff = cc - 2;
      const int trips = (ne - ns + 1 + PC_U - 1) / PC_U;
      // trip q: the producer computes rows ns+U*q.., the consumers emit the rows of trip q-1
      for (int q = 0; q <= trips; q++) {
        PC_MARK();
        if (wave == 0) {
          if (q < trips) {
            const int r0 = ns + q * PC_U;
            const int cnt = min(PC_U, ne - r0 + 1);
            for (int u = 0; u < cnt; u++) {
              const double t0 = wave_shr1_zero(v[3]) * s;
              v[3] = fma(coef[3], v[3], v[2]);
              v[2] = fma(coef[2], v[2], v[1]);
              v[1] = fma(coef[1], v[1], v[0]);
              v[0] = fma(coef[0], v[0], t0);
#pragma unroll
              for (int i = 0; i < C; i++) coef[i] += 1.0;
              if (owned) {
                double2 *dst = reinterpret_cast<double2 *>(&vbuf[q & 1][u][lane * C - H]);
                dst[0] = make_double2(v[0], v[1]);
                dst[1] = make_double2(v[2], v[3]);
              }
            }
          }
        } else if (q > 0) {
          const int r0 = ns + (q - 1) * PC_U;
          const int cnt = min(PC_U, ne - r0 + 1);
          if (cnt == PC_U) {
            // U rows of my column, stage-major
            double x[PC_U], z[PC_U], kf[PC_U], r[PC_U], pl[PC_U];
            double2 t[PC_U];
#pragma unroll
            for (int u = 0; u < PC_U; u++) x[u] = vbuf[(q - 1) & 1][u][ridx];
#pragma unroll
            for (int u = 0; u < PC_U; u++) t[u] = lt[(__double2hiint(x[u]) >> 13) & 127];
#pragma unroll
            for (int u = 0; u < PC_U; u++) {
              const int hi = __double2hiint(x[u]);
              z[u] = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(x[u]));
              kf[u] = (double)((int)((hi >> 20) & 0x7ff) - 1023 + myep);
            }
#pragma unroll
            for (int u = 0; u < PC_U; u++) r[u] = fma(z[u], t[u].x, -1.0);
#pragma unroll
            for (int u = 0; u < PC_U; u++) pl[u] = fma(r[u], 0.2, -0.25);
#pragma unroll
            for (int u = 0; u < PC_U; u++) pl[u] = fma(r[u], pl[u], 1.0 / 3.0);
#pragma unroll
            for (int u = 0; u < PC_U; u++) pl[u] = fma(r[u], pl[u], -0.5);
#pragma unroll
            for (int u = 0; u < PC_U; u++) pl[u] = fma(r[u], pl[u], 1.0);
            const unsigned pitch = stb_row_pitch((unsigned)r0, M);
            if (stb_row_pitch((unsigned)(r0 + PC_U - 1), M) == pitch) {
#pragma unroll
              for (int u = 0; u < PC_U; u++)
                rowbase[(size_t)u * pitch + coff] = fma(kf[u], 0.693147180559945309417, fma(r[u], pl[u], t[u].y));
              rowbase += (size_t)PC_U * pitch;
            } else {
#pragma unroll
              for (int u = 0; u < PC_U; u++) {
                rowbase[coff] = fma(kf[u], 0.693147180559945309417, fma(r[u], pl[u], t[u].y));
                rowbase += stb_row_pitch((unsigned)(r0 + u), M);
              }
            }
          } else {
            for (int u = 0; u < cnt; u++) {
              rowbase[coff] = bfp_log(vbuf[(q - 1) & 1][u][ridx], myep, lt);
              rowbase += stb_row_pitch((unsigned)(r0 + u), M);
            }
          }
        }
        PC_ACC();
        lds_barrier();
      }
    }
    if (wave == 0) {
      // renormalise the lane: the largest of the four significands back to 2^-PC_BIAS * [0.5,1)
      int kmax = -4000;
#pragma unroll
      for (int i = 0; i < C; i++)
        if (v[i] != 0.0) kmax = max(kmax, __builtin_amdgcn_frexp_exp(v[i]));
      if (kmax > -4000) {
#pragma unroll
        for (int i = 0; i < C; i++) v[i] = ldexp(v[i], -kmax - PC_BIAS);
        ep += kmax + PC_BIAS;
      }
    }
  }

  if (wave == 0 && n1 < (int)N) {
    const uint64_t fbase = ((uint64_t)d * 2) * A.W;
    double *fm_out = A.fm + fbase + (uint64_t)((k + 1) & 1) * A.W;
    int *fe_out = A.fe + fbase + (uint64_t)((k + 1) & 1) * A.W;
#pragma unroll
    for (int i = 0; i < C; i++) {
      const int c = c0 + i;
      const bool mine = owned || (j == 0 && c == 1);
      if (mine && c >= 1 && c <= (int)M) {
        fm_out[c] = __builtin_amdgcn_frexp_mant(v[i]);
        fe_out[c] = (v[i] != 0.0) ? ep + __builtin_amdgcn_frexp_exp(v[i]) : STB_EZ;
      }
    }
  }
#ifdef STB_STAMPS
  PC_DUMP();
#endif
}

// S1[n-1] = log S^n_1 = log Gamma(n-a)/Gamma(1-a) for n = 1..N, all tables
__global__ __launch_bounds__(256) void k_s1(const double *a, double *S1, uint64_t s1stride, unsigned N) {
  const int d = blockIdx.y;
  const double ad = a[d];
  const double lg1 = lgamma(1.0 - ad);
  for (unsigned n = 1 + blockIdx.x * blockDim.x + threadIdx.x; n <= N; n += gridDim.x * blockDim.x)
    S1[(uint64_t)d * s1stride + n - 1] = (n == 1) ? 0.0 : lgamma((double)n - ad) - lg1;
}

// ---- chain form: ONE launch per fill, column blocks hand their right edge to the next block ------
//
// The forms above advance every strip by R rows per launch and recompute an R-column halo so that
// strips never talk to each other.  Here a column block owns its 64*P columns for ALL rows: what a
// column needs from its left neighbour (one row up) travels wave to wave, so there is no halo, no
// launch per row block and no frontier round trip.  Nothing in the steady state is a barrier: the
// waves of a block run free and meet through counters in LDS.
//
//  * P producer waves, 64 columns each (one per lane, DPP shift inside the wave), carry the
//    recurrence and write raw significands into an LDS ring of CH_RD trips (a trip = CH_U rows).
//    Producer w reads the last column of producer w-1 from that ring, one trip behind it.
//  * NC consumer waves take (trip, slice) items round-robin, turn 8 rows x 64 columns into logs
//    (stage-major, as in k_fill_pc) and store them; they are off the producers' critical path.
//  * The publisher wave writes the block's last column to global memory as 8-byte granules: the raw
//    double per row (-0.0 for an exact zero, so that 0 = "not written yet") and the lane exponent
//    per trip, with write-through stores.  The fetcher wave of the next block reads 128 rows per
//    round trip with L1-bypassing loads, delivers the leading complete trips into an LDS ring and
//    re-reads the rest.  A granule is its own flag (one aligned 8-byte store), so the hand-off needs
//    no fence and no ordering.
//
// Block (d, j) starts at the trip in which the diagonal enters its first column and lags its left
// neighbour by the hand-off latency; a triangular table starts column block j at row 64*P*j anyway.
// Blocks take their (j, d) from an atomic ticket, j-major, so a block only ever waits for a block
// with a smaller ticket, i.e. one that is running or done: forward progress does not depend on
// dispatch order or on how many blocks are resident.  Every wait is bounded by wall-clock time; on
// expiry the block records an error, stops waiting and runs to its end (stb_fill_status).
#define CH_U 8    // rows per trip
#define CH_RD 8   // trips in the significand ring (power of two)
#define CH_RE 32  // trips in the edge ring (power of two, >= 2 * 16)
#define CH_EOFF (1ull << 40)
#define CH_NEGZERO 0x8000000000000000ull

struct chain_args {
  unsigned *hdr;               // [0] ticket, [1] error code, [2] error detail; zeroed per fill
  unsigned long long *edge_v;  // [D][B][EV]  last column of a block, indexed by row
  unsigned long long *edge_e;  // [D][B][NP]  its lane exponent, indexed by trip, + CH_EOFF
  uint64_t EV, NP;
  int D, B;                    // tables, column blocks per table
  int TP, G;                   // trips per period, trips in all (rows 3 .. 2 + G*CH_U)
  unsigned long long timeout;  // wall_clock64 ticks a wait may last
  unsigned tp_magic;           // ceil(2^32 / TP): trip / TP = umulhi(trip, tp_magic) for trip < 2^26
  const unsigned *cnt;         // DOT = 1: occurrence count per cell, in the table's own layout
  const unsigned *item_ptr;    // DOT = 2: [trips * nsg + 1] first entry of every (trip, slice) item
  const unsigned short *ent_pos;  // DOT = 2: row-in-trip << 6 | column-in-slice of each occurring cell
  const unsigned *ent_cnt;     // DOT = 2: its occurrence count
  unsigned nsg;                // DOT = 2: slices per trip in item_ptr
  double *dotp;                // DOT kernels: [D][B][NC] partial sums of count * log S
};

__device__ __forceinline__ int lds_peek(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS executes one wave's instructions in order: data written before the counter is visible to
// whoever sees the counter.  The empty asm keeps the compiler from reordering around it.
__device__ __forceinline__ void lds_post(int *p, int v) {
  asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// DOT: the logs are not stored; each is multiplied by the cell's occurrence count and summed (the
// whole of aterms' table part, lib/samplea.c:68-80, without a table in memory or a second pass).
//   DOT = 1: dense -- a count slab in the table's layout; every cell's log is computed.
//   DOT = 2: sparse -- per (trip, 64-column slice) item the list of cells that occur at all
//            (position in the 8 x 64 tile + count): only their logs are computed, empty items are
//            skipped without even waiting for the producer.
template <int P, int NC, int NF, int DOT>
__global__ __launch_bounds__(64 * (P + NC + 1 + NF)) void k_fill_chain(fill_args A, chain_args X) {
  constexpr int U = CH_U, RD = CH_RD, RE = CH_RE;
  constexpr int OW = 64 * P;  // columns of a block
  static_assert(P >= 1 && P <= NC && NF >= 1 && P + NC + 1 + NF <= 16, "block shape");
  __shared__ double2 lt[128];
  __shared__ __attribute__((aligned(16))) double vbuf[RD][U][OW];
  __shared__ int ebuf[4][OW];
  __shared__ int slot_p[RD][P];
  __shared__ __attribute__((aligned(16))) double edge_in[RE * U];
  __shared__ int edge_e[RE];
  __shared__ int prod_done[P], cons_cnt[NC], pub_done, edge_ready, s_abort;
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) s_ticket = atomicAdd(X.hdr, 1u);
  if (tid < 128) lt[tid] = g_logtab[tid];
  for (int i = tid; i < RE * U; i += blockDim.x) edge_in[i] = 0.0;
  __syncthreads();
  const int j = (int)(s_ticket / (unsigned)X.D);
  const int d = (int)(s_ticket % (unsigned)X.D);
  if (j >= X.B) return;  // (never: the grid is exactly B*D blocks)

  const unsigned N = A.N, M = A.M;
  const int TP = X.TP, G = X.G;
  const int c0 = 1 + j * OW;  // first column of the block
  auto first_trip = [&](int w) {  // trip in which the diagonal reaches the first column of slice w
    const int c = c0 + 64 * w;
    return (c <= 3) ? 0 : (c - 3) / U;
  };
  const int g0b = first_trip(0);
  const bool has_left = j > 0, has_right = j < X.B - 1;
  double *table = A.tables + (uint64_t)d * A.tstride;
  if (tid < P) prod_done[tid] = first_trip(tid);
  if (tid < NC) cons_cnt[tid] = 0;
  if (tid == 0) {
    pub_done = has_right ? first_trip(P - 1) : 0x7fffffff;
    edge_ready = has_left ? g0b : 0x7fffffff;
    s_abort = 0;
  }
  __syncthreads();

  bool aborted = false;
#ifdef STB_STAMPS
  unsigned long long t_wait = 0, t_start = __builtin_amdgcn_s_memtime();
  int n_wait = 0;
#define CH_WAITED() (t_wait += __builtin_amdgcn_s_memtime() - t_w0, n_wait++)
#else
#define CH_WAITED() do {} while (0)
#endif
  // wait until *cnt >= need; bounded: on expiry (or when another wave gave up) stop waiting for good
  auto wait_ge = [&](const int *cnt, int need, unsigned code) {
    if (aborted || lds_peek(cnt) >= need) return;
#ifdef STB_STAMPS
    const unsigned long long t_w0 = __builtin_amdgcn_s_memtime();
#endif
    const unsigned long long t_begin = wall_clock64();
    for (;;) {
      __builtin_amdgcn_s_sleep(1);
      if (lds_peek(cnt) >= need) {
        CH_WAITED();
        return;
      }
      if (lds_peek(&s_abort)) break;
      if ((unsigned long long)wall_clock64() - t_begin > X.timeout) {
        if (lane == 0) {
          __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (__hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __hip_atomic_store(X.hdr + 2, (unsigned)(j | (d << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(X.hdr + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        break;
      }
    }
    aborted = true;
  };

  if (wave < P) {
    // ================= producers =================
    __builtin_amdgcn_s_setprio(3);
    const int w = wave;
    const int g0w = first_trip(w);
    const int col = 64 * w + lane;  // my column inside the block
    const int c = c0 + col;
    const double a = A.a[d];
    // row 2 of the table: S^2_1 = 1 - a, S^2_2 = 1; everything else starts above the diagonal
    double v = (c == 1) ? ldexp(1.0 - a, -1 - PC_BIAS) : (c == 2) ? ldexp(1.0, -1 - PC_BIAS) : 0.0;
    double coef = (double)(2 + g0w * U) - (double)c * a;  // n - 1 - c a for the first row of trip g0w
    double s = 1.0;
    int ep = 1 + PC_BIAS;
    int p = g0w / TP, tin = g0w - p * TP;
    // the consumer item that last used the ring slot a trip is about to overwrite
    int chk_i = (g0w - RD - g0b) * P + w;
    int chk_c = (chk_i >= 0) ? chk_i % NC : w, chk_k = (chk_i >= 0) ? chk_i / NC : 0;  // (first i >= 0 is w)
    const int *left_cnt = (w == 0) ? &edge_ready : &prod_done[w - 1];
    const int *next_cnt = (w < P - 1) ? &prod_done[w + 1] : &pub_done;
    // What a trip needs from the other waves -- the left neighbour's progress, the consumers' and
    // the right neighbour's progress on the ring slot it overwrites -- and its U left inputs are
    // read one trip AHEAD, under the previous trip's arithmetic, so that no LDS round trip sits on
    // the row chain.  The inputs are speculative: they are valid if the counter read BEFORE them
    // (LDS is in order) already covered the trip; otherwise wait and read again.
    int n_left, n_cons, n_next;
    double ne[U];
    auto load_left = [&](double(&x)[U], int g) {
      if (w == 0) {
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = edge_in[(g & (RE - 1)) * U + u];
      } else {
        x[0] = vbuf[(g - 1) & (RD - 1)][U - 1][64 * w - 1];
#pragma unroll
        for (int u = 1; u < U; u++) x[u] = vbuf[g & (RD - 1)][u - 1][64 * w - 1];
      }
    };
    auto look_ahead = [&](int g) {
      n_left = lds_peek(left_cnt);
      n_cons = lds_peek(&cons_cnt[chk_c]);
      n_next = lds_peek(next_cnt);
      asm volatile("" ::: "memory");
      load_left(ne, g);
    };
    look_ahead(g0w);
    for (int g = g0w; g < G; g++) {
      double e[U];
#pragma unroll
      for (int u = 0; u < U; u++) e[u] = ne[u];
      const bool chk = chk_i >= 0;
      const int next_need = (w < P - 1) ? g - RD + 2 : g - RD + 1;
      if (n_left < g + 1 || (chk && (n_cons < chk_k + 1 || n_next < next_need))) {
        wait_ge(left_cnt, g + 1, 0x100u + (unsigned)g);
        if (chk) {
          wait_ge(&cons_cnt[chk_c], chk_k + 1, 0x300u + (unsigned)g);  // slot g % RD converted
          wait_ge(next_cnt, next_need, 0x400u + (unsigned)g);          // ... read by w+1 / published
        }
        asm volatile("" ::: "memory");
        load_left(e, g);
      }
      if (chk) {
        chk_c += P;
        if (chk_c >= NC) {
          chk_c -= NC;
          chk_k++;
        }
      }
      chk_i += P;
      if (g + 1 < G) look_ahead(g + 1);
      if (g == g0w || tin == 0) {
        // ---- period set-up ----
        if (g != g0w && v != 0.0) {  // renormalise: significand back to 2^-PC_BIAS * [0.5,1)
          const int k = __builtin_amdgcn_frexp_exp(v);
          v = ldexp(v, -k - PC_BIAS);
          ep += k + PC_BIAS;
        }
        // freeze the scale of the cross-lane input for the period (bounds: see k_fill_pc)
        int el = ep;
        if (w == 0) {
          if (has_left) el = edge_e[g & (RE - 1)];
        } else {
          el = ebuf[p & 3][64 * w - 1];
          // the row above the first row of a period was produced under the previous exponent
          if (tin == 0 && p >= 1 && lane == 0) e[0] = ldexp(e[0], ebuf[(p - 1) & 3][64 * w - 1] - el);
        }
        int dl = wave_shr1(ep, ep) - ep;
        if (lane == 0) dl = el - ep;
        s = ldexp(1.0, min(max(dl, -1100), 220));
        ebuf[p & 3][col] = ep;
      }
#ifdef STB_STAMPS
      if (g_dbg2 && d == 0 && w == 0 && lane == 0 && j < 160 && g < 1280) g_dbg2[((size_t)j * 1280 + g) * 4 + 3] = wall_clock64();
#endif
      if (lane == 0) slot_p[g & (RD - 1)][w] = p & 3;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double t0 = wave_shr1(v, e[u]) * s;
        v = fma(coef, v, t0);
        coef += 1.0;
        vbuf[g & (RD - 1)][u][col] = v;
      }
#ifdef STB_STAMPS
      if (g_dbg2 && d == 0 && w == P - 1 && lane == 0 && j < 160 && g < 1280) g_dbg2[((size_t)j * 1280 + g) * 4 + 0] = wall_clock64();
#endif
      lds_post(&prod_done[w], g + 1);
      if (++tin == TP) {
        tin = 0;
        p++;
      }
    }
  } else if (wave < P + NC) {
    // ================= consumers =================
    const int ci = wave - P;
    int w = ci % P, t = g0b + ci / P;
    int done = 0;
    double acc = 0.0;  // (DOT) this lane's share of the sum
    if (DOT == 2) {
      // item (t, w) of block j is slice sg = j P + w of the table: its cells are item_ptr[idx] ..
      // item_ptr[idx + 1] with idx = t * nsg + sg.  The range of the NEXT item is read while this
      // one is processed (a wave-uniform load each).
      auto item_range = [&](int tt, int ww, unsigned &b0, unsigned &b1) {
        const unsigned idx = (unsigned)tt * X.nsg + (unsigned)(j * P + ww);
        b0 = X.item_ptr[idx];
        b1 = X.item_ptr[idx + 1];
      };
      unsigned nb = 0, ne = 0;
      if (t < G) item_range(t, w, nb, ne);
      for (; t < G;) {
        const unsigned beg = nb, end = ne;
        int w2 = w + NC % P, t2 = t + NC / P;
        if (w2 >= P) {
          w2 -= P;
          t2++;
        }
        if (t2 < G) item_range(t2, w2, nb, ne);
        if (beg != end && t >= first_trip(w)) {
          // the first 64 entries can come in while we wait for the producer
          unsigned k = beg + lane;
          unsigned pos = (k < end) ? X.ent_pos[k] : 0u, c = (k < end) ? X.ent_cnt[k] : 0u;
          wait_ge(&prod_done[w], t + 1, 0x600u + (unsigned)t);
          const int slot = t & (RD - 1);
          const int pidx = ((TP == 1) ? t : (int)__umulhi((unsigned)t, X.tp_magic)) & 3;  // period of trip t
          for (;;) {
            const int col = 64 * w + (int)(pos & 63u);
            const double val = bfp_log(vbuf[slot][pos >> 6][col], ebuf[pidx][col], lt);
            acc += (c != 0) ? (double)c * val : 0.0;
            if (k - lane + 64 >= end) break;  // (wave-uniform)
            k += 64;
            pos = (k < end) ? X.ent_pos[k] : 0u;
            c = (k < end) ? X.ent_cnt[k] : 0u;
          }
        }
        done++;
        lds_post(&cons_cnt[ci], done);
        w = w2;
        t = t2;
      }
    }
    for (; t < G;) {
      if (t >= first_trip(w)) {
        wait_ge(&prod_done[w], t + 1, 0x600u + (unsigned)t);
        const int ridx = 64 * w + lane;
        const int cc = c0 + ridx;
        const int coff = cc - 2;  // offset in a table row (column 1: the slack before the row)
        const int slot = t & (RD - 1);
        const int myep = ebuf[slot_p[slot][w]][ridx];
        const int r0 = 3 + t * U;
        const unsigned pitch = stb_row_pitch((unsigned)r0, M);
        const bool fast = (unsigned)(r0 + U - 1) <= N && stb_row_pitch((unsigned)(r0 + U - 1), M) == pitch &&
                          !(j == 0 && t == 0 && w == 0);
        const uint64_t rowoff = stb_row_offset((unsigned)r0, M);
        double *rowbase = table + rowoff;
        const unsigned *cntbase = X.cnt + rowoff;
        if (fast) {
          double x[U], z[U], kf[U], r[U], pl[U];
          double2 tt[U];
          unsigned cn[U];
          if (DOT == 1) {
#pragma unroll
            for (int u = 0; u < U; u++) cn[u] = cntbase[(size_t)u * pitch + coff];
          }
#pragma unroll
          for (int u = 0; u < U; u++) x[u] = vbuf[slot][u][ridx];
#pragma unroll
          for (int u = 0; u < U; u++) tt[u] = lt[(__double2hiint(x[u]) >> 13) & 127];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int hi = __double2hiint(x[u]);
            z[u] = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(x[u]));
            kf[u] = (double)((int)((hi >> 20) & 0x7ff) - 1023 + myep);
          }
#pragma unroll
          for (int u = 0; u < U; u++) r[u] = fma(z[u], tt[u].x, -1.0);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], 0.2, -0.25);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], pl[u], 1.0 / 3.0);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], pl[u], -0.5);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], pl[u], 1.0);
#pragma unroll
          for (int u = 0; u < U; u++) {
            const double val = fma(kf[u], 0.693147180559945309417, fma(r[u], pl[u], tt[u].y));
            if (DOT) {
              // (cells outside the table proper -- the row slack -- have count 0 and may hold anything)
              acc += (cn[u] != 0) ? (double)cn[u] * val : 0.0;
            } else {
              rowbase[(size_t)u * pitch + coff] = val;
            }
          }
        } else {
          for (int u = 0; u < U; u++) {
            const int rr = r0 + u;
            if ((unsigned)rr <= N && cc >= 2) {
              const double val = bfp_log(vbuf[slot][u][ridx], myep, lt);
              if (DOT) {
                const unsigned c = cntbase[coff];
                acc += (c != 0) ? (double)c * val : 0.0;
              } else {
                rowbase[coff] = val;
              }
            }
            const unsigned pt = stb_row_pitch((unsigned)rr, M);
            rowbase += pt;
            cntbase += pt;
          }
        }
      }
      done++;
      lds_post(&cons_cnt[ci], done);
      w += NC % P;
      t += NC / P;
      if (w >= P) {
        w -= P;
        t++;
      }
    }
    if (DOT) {
      // fixed-shape tree over the wave: the same bits on every run
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) X.dotp[((uint64_t)d * X.B + j) * NC + ci] = acc;
    }
  } else if (wave == P + NC) {
    // ================= publisher =================
    if (has_right) {
      __builtin_amdgcn_s_setprio(2);
      unsigned long long *ev_out = X.edge_v + ((uint64_t)d * X.B + j) * X.EV;
      unsigned long long *ee_out = X.edge_e + ((uint64_t)d * X.B + j) * X.NP;
      const int t_first = first_trip(P - 1);
      int pp = t_first / TP, ptin = t_first - pp * TP;  // period of trip t, tracked without dividing
      for (int t = t_first; t < G; t++) {
        // (latency matters here, not issue slots: spin without sleeping)
        if (!aborted) {
          const unsigned long long t_begin = wall_clock64();
          while (lds_peek(&prod_done[P - 1]) < t + 1) {
            if (lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin > X.timeout) {
              wait_ge(&prod_done[P - 1], t + 1, 0x700u + (unsigned)t);  // (records the failure)
              break;
            }
          }
        }
        const int slot = t & (RD - 1);
        if (lane < U) {
          unsigned long long b = (unsigned long long)__double_as_longlong(vbuf[slot][lane][OW - 1]);
          if ((b << 1) == 0) b = CH_NEGZERO;
          __hip_atomic_store(ev_out + 3 + t * U + lane, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == U) {
          const long long ex = (long long)ebuf[pp & 3][OW - 1] + (long long)CH_EOFF;
          __hip_atomic_store(ee_out + t, (unsigned long long)ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef STB_STAMPS
        if (g_dbg2 && d == 0 && lane == 0 && j < 160 && t < 1280) g_dbg2[((size_t)j * 1280 + t) * 4 + 1] = wall_clock64();
#endif
        lds_post(&pub_done, t + 1);
        if (++ptin == TP) {
          ptin = 0;
          pp++;
        }
      }
    }
  } else {
    // ================= fetchers (waves P+NC+1 ..) =================
    if (has_left) {
      const unsigned long long *ev_in = X.edge_v + ((uint64_t)d * X.B + (j - 1)) * X.EV;
      const unsigned long long *ee_in = X.edge_e + ((uint64_t)d * X.B + (j - 1)) * X.NP;
      unsigned long long t_begin = 0;
      bool timing = false;
      for (int t = g0b; t < G;) {
        // (NF fetcher waves run this same loop out of step: each delivers what its own load found
        // complete beyond what has been delivered already, so the polling period divides by NF)
        t = max(t, lds_peek(&edge_ready));
        if (t >= G) break;
        // trips t .. t+nt-1 may be written: their ring slots were read by the first producer
        int lim = lds_peek(&prod_done[0]) + RE;
        if (lim > G) lim = G;
        if (lim <= t) {
          wait_ge(&prod_done[0], t - RE + 1, 0x800u + (unsigned)t);
          if (aborted) break;
          continue;
        }
        const int nt = min(16, lim - t);
        // 128 rows (the values one row above the rows they feed) and 17 trip exponents, one round trip
        const int row0 = 2 + t * U;
        const int ra = row0 + lane, rb = row0 + 64 + lane;
        const bool need_a = lane < 8 * nt, need_b = 64 + lane < 8 * nt;
        const bool need_e = lane <= nt;
        unsigned long long va = 0, vb = 0, ve = 0;
        if (need_a) va = __hip_atomic_load(ev_in + ra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need_b) vb = __hip_atomic_load(ev_in + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need_e) ve = __hip_atomic_load(ee_in + t - 1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ma = __ballot(!need_a || va != 0);
        const unsigned long long mb = __ballot(!need_b || vb != 0);
        const unsigned long long me = __ballot(!need_e || ve != 0);
        int nr = 0;  // leading trips with all 8 rows, their exponent and the one before it present
        for (; nr < nt; nr++) {
          const unsigned long long rows = (nr < 8) ? (ma >> (8 * nr)) : (mb >> (8 * (nr - 8)));
          if ((rows & 0xffull) != 0xffull || ((me >> nr) & 3ull) != 3ull) break;
        }
        if (nr == 0) {
          if (!timing) {
            timing = true;
            t_begin = wall_clock64();
          }
          const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (err != 0 || lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin > X.timeout) {
            if (lane == 0) {
              __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (err == 0) {
                __hip_atomic_store(X.hdr + 2, (unsigned)(j | (d << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(X.hdr + 1, 0x900u + (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
            lds_post(&edge_ready, 0x7fffffff);  // release the producer: it runs on with stale edges
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        timing = false;
        // exponent of each row's trip (lane q of ve holds trip t-1+q) and of the trip before it
        const int ex = (int)(long long)(ve - CH_EOFF);
        const int ka = lane >> 3, kb = 8 + (lane >> 3);
        const int ea = __shfl(ex, ka + 1), ea1 = __shfl(ex, ka);
        const int eb = __shfl(ex, kb + 1), eb1 = __shfl(ex, kb);
        double xa = __longlong_as_double((long long)va), xb = __longlong_as_double((long long)vb);
        // the first row of a trip comes from the trip before: bring it to this trip's exponent
        if ((lane & 7) == 0) {
          xa = ldexp(xa, ea1 - ea);
          xb = ldexp(xb, eb1 - eb);
        }
        const int cur = lds_peek(&edge_ready);  // trips below it were delivered by another fetcher
        if (ka < nr && t + ka >= cur) edge_in[((t + ka) & (RE - 1)) * U + (lane & 7)] = xa;
        if (kb < nr && t + kb >= cur) edge_in[((t + kb) & (RE - 1)) * U + (lane & 7)] = xb;
        if (lane >= 1 && lane <= nr && t - 1 + lane >= cur) edge_e[(t - 1 + lane) & (RE - 1)] = ex;
#ifdef STB_STAMPS
        if (g_dbg2 && d == 0 && lane < nr && j < 160 && t + lane < 1280) g_dbg2[((size_t)j * 1280 + t + lane) * 4 + 2] = wall_clock64();
#endif
        t += nr;
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_max(&edge_ready, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
      }
    }
  }
#ifdef STB_STAMPS
  if (g_dbg && lane == 0 && d == 0 && j < 512) {
    unsigned long long *q_ = g_dbg + ((size_t)j * 16 + wave) * 4;
    q_[0] = t_wait;
    q_[1] = __builtin_amdgcn_s_memtime() - t_start;
    q_[2] = n_wait;
    q_[3] = t_start;
  }
#endif
}

// ---- chain form with the logs on OTHER compute units (one or two tables) -----------------------
//
// With one table in flight k_fill_chain leaves 200+ compute units idle while every busy one is
// saturated by its own consumer waves (a slice's logs cost ~3x its recurrence).  Here a producer
// block is only the chain -- P producer waves of two columns per lane, publisher, fetcher, as
// above -- and writes the raw significands straight to their place in the table with
// write-through stores, plus one exponent per lane and period to a side array.  The remaining
// blocks of the SAME launch are converters: block q owns the 64-column chunk q, its 8 waves take
// CX_ITEM-trip items round-robin, wait for the owning producer's progress word, and turn the raw
// significands into logs in place.  A producer publishes "trips complete" only for stores that
// have left the wave (s_waitcnt vmcnt(N), N = the stores of the last CX_LAG trips), so a converter
// that has seen the word may read the bytes (sc1 loads; first touch of those lines on its side).
// Tickets: producer blocks first (j-major), then converter blocks; every wait is on a block with a
// smaller ticket.  The static LDS (the ring) keeps this kernel at one block per compute unit, so
// producers never share theirs.
#define CX_LAG 6   // trips whose raw stores may still be in flight when progress is published
#define CX_ITEM 4  // trips per converter item
#define CX_FLUSH 8 // trips between two write-backs of the raw significands

// one 16-byte write-through store (sc1: the line is written to memory, not kept dirty in this
// XCD's L2), not counted by the compiler: callers order it with their own s_waitcnt vmcnt
typedef double stb_dvec2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_pair_wt(double *p, double2 v) {
  stb_dvec2 x;
  x.x = v.x;
  x.y = v.y;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}

// the cold part of a bounded wait on an LDS counter, out of line so that the callers' row loops stay
// straight-line code: returns false when the wait was given up (timeout, or another wave gave up)
__device__ __attribute__((noinline)) bool chain_wait_slow(const int *cnt, int need, int *abort_flag, unsigned *hdr,
                                                          unsigned long long timeout, unsigned code, unsigned who, int nap) {
  const unsigned long long t_begin = wall_clock64();
  for (;;) {
    // (a spinning wave takes issue slots from the producer it shares a SIMD with)
    for (int i = 0; i < nap; i++) __builtin_amdgcn_s_sleep(2);
    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) return true;
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
    if ((unsigned long long)wall_clock64() - t_begin > timeout) {
      if ((threadIdx.x & 63) == 0) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_load(hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __hip_atomic_store(hdr + 2, who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(hdr + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      return false;
    }
  }
}

struct chainx_args {
  unsigned *progress;        // [D][B][P] trips complete in the table, per producer wave; zeroed per fill
  int *expo;                 // [D][NPer][EWh] lane exponent per period and column pair
  unsigned long long *dump;  // 64 words nobody reads: where the stores of absent columns go
  uint64_t EWh;
  int NPer, Q;               // periods, 64-column chunks per table
};

template <int P>
__global__ __launch_bounds__(512) void k_fill_chainx(fill_args A, chain_args X, chainx_args Y) {
  constexpr int U = CH_U, RD = 16, RE = CH_RE;
  constexpr int OW = 128 * P;  // columns of a producer block
  __shared__ double2 lt[128];
  // only the LAST column of each producer's slice goes through LDS (to the next producer or the
  // publisher); a ring of RD trips leaves the producers ~14 trips of slack against each other
  __shared__ __attribute__((aligned(16))) double xedge[P][RD][U];
  __shared__ double lds_pad[11776];  // (92 KB: one block per compute unit, see above)
  __shared__ int ebuf[4][OW];
  __shared__ int slot_p[RD][P];
  __shared__ __attribute__((aligned(16))) double edge_in[RE * U];
  __shared__ int edge_e[RE];
  __shared__ int prod_done[P], stored_done[P], pub_done, edge_ready, s_abort, seen_prog;
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: keeps the role loops on the scalar unit)
  if (tid == 0) s_ticket = atomicAdd(X.hdr, 1u);
  if (tid < 128) lt[tid] = g_logtab[tid];
  for (int i = tid; i < RE * U; i += blockDim.x) edge_in[i] = 0.0;
  if (tid == 0) lds_pad[X.G % 11776] = 0.0;  // (keeps the pad allocated)
  __syncthreads();
  const unsigned N = A.N, M = A.M;
  const int TP = X.TP, G = X.G;
  const unsigned nprod = (unsigned)X.B * (unsigned)X.D;
  const bool converter = s_ticket >= nprod;
  const int j = converter ? 0 : (int)(s_ticket / (unsigned)X.D);
  const int d = converter ? (int)((s_ticket - nprod) % (unsigned)X.D) : (int)(s_ticket % (unsigned)X.D);
  const int c0 = j * OW;  // first column of the block; column 0 is a dummy that stays zero
  auto first_trip = [&](int w) {  // trip in which the diagonal reaches the first column of slice w
    const int c = c0 + 128 * w;
    return (c <= 3) ? 0 : (c - 3) / U;
  };
  const int g0b = first_trip(0);
  const bool has_left = j > 0, has_right = j < X.B - 1;
  if (tid < P) prod_done[tid] = first_trip(tid);
  if (tid < P) stored_done[tid] = 0;
  if (tid == 0) {
    seen_prog = 0;
    pub_done = first_trip(P - 1);
    edge_ready = has_left ? g0b : 0x7fffffff;
    s_abort = 0;
  }
  __syncthreads();
  double *table = A.tables + (uint64_t)d * A.tstride;
  bool aborted = false;
#ifdef STB_STAMPS
  unsigned long long t_wait = 0;
  const unsigned long long w_start = wall_clock64();
  int n_wait = 0;
#define CX_DUMP(ID)                                                           \
  do {                                                                        \
    if (g_dbg && lane == 0 && d == 0 && (ID) < 512) {                         \
      unsigned long long *q_ = g_dbg + ((size_t)(ID) * 16 + wave) * 4;        \
      q_[0] = t_wait;                                                         \
      q_[1] = wall_clock64();                                                 \
      q_[2] = n_wait;                                                         \
      q_[3] = w_start;                                                        \
    }                                                                         \
  } while (0)
#else
#define CX_DUMP(ID) do {} while (0)
#endif

  if (converter) {
    // ======================= converter block: chunk q of table d =======================
    const int q = (int)((s_ticket - nprod) / (unsigned)X.D);
    if (q >= Y.Q) return;
    const int jo = (64 * q) / OW, wo = ((64 * q) % OW) / 128;  // owning block and producer wave
    const int c0s = jo * OW + 128 * wo;                        // first column of the owning slice
    const int t0 = (c0s <= 3) ? 0 : (c0s - 3) / U;             // its first trip
    const int cc = 64 * q + lane;                              // my column
    const bool ok = cc >= 2 && (unsigned)cc <= M;
    const unsigned *prog = Y.progress + (((uint64_t)d * X.B + jo) * P + wo) * 32;
    const int *expo = Y.expo + (uint64_t)d * Y.NPer * Y.EWh + (cc >> 1);
    double *dump = reinterpret_cast<double *>(Y.dump) + lane;
    for (int ii = wave;; ii += 8) {
      const int ta = t0 + CX_ITEM * ii;
      if (ta >= G) break;
      const int tb = min(G, ta + CX_ITEM);
      // ---- wait for the producer: poll its word sparingly (a hot word slows the store that
      // updates it) and share what was seen through LDS ----
      if (lds_peek(&seen_prog) < tb) {
        const unsigned long long t_begin = wall_clock64();
#ifdef STB_STAMPS
        n_wait++;
#endif
        for (;;) {
          const int pr = (int)__hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (pr > lds_peek(&seen_prog)) lds_post(&seen_prog, pr);
          if (pr >= tb) break;
          bool seen = false;
          for (int k = 0; k < 4 + wave && !seen; k++) {  // (staggered: the waves do not poll in step)
            __builtin_amdgcn_s_sleep(8);
            seen = lds_peek(&seen_prog) >= tb;
          }
          if (seen) break;
          const bool late = (unsigned long long)wall_clock64() - t_begin > X.timeout;
          const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (err != 0 || late) {
            if (err == 0 && lane == 0) {
              __hip_atomic_store(X.hdr + 2, (unsigned)(q | (d << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(X.hdr + 1, 0xA00u + (unsigned)ta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            aborted = true;
            break;
          }
        }
      }
#ifdef STB_STAMPS
      if (n_wait) t_wait += 0;
#endif
      if (aborted) break;
      asm volatile("" ::: "memory");
      for (int t = ta; t < tb; t++) {
        const int r0 = 3 + t * U;
        const unsigned pitch = stb_row_pitch((unsigned)r0, M);
        const bool fast = (unsigned)(r0 + U - 1) <= N && stb_row_pitch((unsigned)(r0 + U - 1), M) == pitch;
        const int myep = __hip_atomic_load(expo + (uint64_t)(t / TP) * Y.EWh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double *cell = ok ? table + stb_row_offset((unsigned)r0, M) + (cc - 2) : dump;
        if (fast) {
          const size_t inc = ok ? pitch : 0;
          double x[U], z[U], kf[U], r[U], pl[U];
          double2 tt[U];
#pragma unroll
          for (int u = 0; u < U; u++)
            x[u] = __longlong_as_double((long long)__hip_atomic_load(
                reinterpret_cast<unsigned long long *>(cell + u * inc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
          for (int u = 0; u < U; u++) tt[u] = lt[(__double2hiint(x[u]) >> 13) & 127];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int hi = __double2hiint(x[u]);
            z[u] = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(x[u]));
            kf[u] = (double)((int)((hi >> 20) & 0x7ff) - 1023 + myep);
          }
#pragma unroll
          for (int u = 0; u < U; u++) r[u] = fma(z[u], tt[u].x, -1.0);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], 0.2, -0.25);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], pl[u], 1.0 / 3.0);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], pl[u], -0.5);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], pl[u], 1.0);
#pragma unroll
          for (int u = 0; u < U; u++) cell[u * inc] = fma(kf[u], 0.693147180559945309417, fma(r[u], pl[u], tt[u].y));
        } else {
          for (int u = 0; u < U; u++) {
            const int rr = r0 + u;
            if ((unsigned)rr <= N) {
              const double x = __longlong_as_double((long long)__hip_atomic_load(
                  reinterpret_cast<unsigned long long *>(cell), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
              *cell = bfp_log(x, myep, lt);
            }
            if (ok) cell += stb_row_pitch((unsigned)rr, M);
          }
        }
      }
    }
    CX_DUMP(256 + q);
    return;
  }

  // ======================= producer block (d, j) =======================
  // helper waves: publisher, fetcher, flusher; with two producers the flusher takes wave 6 so that
  // (waves going round-robin over the four SIMDs) no helper shares a SIMD with a producer
  constexpr int W_PUB = P, W_FETCH = P + 1, W_FLUSH = (P <= 2) ? 6 : P + 2;
  if (wave >= P && wave != W_PUB && wave != W_FETCH && wave != W_FLUSH) return;

  auto wait_ge = [&](const int *cnt, int need, unsigned code) {
    if (aborted || lds_peek(cnt) >= need) return;
    if (!chain_wait_slow(cnt, need, &s_abort, X.hdr, X.timeout, code, (unsigned)(j | (d << 16)), wave < P ? 1 : 4))
      aborted = true;
  };

  if (wave < P) {
    // ================= producers: two columns per lane sharing one exponent =================
    __builtin_amdgcn_s_setprio(3);
    const int w = wave;
    const int g0w = first_trip(w);
    const int col = 128 * w + 2 * lane;  // first of my two columns inside the block
    const int cA = c0 + col, cB = cA + 1;
    const double a = A.a[d];
    // row 2 of the table: S^2_1 = 1 - a, S^2_2 = 1; everything else starts above the diagonal
    double v0 = (cA == 2) ? ldexp(1.0, -1 - PC_BIAS) : 0.0;
    double v1 = (cB == 1) ? ldexp(1.0 - a, -1 - PC_BIAS) : 0.0;
    double coef0 = (double)(2 + g0w * U) - (double)cA * a;  // n - 1 - c a for the first row of trip g0w
    double coef1 = (double)(2 + g0w * U) - (double)cB * a;
    double s = 1.0;
    int ep = 1 + PC_BIAS;
    int p = g0w / TP, tin = g0w - p * TP;
    // where my pair of cells of a row lives (16-byte aligned: cA is even and rows are 512-byte
    // aligned); the pair (0, 1) has no slot and pairs past M are not stored: those go to the dump
    const bool okP = cA >= 2 && (unsigned)cA <= M;
    double *dump = reinterpret_cast<double *>(Y.dump) + 2 * lane;
    double *pA = okP ? table + stb_row_offset((unsigned)(3 + g0w * U), M) + (cA - 2) : dump;
    int *expo = Y.expo + (uint64_t)d * Y.NPer * Y.EWh + (cA >> 1);
    const int *left_cnt = (w == 0) ? &edge_ready : &prod_done[w - 1];
    const int *next_cnt = (w < P - 1) ? &prod_done[w + 1] : &pub_done;
    int n_left, n_next;
    double ne[U];
    auto load_left = [&](double(&x)[U], int g) {
      if (w == 0) {
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = edge_in[(g & (RE - 1)) * U + u];
      } else {
        x[0] = xedge[w - 1][(g - 1) & (RD - 1)][U - 1];
#pragma unroll
        for (int u = 1; u < U; u++) x[u] = xedge[w - 1][g & (RD - 1)][u - 1];
      }
    };
    auto look_ahead = [&](int g) {
      n_left = lds_peek(left_cnt);
      n_next = lds_peek(next_cnt);
      asm volatile("" ::: "memory");
      load_left(ne, g);
    };
    look_ahead(g0w);
    // One trip.  The hot path is straight-line: everything that is rare (a counter that is short, a
    // period boundary, the last, partial trip) sits behind one unlikely branch each.
    auto trip = [&](int g, auto partial_tag) {
      constexpr bool partial = decltype(partial_tag)::value;
      double e[U];
#pragma unroll
      for (int u = 0; u < U; u++) e[u] = ne[u];
      const int next_need = (w < P - 1) ? g - RD + 2 : g - RD + 1;
      if (__builtin_expect(n_left < g + 1 || n_next < next_need, 0)) {
        wait_ge(left_cnt, g + 1, 0x100u + (unsigned)g);
        wait_ge(next_cnt, next_need, 0x400u + (unsigned)g);  // slot g % RD read by w+1 / published
        asm volatile("" ::: "memory");
        load_left(e, g);
      }
      if (g + 1 < G) look_ahead(g + 1);
      if (__builtin_expect(g == g0w || tin == 0, 0)) {
        // ---- period set-up ----
        if (g != g0w) {  // renormalise: the larger significand back to 2^-PC_BIAS * [0.5,1)
          int kmax = -4000;
          if (v0 != 0.0) kmax = __builtin_amdgcn_frexp_exp(v0);
          if (v1 != 0.0) kmax = max(kmax, __builtin_amdgcn_frexp_exp(v1));
          if (kmax > -4000) {
            v0 = ldexp(v0, -kmax - PC_BIAS);
            v1 = ldexp(v1, -kmax - PC_BIAS);
            ep += kmax + PC_BIAS;
          }
        }
        int el = ep;
        if (w == 0) {
          if (has_left) el = edge_e[g & (RE - 1)];
        } else {
          el = ebuf[p & 3][128 * w - 1];
          // the row above the first row of a period was produced under the previous exponent
          if (tin == 0 && p >= 1 && lane == 0) e[0] = ldexp(e[0], ebuf[(p - 1) & 3][128 * w - 1] - el);
        }
        int dl = wave_shr1(ep, ep) - ep;
        if (lane == 0) dl = el - ep;
        s = ldexp(1.0, min(max(dl, -1100), 220));
        *reinterpret_cast<int2 *>(&ebuf[p & 3][col]) = make_int2(ep, ep);
        __hip_atomic_store(expo + (uint64_t)p * Y.EWh, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) slot_p[g & (RD - 1)][w] = p & 3;
      const int r0 = 3 + g * U;
      // (a trip never straddles a change of the row pitch: rows 3+8g .. 10+8g have lengths
      // 1+8g .. 8+8g, inside one group of 64)
      const size_t incA = okP ? stb_row_pitch((unsigned)r0, M) : 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double t0 = wave_shr1(v1, e[u]) * s;
        v1 = fma(coef1, v1, v0);
        v0 = fma(coef0, v0, t0);
        coef0 += 1.0;
        coef1 += 1.0;
        const double2 vv = make_double2(v0, v1);
        if (lane == 63) xedge[w][g & (RD - 1)][u] = v1;
        if (partial)
          *reinterpret_cast<double2 *>(((unsigned)(r0 + u) <= N) ? pA : dump) = vv;
        else
          *reinterpret_cast<double2 *>(pA) = vv;
        pA += incA;
      }
      lds_post(&prod_done[w], g + 1);
      // trips up to g - CX_LAG have left the wave: at most the stores of the last CX_LAG trips
      // (U each, plus an exponent word now and then) can still be in flight.  The flusher wave
      // passes the count on to the converters: a store to a polled word must not sit in THIS queue.
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CX_LAG * U) : "memory");
      if (g + 1 - CX_LAG > g0w) lds_post(&stored_done[w], g + 1 - CX_LAG);
      if (++tin == TP) {
        tin = 0;
        p++;
      }
    };
    const int Gfull = ((int)N >= 2 + U) ? ((int)N - 2) / U : 0;  // trips whose rows all exist
    int g = g0w;
    for (; g < Gfull; g++) trip(g, std::false_type{});
    for (; g < G; g++) trip(g, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_post(&stored_done[w], G);
  } else if (wave == W_PUB) {
    // ================= publisher: the block's last column, and everybody's progress =================
    unsigned long long *ev_out = X.edge_v + ((uint64_t)d * X.B + j) * X.EV;
    unsigned long long *ee_out = X.edge_e + ((uint64_t)d * X.B + j) * X.NP;
    for (int t = first_trip(P - 1); t < G; t++) {
      wait_ge(&prod_done[P - 1], t + 1, 0x700u + (unsigned)t);
      const int slot = t & (RD - 1);
      if (has_right) {
        if (lane < U) {
          unsigned long long b = (unsigned long long)__double_as_longlong(xedge[P - 1][slot][lane]);
          if ((b << 1) == 0) b = CH_NEGZERO;
          __hip_atomic_store(ev_out + 3 + t * U + lane, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == U) {
          const long long ex = (long long)ebuf[slot_p[slot][P - 1]][OW - 1] + (long long)CH_EOFF;
          __hip_atomic_store(ee_out + t, (unsigned long long)ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      lds_post(&pub_done, t + 1);
    }
  } else if (wave == W_FETCH) {
    // ================= fetcher =================
    if (has_left) {
      const unsigned long long *ev_in = X.edge_v + ((uint64_t)d * X.B + (j - 1)) * X.EV;
      const unsigned long long *ee_in = X.edge_e + ((uint64_t)d * X.B + (j - 1)) * X.NP;
      unsigned long long t_begin = 0;
      bool timing = false;
      for (int t = g0b; t < G;) {
        int lim = lds_peek(&prod_done[0]) + RE;
        if (lim > G) lim = G;
        if (lim <= t) {
          wait_ge(&prod_done[0], t - RE + 1, 0x800u + (unsigned)t);
          if (aborted) break;
          continue;
        }
        const int nt = min(16, lim - t);
        const int row0 = 2 + t * U;
        const int ra = row0 + lane, rb = row0 + 64 + lane;
        const bool need_a = lane < 8 * nt, need_b = 64 + lane < 8 * nt;
        const bool need_e = lane <= nt;
        unsigned long long va = 0, vb = 0, ve = 0;
        if (need_a) va = __hip_atomic_load(ev_in + ra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need_b) vb = __hip_atomic_load(ev_in + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need_e) ve = __hip_atomic_load(ee_in + t - 1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ma = __ballot(!need_a || va != 0);
        const unsigned long long mb = __ballot(!need_b || vb != 0);
        const unsigned long long me = __ballot(!need_e || ve != 0);
        int nr = 0;
        for (; nr < nt; nr++) {
          const unsigned long long rows = (nr < 8) ? (ma >> (8 * nr)) : (mb >> (8 * (nr - 8)));
          if ((rows & 0xffull) != 0xffull || ((me >> nr) & 3ull) != 3ull) break;
        }
        if (nr == 0) {
          if (!timing) {
            timing = true;
            t_begin = wall_clock64();
          }
          const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (err != 0 || lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin > X.timeout) {
            if (lane == 0) {
              __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (err == 0) {
                __hip_atomic_store(X.hdr + 2, (unsigned)(j | (d << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(X.hdr + 1, 0x900u + (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
            lds_post(&edge_ready, 0x7fffffff);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        timing = false;
        const int ex = (int)(long long)(ve - CH_EOFF);
        const int ka = lane >> 3, kb = 8 + (lane >> 3);
        const int ea = __shfl(ex, ka + 1), ea1 = __shfl(ex, ka);
        const int eb = __shfl(ex, kb + 1), eb1 = __shfl(ex, kb);
        double xa = __longlong_as_double((long long)va), xb = __longlong_as_double((long long)vb);
        if ((lane & 7) == 0) {
          xa = ldexp(xa, ea1 - ea);
          xb = ldexp(xb, eb1 - eb);
        }
        if (ka < nr) edge_in[((t + ka) & (RE - 1)) * U + (lane & 7)] = xa;
        if (kb < nr) edge_in[((t + kb) & (RE - 1)) * U + (lane & 7)] = xb;
        if (lane >= 1 && lane <= nr) edge_e[(t - 1 + lane) & (RE - 1)] = ex;
        t += nr;
        lds_post(&edge_ready, t);
      }
    }
  } else {
    // ================= flusher: makes the raw significands visible and tells the converters =================
    // The producers store with plain (write-back) stores, which the L2 acknowledges quickly, and
    // post in LDS how many trips have left their queues.  This wave writes the XCD's dirty lines
    // back (agent-scope release) every CX_FLUSH trips and only then passes the counts on.
    unsigned *prog = Y.progress + (((uint64_t)d * X.B + j) * P + (lane < P ? lane : 0)) * 32;
    int told = 0;
    const unsigned long long t_begin = wall_clock64();
    for (;;) {
      const int sd = (lane < P) ? lds_peek(&stored_done[lane]) : 0x7fffffff;
      int m = sd;  // min over the producers
#pragma unroll
      for (int o = 1; o < P; o <<= 1) m = min(m, __shfl_xor(m, o));
      m = __builtin_amdgcn_readfirstlane(m);
      if (m >= told + CX_FLUSH || (m >= G && told < G)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < P) __hip_atomic_store(prog, (unsigned)sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        told = m;
        if (m >= G) break;
      } else {
        __builtin_amdgcn_s_sleep(32);
        if (lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin > 4 * X.timeout) break;
      }
    }
  }
  CX_DUMP(j);
}

// ---- chain form of the V-table fill (lib/stable.c:451-482) ------------------------
//
// V^n_m needs V^{n-1}_m and V^{n-1}_{m-1}: the same stencil as the S table, with plain doubles (the
// ratios stay O(1): no exponents, no logs).  So a block is only the chain: P producer waves of one
// column per lane that store their row segment straight into the table, a publisher and a fetcher
// exactly as in k_fill_chain (8-byte granules; -0.0 stands for an exact zero).  The cell update is
// the reference's own expression with contraction off, so the table is bit-identical to it.
template <int P>
__global__ __launch_bounds__(64 * (P + 2)) void k_fillv_chain(fill_args A, chain_args X) {
  constexpr int U = CH_U, RD = 16, RE = CH_RE;
  constexpr int OW = 64 * P;
  __shared__ __attribute__((aligned(16))) double xedge[P][RD][U];  // last column of each slice
  __shared__ __attribute__((aligned(16))) double edge_in[RE * U];
  __shared__ int prod_done[P], pub_done, edge_ready, s_abort;
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_ticket = atomicAdd(X.hdr, 1u);
  for (int i = tid; i < RE * U; i += blockDim.x) edge_in[i] = 0.0;
  __syncthreads();
  const int j = (int)(s_ticket / (unsigned)X.D);
  const int d = (int)(s_ticket % (unsigned)X.D);
  if (j >= X.B) return;
  const unsigned N = A.N, M = A.M;
  const int G = X.G;          // trips: rows 2 + U g .. 9 + U g
  const int c0 = 1 + j * OW;  // first column of the block
  auto first_trip = [&](int w) {  // trip in which the diagonal reaches the first column of slice w
    const int c = c0 + 64 * w;
    return (c <= 2) ? 0 : (c - 2) / U;
  };
  const int g0b = first_trip(0);
  const bool has_left = j > 0, has_right = j < X.B - 1;
  if (tid < P) prod_done[tid] = first_trip(tid);
  if (tid == 0) {
    pub_done = first_trip(P - 1);
    edge_ready = has_left ? g0b : 0x7fffffff;
    s_abort = 0;
  }
  __syncthreads();
  double *table = A.tables + (uint64_t)d * A.tstride;
  bool aborted = false;
  auto wait_ge = [&](const int *cnt, int need, unsigned code) {
    if (aborted || lds_peek(cnt) >= need) return;
    if (!chain_wait_slow(cnt, need, &s_abort, X.hdr, X.timeout, code, (unsigned)(j | (d << 16)), wave < P ? 1 : 2))
      aborted = true;
  };

  if (wave < P) {
    // ================= producers =================
    __builtin_amdgcn_s_setprio(3);
    const int w = wave;
    const int g0w = first_trip(w);
    const int c = c0 + 64 * w + lane;  // my column
    const double a = A.a[d];
    const double ca = (double)c * a, cb = (double)(c - 1) * a;
    // row 1 (and every row above the diagonal): V_1 = +inf by convention, everything else 0
    double v = (c == 1) ? HUGE_VAL : 0.0;
    // columns 2..M have a slot; column 1 and the columns past M go to the dump
    const bool ok = c >= 2 && (unsigned)c <= M;
    double *dump = reinterpret_cast<double *>(X.edge_e) + lane;  // (64 words the V fill does not use)
    double *pc = ok ? table + stb_vrow_offset((unsigned)(2 + g0w * U), M) + (c - 2) : dump;
    const int *left_cnt = (w == 0) ? &edge_ready : &prod_done[w - 1];
    const int *next_cnt = (w < P - 1) ? &prod_done[w + 1] : &pub_done;
    int n_left, n_next;
    double ne[U];
    auto load_left = [&](double(&x)[U], int g) {
      if (w == 0) {
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = edge_in[(g & (RE - 1)) * U + u];
      } else {
        x[0] = xedge[w - 1][(g - 1) & (RD - 1)][U - 1];
#pragma unroll
        for (int u = 1; u < U; u++) x[u] = xedge[w - 1][g & (RD - 1)][u - 1];
      }
    };
    auto look_ahead = [&](int g) {
      n_left = lds_peek(left_cnt);
      n_next = lds_peek(next_cnt);
      asm volatile("" ::: "memory");
      load_left(ne, g);
    };
    look_ahead(g0w);
    auto trip = [&](int g, auto partial_tag) {
      constexpr bool partial = decltype(partial_tag)::value;
      double e[U];
#pragma unroll
      for (int u = 0; u < U; u++) e[u] = ne[u];
      const int next_need = (w < P - 1) ? g - RD + 2 : g - RD + 1;
      if (__builtin_expect(n_left < g + 1 || n_next < next_need, 0)) {
        wait_ge(left_cnt, g + 1, 0x100u + (unsigned)g);
        wait_ge(next_cnt, next_need, 0x400u + (unsigned)g);
        asm volatile("" ::: "memory");
        load_left(e, g);
      }
      if (g + 1 < G) look_ahead(g + 1);
      const int r0 = 2 + g * U;
      // (rows 2+8g .. 9+8g have lengths 1+8g .. 8+8g: one pitch per trip, as for the S table)
      const size_t inc = ok ? stb_vrow_pitch((unsigned)r0, M) : 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int n = r0 + u;
        const double left = wave_shr1(v, e[u]);
        // lib/stable.c:475-480 (see v_cell), with this lane's constants hoisted
        double y;
        {
#pragma clang fp contract(off)
          const double nm1 = (double)(n - 1);
          const double num = 1.0 + ((c < n) ? ((nm1 - ca) * v) : 0.0);
          const double den = 1.0 / left + (nm1 - cb);
          y = num / den;
        }
        v = (c == 1) ? HUGE_VAL : (c > n) ? 0.0 : y;
        if (lane == 63) xedge[w][g & (RD - 1)][u] = v;
        if (partial)
          *(((unsigned)n <= N) ? pc : dump) = v;
        else
          *pc = v;
        pc += inc;
      }
      lds_post(&prod_done[w], g + 1);
    };
    const int Gfull = ((int)N >= 1 + U) ? ((int)N - 1) / U : 0;  // trips whose rows all exist
    int g = g0w;
    for (; g < Gfull; g++) trip(g, std::false_type{});
    for (; g < G; g++) trip(g, std::true_type{});
  } else if (wave == P) {
    // ================= publisher =================
    if (has_right) {
      unsigned long long *ev_out = X.edge_v + ((uint64_t)d * X.B + j) * X.EV;
      for (int t = first_trip(P - 1); t < G; t++) {
        wait_ge(&prod_done[P - 1], t + 1, 0x700u + (unsigned)t);
        if (lane < U) {
          unsigned long long b = (unsigned long long)__double_as_longlong(xedge[P - 1][t & (RD - 1)][lane]);
          if ((b << 1) == 0) b = CH_NEGZERO;
          __hip_atomic_store(ev_out + 2 + t * U + lane, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_post(&pub_done, t + 1);
      }
    } else {
      // nobody to publish to: only release the ring slots
      for (int t = first_trip(P - 1); t < G; t++) {
        wait_ge(&prod_done[P - 1], t + 1, 0x700u + (unsigned)t);
        lds_post(&pub_done, t + 1);
      }
    }
  } else {
    // ================= fetcher =================
    if (has_left) {
      const unsigned long long *ev_in = X.edge_v + ((uint64_t)d * X.B + (j - 1)) * X.EV;
      unsigned long long t_begin = 0;
      bool timing = false;
      for (int t = g0b; t < G;) {
        int lim = lds_peek(&prod_done[0]) + RE;
        if (lim > G) lim = G;
        if (lim <= t) {
          wait_ge(&prod_done[0], t - RE + 1, 0x800u + (unsigned)t);
          if (aborted) break;
          continue;
        }
        const int nt = min(16, lim - t);
        const int row0 = 1 + t * U;  // the rows one above the rows of trip t
        const bool need_a = lane < 8 * nt, need_b = 64 + lane < 8 * nt;
        unsigned long long va = 0, vb = 0;
        if (need_a) va = __hip_atomic_load(ev_in + row0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need_b) vb = __hip_atomic_load(ev_in + row0 + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ma = __ballot(!need_a || va != 0);
        const unsigned long long mb = __ballot(!need_b || vb != 0);
        int nr = 0;
        for (; nr < nt; nr++) {
          const unsigned long long rows = (nr < 8) ? (ma >> (8 * nr)) : (mb >> (8 * (nr - 8)));
          if ((rows & 0xffull) != 0xffull) break;
        }
        if (nr == 0) {
          if (!timing) {
            timing = true;
            t_begin = wall_clock64();
          }
          const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (err != 0 || lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin > X.timeout) {
            if (lane == 0) {
              __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (err == 0) {
                __hip_atomic_store(X.hdr + 2, (unsigned)(j | (d << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(X.hdr + 1, 0x900u + (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
            lds_post(&edge_ready, 0x7fffffff);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        timing = false;
        const int ka = lane >> 3, kb = 8 + (lane >> 3);
        // (-0.0 is the granule of an exact zero: the cells it feeds are forced to 0 anyway)
        const double xa = (va == CH_NEGZERO) ? 0.0 : __longlong_as_double((long long)va);
        const double xb = (vb == CH_NEGZERO) ? 0.0 : __longlong_as_double((long long)vb);
        if (ka < nr) edge_in[((t + ka) & (RE - 1)) * U + (lane & 7)] = xa;
        if (kb < nr) edge_in[((t + kb) & (RE - 1)) * U + (lane & 7)] = xb;
        t += nr;
        lds_post(&edge_ready, t);
      }
    }
  }
}

static int ensure_logtab() {
  static bool done[64] = {false};
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail("device index %d out of range", dev);
  if (done[dev]) return 0;
  double2 h[128];
  for (int i = 0; i < 128; i++) {
    const long double c = 1.0L + ((long double)i + 0.5L) / 128.0L;
    const double invc = (double)(1.0L / c);
    h[i].x = invc;
    h[i].y = (double)(-logl((long double)invc));
  }
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_logtab), h, sizeof(h)));
  done[dev] = true;
  return 0;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static unsigned frontier_pitch(unsigned M) { return (unsigned)align_up((size_t)M + 2, 64); }

// periods of a launch never exceed this (R <= 252, P >= 1 ... in practice 2-4)
#define STB_PPL_MAX 8

static int env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  if (!s || !*s) return dflt;
  return atoi(s);
}

// geometry of the chain form (k_fill_chain): column blocks per table, trips, edge stream lengths
struct chain_geom {
  int P, NC, NF, B, G;
  uint64_t EV, NP;
  size_t bytes;  // header + edge streams for D tables
};
static chain_geom chain_geometry(unsigned N, unsigned M, int D) {
  chain_geom g;
  // block shape: up to ~5 tables of 10^4 columns narrower blocks on more compute units win (the
  // fill is a latency chain), with three fetcher waves to shorten the hand-off; for more tables in
  // flight wider blocks with fewer hand-offs do (measured, MI355X: D = 1: 0.96 vs 1.10 ms;
  // D = 4: 1.08 vs 1.16; D = 8: 1.48 vs 1.36)
  const bool narrow = (uint64_t)D * M <= 50000;
  g.P = env_int("STB_CHAIN_P", narrow ? 2 : 4);
  if (g.P != 1 && g.P != 2 && g.P != 4) g.P = narrow ? 2 : 4;
  g.NC = env_int("STB_CHAIN_NC", g.P == 4 ? 10 : g.P == 2 ? 6 : 3);
  if (g.NC < g.P) g.NC = g.P;
  g.NF = env_int("STB_CHAIN_NF", narrow ? 3 : 1);
  if (g.NF < 1) g.NF = 1;
  if (g.NF > 3) g.NF = 3;
  if (g.NF == 2) g.NF = 3;  // (compiled shapes have one or three fetchers)
  if (g.P + g.NC + 1 + g.NF > 16) g.NF = 1;
  if (g.P + g.NC + 1 + g.NF > 16) g.NC = 15 - g.NF - g.P;
  const unsigned cols = (M < N - 1) ? M : N - 1;  // columns 1..min(M, N-1) hold stored cells
  g.B = (int)((cols + 64 * g.P - 1) / (64 * g.P));
  if (g.B < 1) g.B = 1;
  g.G = (N > 2) ? (int)((N - 2 + CH_U - 1) / CH_U) : 0;
  g.EV = (uint64_t)3 + (uint64_t)g.G * CH_U + 136;  // the fetcher reads 128 rows at a time
  g.NP = (uint64_t)g.G + 24;                        // ... and 17 trip exponents
  g.bytes = 256 + (size_t)D * g.B * (g.EV + g.NP) * sizeof(unsigned long long);
  return g;
}

// geometry of the chain form with external converters (k_fill_chainx)
struct chainx_geom {
  int P, B, G, Q, NPer;
  uint64_t EV, NP, EWh;
  size_t prog_bytes, zero_bytes, bytes;  // progress words; what is zeroed per fill; everything
};
static chainx_geom chainx_geometry(unsigned N, unsigned M, int D) {
  chainx_geom g;
  g.P = env_int("STB_CHAINX_P", 4);
  if (g.P != 1 && g.P != 2 && g.P != 4) g.P = 4;
  const unsigned cols = (M < N - 1) ? M : N - 1;
  const int OW = 128 * g.P;
  g.B = (int)((cols + 1 + OW - 1) / OW);  // columns 0 (a dummy) .. cols
  if (g.B < 1) g.B = 1;
  g.Q = (int)((cols + 1 + 63) / 64);
  g.G = (N > 2) ? (int)((N - 2 + CH_U - 1) / CH_U) : 0;
  g.EV = (uint64_t)3 + (uint64_t)g.G * CH_U + 136;
  g.NP = (uint64_t)g.G + 24;
  g.NPer = g.G + 2;  // (a period is at least one trip)
  g.EWh = (uint64_t)g.B * OW / 2;
  g.prog_bytes = align_up((size_t)D * g.B * g.P * 32 * sizeof(unsigned), 256);  // one 128-byte line per word
  g.zero_bytes = 256 + g.prog_bytes + (size_t)D * g.B * (g.EV + g.NP) * sizeof(unsigned long long);
  g.bytes = align_up(g.zero_bytes, 256) + 2048 + (size_t)D * g.NPer * g.EWh * sizeof(int);
  return g;
}

static size_t fill_workspace_need(unsigned N, unsigned M, int D) {
  size_t W = frontier_pitch(M);
  const size_t ring = (size_t)D * STB_EP_RING * STB_PPL_MAX * W * sizeof(int);
  size_t chain = (N >= 3 && M >= 2 && D >= 1) ? chain_geometry(N, M, D).bytes + 256 : 0;
  if (N >= 3 && M >= 2 && D >= 1 && D <= 2) {
    const size_t cx = chainx_geometry(N, M, D).bytes + 256;
    if (cx > chain) chain = cx;
  }
  return align_up((size_t)D * sizeof(double), 256) + (size_t)D * 2 * W * (sizeof(double) + sizeof(int)) +
         (ring > chain ? ring : chain) + 512;
}

// enough for D tables and for any smaller batch run in the same workspace
extern "C" size_t stb_fill_workspace_bytes(unsigned N, unsigned M, int D) {
  size_t need = fill_workspace_need(N, M, D);
  for (int d2 = 1; d2 < D && d2 <= 4096; d2++) {  // (the block shape, hence the edge streams, depends on the batch)
    const size_t n2 = fill_workspace_need(N, M, d2);
    if (n2 > need) need = n2;
  }
  return need;
}

#define STB_MODE_BFP 3    // S table, block-floating cells + table log (default)
#define STB_MODE_SPLIT 4  // same arithmetic, recurrence and log in separate kernels / streams
#define STB_MODE_PC 5     // same arithmetic, producer wave + consumer waves through LDS
#define STB_MODE_CHAIN 6  // same arithmetic, one launch: column blocks chained through edge granules
#define STB_MODE_CHAINX 7 // the chain alone in its blocks, logs by converter blocks of the same launch

// auxiliary streams and an event pool for the split variant (per host thread and device)
struct split_ctx {
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[1024];
  int made = 0;
};
static thread_local split_ctx g_split[16];

static int split_event(split_ctx &c, int i, hipEvent_t *out) {
  if (i >= 1024) return fail("split fill: too many row groups");
  while (c.made <= i) {
    HIPCHK(hipEventCreateWithFlags(&c.ev[c.made], hipEventDisableTiming));
    c.made++;
  }
  *out = c.ev[i];
  return 0;
}

// optional per-launch timing: when armed, every fill kernel is launched with a begin/end event pair
// (hipExtLaunchKernelGGL stamps them with the dispatch's own start/stop, i.e. what a kernel trace
// reports), so a caller can obtain the kernel-only time of a fill without a profiler attached.
struct fill_prof {
  bool armed = false;
  int used = 0;
  hipEvent_t ev[2 * 4096];
  int made = 0;
};
static thread_local fill_prof g_prof;

template <int C>
static void launch_fill(const fill_args &A, int k, dim3 grid, int mode, int P, hipStream_t st) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_prof.armed && g_prof.used + 2 <= 2 * 4096) {
    while (g_prof.made < g_prof.used + 2) {
      if (hipEventCreate(&g_prof.ev[g_prof.made]) != hipSuccess) break;
      g_prof.made++;
    }
    if (g_prof.made >= g_prof.used + 2) {
      e0 = g_prof.ev[g_prof.used];
      e1 = g_prof.ev[g_prof.used + 1];
      g_prof.used += 2;
    }
  }
#define STB_LAUNCH(KERN, ...)                                                          \
  do {                                                                                  \
    if (e0)                                                                             \
      hipExtLaunchKernelGGL(KERN, grid, dim3(64), 0, st, e0, e1, 0, __VA_ARGS__);       \
    else                                                                                \
      hipLaunchKernelGGL(KERN, grid, dim3(64), 0, st, __VA_ARGS__);                     \
  } while (0)
  if (mode == STB_MODE_LOGDOM)
    STB_LAUNCH((k_fill_rows<C, STB_MODE_LOGDOM>), A, k);
  else if (mode == STB_MODE_VRATIO)
    STB_LAUNCH((k_fill_rows<C, STB_MODE_VRATIO>), A, k);
  else if (mode == STB_MODE_SCALED)
    STB_LAUNCH((k_fill_rows<C, STB_MODE_SCALED>), A, k);
  else
    STB_LAUNCH((k_fill_bfp<C>), A, k, P);
#undef STB_LAUNCH
}

extern "C" void stb_fill_profile_begin(void) {
  g_prof.armed = true;
  g_prof.used = 0;
}

extern "C" int stb_fill_profile_end(double *kernel_ms_total, int *launches) {
  STB_ENTRY;
  // caller must have synchronised the stream(s) the fills ran on
  g_prof.armed = false;
  double tot = 0.0;
  const int n = g_prof.used / 2;
  for (int i = 0; i < n; i++) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
    tot += ms;
  }
  if (kernel_ms_total) *kernel_ms_total = tot;
  if (launches) *launches = n;
  g_prof.used = 0;
  return 0;
}

// header of the last chain fill issued by this thread (ticket, error code, error detail)
static thread_local unsigned *g_chain_hdr = nullptr;

// set by stb_groups_aterms around its fill: run the chain form as a DOT kernel (no table stored)
struct dot_request {
  const unsigned *cnt = nullptr;             // dense: count slab
  const unsigned *item_ptr = nullptr;        // sparse: see chain_args
  const unsigned short *ent_pos = nullptr;
  const unsigned *ent_cnt = nullptr;
  unsigned nsg = 0;
  double *dotp = nullptr;
  int parts_per_table = 0;  // out: B * NC
};
static thread_local dot_request *g_dot_req = nullptr;

extern "C" int stb_fill_status(void) {
  STB_ENTRY;
  if (!g_chain_hdr) return 0;
  unsigned h[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpy(h, g_chain_hdr, sizeof(h), hipMemcpyDeviceToHost));  // waits for the fill
  if (h[1] != 0)
    return fail("stb_fill_S: chain fill gave up waiting for a neighbour block (trip %u, block %u of table %u)",
                h[1] - 1, h[2] & 0xffffu, h[2] >> 16);
  return 0;
}

extern "C" int stb_default_variant(void);
extern "C" int stb_fill_tuning(unsigned N, unsigned M, int D, int *C_out, int *R_out, int *launches) {
  const bool few = (uint64_t)D * M <= 200000 && N >= 3 && N < (1u << 27);
  const int v = stb_default_variant();
  const int form = v == STB_FILL_SPLIT ? 1 : v == STB_FILL_PC ? 2 : v == STB_FILL_FUSED ? 0 : (v == STB_FILL_CHAIN || v == STB_FILL_CHAINX) ? 3 : (few ? 3 : 2);
  if (v == STB_FILL_CHAINX && D <= 2) {
    const chainx_geom g = chainx_geometry(N, M, D);
    if (C_out) *C_out = g.P;
    if (R_out) *R_out = (int)N;
    if (launches) *launches = 1;
    return 4;
  }
  if (form == 3) {
    const chain_geom g = chain_geometry(N, M, D);
    if (C_out) *C_out = g.P;
    if (R_out) *R_out = (int)N;
    if (launches) *launches = 1;
    return 3;
  }
  int C = env_int("STB_FILL_C", form == 2 ? 4 : 2);
  int R = env_int("STB_FILL_R", form == 2 ? 128 : form == 1 ? 96 : 64);
  if (form == 2 && R > 128) R = 128;
  if (R < 1) R = 1;
  if (C_out) *C_out = C;
  if (R_out) *R_out = R;
  if (launches) *launches = ((int)N - 1 + R - 1) / R;
  return form; /* 0 fused (k_fill_bfp), 1 split (k_rec + k_logconv), 2 producer/consumer (k_fill_pc), 3 chain, 4 chain with external converters */
}

static int fill_common(const double *a_host, int D, unsigned N, unsigned M, double *d_tables,
                       uint64_t table_stride, double *d_S1, uint64_t s1_stride, void *d_ws,
                       size_t ws_bytes, int mode, hipStream_t st) {
  const char *who = (mode == STB_MODE_VRATIO) ? "stb_fill_V" : "stb_fill_S";
  if (D < 1) return fail("%s: D=%d", who, D);
  if (N < 2 || M < 2) return fail("%s: bounds N=%u M=%u too small", who, N, M);
  if (!a_host || !d_tables || !d_ws || (mode != STB_MODE_VRATIO && !d_S1))
    return fail("%s: null pointer", who);
  if (ws_bytes < fill_workspace_need(N, M, D))
    return fail("%s: workspace %zu < %zu", who, ws_bytes, fill_workspace_need(N, M, D));
  const uint64_t need = (mode == STB_MODE_VRATIO) ? stb_vtable_elems(N, M) : stb_table_elems(N, M);
  if (D > 1 && (table_stride < need || (mode != STB_MODE_VRATIO && s1_stride < N)))
    return fail("%s: strides too small", who);
  if (D > 1 && (table_stride & 1)) return fail("%s: table stride must be even", who);
  for (int d = 0; d < D; d++)
    if (!(a_host[d] >= 0.0 && a_host[d] < 1.0))
      return fail("%s: discount %g outside [0,1)", who, a_host[d]);

  // tunables: columns per lane and rows per launch.  Few tables in flight -> the fill is bound by
  // the latency of one row step, so narrow lanes (C=1); many tables -> throughput, wider lanes.
  const bool few = (uint64_t)D * M < 40000;
  int C = env_int("STB_FILL_C", mode == STB_MODE_SPLIT ? 2 : (few ? 1 : 2));
  int R = env_int("STB_FILL_R", mode == STB_MODE_SPLIT ? 96 : (few ? 48 : 64));
  if (C != 1 && C != 2 && C != 4) return fail("STB_FILL_C must be 1, 2 or 4");
  if (R < 1) R = 1;
  int H = (R + C - 1) / C * C;
  if (H > 64 * C - C) return fail("STB_FILL_R=%d too large for C=%d", R, C);
  // rows per renormalisation period
  int P = 1;
  if (mode == STB_MODE_CHAINX && D > 2) mode = STB_MODE_CHAIN;  // converters need a compute unit per chunk
  if ((mode == STB_MODE_PC || mode == STB_MODE_CHAIN || mode == STB_MODE_CHAINX) && N >= (1u << 27))
    mode = STB_MODE_BFP;  // see the scale bound in k_fill_pc
  if ((mode == STB_MODE_CHAIN || mode == STB_MODE_CHAINX) && N < 3) mode = STB_MODE_BFP;
  if (mode == STB_MODE_PC) {
    // geometry is fixed by the block shape: 256 columns per block, NCW consumer waves
    C = 4;
    const int ncw = env_int("STB_PC_CONSUMERS", 2) == 3 ? 3 : 2;
    H = 256 - 64 * ncw;
    R = env_int("STB_FILL_R", H);
    if (R > H) R = H;
    if (R < 1) R = 1;
  }
  if (mode == STB_MODE_BFP || mode == STB_MODE_SPLIT || mode == STB_MODE_PC || mode == STB_MODE_CHAIN ||
      mode == STB_MODE_CHAINX) {
    if (ensure_logtab()) return 1;
    // A cell grows per row by U^n_m = n - m a + S^n_{m-1}/S^n_m, and the last term reaches n(n-1)/2
    // next to the diagonal, so the bound is N^2 per row, not N.  v starts at 2^-BFP_BIAS and the
    // scale factor s is capped at 2^100: 1700 bits of head-room.
    int bits = 1;
    while ((1ull << bits) < (unsigned long long)N) bits++;
    bits = 2 * bits + 1;
    P = (mode == STB_MODE_PC || mode == STB_MODE_CHAIN || mode == STB_MODE_CHAINX ? 1450 : 1700) / bits;  // these start at 2^-700
    int Penv = env_int("STB_FILL_P", 0);
    if (Penv > 0 && Penv < P) P = Penv;
    if (P < 1) P = 1;
    if (P >= R) P = R;
    else P = (R + (R + P - 1) / P - 1) / ((R + P - 1) / P);  // equal-length periods inside a launch
  }

  fill_args A;
  char *ws = (char *)d_ws;
  A.a = (const double *)ws;
  ws += align_up((size_t)D * sizeof(double), 256);
  A.W = frontier_pitch(M);
  A.fm = (double *)ws;
  ws += (size_t)D * 2 * A.W * sizeof(double);
  A.fe = (int *)ws;
  A.tables = d_tables;
  A.tstride = table_stride;
  A.S1 = d_S1;
  A.s1stride = s1_stride;
  A.N = N;
  A.M = M;
  A.R = R;
  A.H = H;
  A.Wv = 64 * C - H;
  HIPCHK(hipMemcpyAsync((void *)A.a, a_host, (size_t)D * sizeof(double), hipMemcpyHostToDevice, st));

  const int nlaunch = ((int)N - 1 + R - 1) / R;  // rows 2..N
  if (mode == STB_MODE_VRATIO && env_int("STB_FILLV_CHAIN", 1) && N >= 2) {
    // the V table in chain form: P producer waves per block, no consumers (nothing to convert)
    int Pv = env_int("STB_FILLV_P", 4);
    if (Pv != 1 && Pv != 2 && Pv != 4) Pv = 4;
    const unsigned cols = (M < N) ? M : N;  // columns 1..min(M, N) (row n stores m <= n)
    chain_args X;
    X.TP = 1;
    X.tp_magic = 0;
    X.item_ptr = nullptr;
    X.ent_pos = nullptr;
    X.ent_cnt = nullptr;
    X.nsg = 0;
    X.G = (int)((N - 1 + CH_U - 1) / CH_U);  // rows 2..N
    X.D = D;
    X.B = (int)((cols + 64 * Pv - 1) / (64 * Pv));
    if (X.B < 1) X.B = 1;
    X.EV = (uint64_t)2 + (uint64_t)X.G * CH_U + 136;
    X.NP = 0;
    X.cnt = nullptr;
    X.dotp = nullptr;
    char *cb = (char *)align_up((size_t)((char *)A.fe + (size_t)D * 2 * A.W * sizeof(int)), 256);
    const size_t vbytes = 256 + 512 + (size_t)D * X.B * X.EV * sizeof(unsigned long long);
    if ((size_t)(cb - (char *)d_ws) + vbytes > ws_bytes) return fail("%s: workspace too small for the chain form", who);
    X.hdr = (unsigned *)cb;
    X.edge_e = (unsigned long long *)(cb + 256);  // (here: the dump for columns without a slot)
    X.edge_v = (unsigned long long *)(cb + 256 + 512);
    X.timeout = (unsigned long long)env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;
    HIPCHK(hipMemsetAsync(cb, 0, align_up(vbytes, 16), st));
    g_chain_hdr = X.hdr;
    hipEvent_t p0 = nullptr, p1 = nullptr;
    if (g_prof.armed && g_prof.used + 2 <= 2 * 4096) {
      while (g_prof.made < g_prof.used + 2) {
        if (hipEventCreate(&g_prof.ev[g_prof.made]) != hipSuccess) break;
        g_prof.made++;
      }
      if (g_prof.made >= g_prof.used + 2) {
        p0 = g_prof.ev[g_prof.used];
        p1 = g_prof.ev[g_prof.used + 1];
        g_prof.used += 2;
      }
    }
    const dim3 grid((unsigned)X.B * (unsigned)D);
#define STB_LAUNCH_VCHAIN(PP)                                                                          \
  do {                                                                                                 \
    if (p0)                                                                                            \
      hipExtLaunchKernelGGL((k_fillv_chain<PP>), grid, dim3(64 * (PP + 2)), 0, st, p0, p1, 0, A, X);   \
    else                                                                                               \
      hipLaunchKernelGGL((k_fillv_chain<PP>), grid, dim3(64 * (PP + 2)), 0, st, A, X);                 \
  } while (0)
    if (Pv == 1) STB_LAUNCH_VCHAIN(1);
    else if (Pv == 2) STB_LAUNCH_VCHAIN(2);
    else STB_LAUNCH_VCHAIN(4);
#undef STB_LAUNCH_VCHAIN
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (mode == STB_MODE_CHAINX) {
    const chainx_geom cg = chainx_geometry(N, M, D);
    int Pc = 1450;
    {
      int bits = 1;
      while ((1ull << bits) < (unsigned long long)N) bits++;
      Pc /= 2 * bits + 1;
    }
    const int Penv = env_int("STB_FILL_P", 0);
    if (Penv > 0 && Penv < Pc) Pc = Penv;
    chain_args X;
    chainx_args Y;
    X.TP = Pc / CH_U;
    if (X.TP < 1) return fail("%s: renormalisation period %d shorter than a trip", who, Pc);
    X.tp_magic = (unsigned)((0x100000000ull + (unsigned)X.TP - 1) / (unsigned)X.TP);
    X.G = cg.G;
    X.D = D;
    X.B = cg.B;
    X.EV = cg.EV;
    X.NP = cg.NP;
    char *cb = (char *)align_up((size_t)((char *)A.fe + (size_t)D * 2 * A.W * sizeof(int)), 256);
    if ((size_t)(cb - (char *)d_ws) + cg.bytes > ws_bytes) return fail("%s: workspace too small for the chain form", who);
    X.hdr = (unsigned *)cb;
    Y.progress = (unsigned *)(cb + 256);
    X.edge_e = (unsigned long long *)(cb + 256 + cg.prog_bytes);
    X.edge_v = X.edge_e + (size_t)D * cg.B * X.NP;
    char *tail = cb + align_up(cg.zero_bytes, 256);
    Y.dump = (unsigned long long *)tail;
    Y.expo = (int *)(tail + 2048);
    Y.EWh = cg.EWh;
    Y.NPer = cg.NPer;
    Y.Q = cg.Q;
    X.timeout = (unsigned long long)env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;  // 100 MHz ticks
    HIPCHK(hipMemsetAsync(cb, 0, align_up(cg.zero_bytes, 16), st));
    g_chain_hdr = X.hdr;
    hipLaunchKernelGGL(k_s1, dim3((N + 255) / 256 < 64 ? (N + 255) / 256 : 64, D), dim3(256), 0, st, A.a, d_S1, s1_stride, N);
    hipEvent_t p0 = nullptr, p1 = nullptr;
    if (g_prof.armed && g_prof.used + 2 <= 2 * 4096) {
      while (g_prof.made < g_prof.used + 2) {
        if (hipEventCreate(&g_prof.ev[g_prof.made]) != hipSuccess) break;
        g_prof.made++;
      }
      if (g_prof.made >= g_prof.used + 2) {
        p0 = g_prof.ev[g_prof.used];
        p1 = g_prof.ev[g_prof.used + 1];
        g_prof.used += 2;
      }
    }
#ifdef STB_STAMPS
    static unsigned long long *h_xdbg = nullptr;
    {
      unsigned long long *z = nullptr;
      if (getenv("STB_STAMP_FILE")) {
        if (!h_xdbg) HIPCHK(hipMalloc(&h_xdbg, sizeof(unsigned long long) * 512 * 16 * 4));
        HIPCHK(hipMemsetAsync(h_xdbg, 0, sizeof(unsigned long long) * 512 * 16 * 4, st));
        z = h_xdbg;
      }
      HIPCHK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_dbg), &z, sizeof(z), 0, hipMemcpyHostToDevice, st));
    }
#endif
    const dim3 grid(((unsigned)cg.B + (unsigned)cg.Q) * (unsigned)D);
#define STB_LAUNCH_CHAINX(PP)                                                                  \
  do {                                                                                         \
    if (p0)                                                                                    \
      hipExtLaunchKernelGGL((k_fill_chainx<PP>), grid, dim3(512), 0, st, p0, p1, 0, A, X, Y);  \
    else                                                                                       \
      hipLaunchKernelGGL((k_fill_chainx<PP>), grid, dim3(512), 0, st, A, X, Y);                \
  } while (0)
    if (cg.P == 1) STB_LAUNCH_CHAINX(1);
    else if (cg.P == 2) STB_LAUNCH_CHAINX(2);
    else STB_LAUNCH_CHAINX(4);
#undef STB_LAUNCH_CHAINX
    HIPCHK(hipGetLastError());
#ifdef STB_STAMPS
    if (getenv("STB_STAMP_FILE") && h_xdbg) {
      HIPCHK(hipStreamSynchronize(st));
      const size_t cnt = (size_t)512 * 16 * 4;
      unsigned long long *h = (unsigned long long *)malloc(cnt * sizeof(*h));
      HIPCHK(hipMemcpy(h, h_xdbg, cnt * sizeof(*h), hipMemcpyDeviceToHost));
      FILE *f = fopen(getenv("STB_STAMP_FILE"), "w");
      for (int jj = 0; jj < 512; jj++)
        for (int w = 0; w < 16; w++) {
          unsigned long long *q = h + ((size_t)jj * 16 + w) * 4;
          if (q[1]) fprintf(f, "%d %d %llu %llu %llu %llu\n", jj, w, q[0], q[1], q[2], q[3]);
        }
      fclose(f);
      free(h);
    }
#endif
    return 0;
  }
  if (mode == STB_MODE_CHAIN) {
    const chain_geom cg = chain_geometry(N, M, D);
    int Pc = 1450;
    {
      int bits = 1;
      while ((1ull << bits) < (unsigned long long)N) bits++;
      Pc /= 2 * bits + 1;
    }
    const int Penv = env_int("STB_FILL_P", 0);
    if (Penv > 0 && Penv < Pc) Pc = Penv;
    chain_args X;
    X.TP = Pc / CH_U;
    if (X.TP < 1) return fail("%s: renormalisation period %d shorter than a trip", who, Pc);
    X.tp_magic = (unsigned)((0x100000000ull + (unsigned)X.TP - 1) / (unsigned)X.TP);
    X.G = cg.G;
    X.D = D;
    X.B = cg.B;
    X.EV = cg.EV;
    X.NP = cg.NP;
    char *cb = (char *)align_up((size_t)((char *)A.fe + (size_t)D * 2 * A.W * sizeof(int)), 256);
    if ((size_t)(cb - (char *)d_ws) + cg.bytes > ws_bytes) return fail("%s: workspace too small for the chain form", who);
    X.hdr = (unsigned *)cb;
    X.edge_e = (unsigned long long *)(cb + 256);
    X.edge_v = X.edge_e + (size_t)D * cg.B * X.NP;
    X.timeout = (unsigned long long)env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;  // 100 MHz ticks
    HIPCHK(hipMemsetAsync(cb, 0, align_up(cg.bytes, 16), st));
    g_chain_hdr = X.hdr;
    hipLaunchKernelGGL(k_s1, dim3((N + 255) / 256 < 64 ? (N + 255) / 256 : 64, D), dim3(256), 0, st, A.a, d_S1, s1_stride, N);
    hipEvent_t p0 = nullptr, p1 = nullptr;
    if (g_prof.armed && g_prof.used + 2 <= 2 * 4096) {
      while (g_prof.made < g_prof.used + 2) {
        if (hipEventCreate(&g_prof.ev[g_prof.made]) != hipSuccess) break;
        g_prof.made++;
      }
      if (g_prof.made >= g_prof.used + 2) {
        p0 = g_prof.ev[g_prof.used];
        p1 = g_prof.ev[g_prof.used + 1];
        g_prof.used += 2;
      }
    }
#ifdef STB_STAMPS
    static unsigned long long *h_cdbg = nullptr;
    {
      unsigned long long *z = nullptr;
      if (getenv("STB_STAMP_FILE")) {
        if (!h_cdbg) HIPCHK(hipMalloc(&h_cdbg, sizeof(unsigned long long) * 512 * 16 * 4));
        HIPCHK(hipMemsetAsync(h_cdbg, 0, sizeof(unsigned long long) * 512 * 16 * 4, st));
        z = h_cdbg;
      }
      HIPCHK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_dbg), &z, sizeof(z), 0, hipMemcpyHostToDevice, st));
    }
    static unsigned long long *h_tl = nullptr;
    {
      unsigned long long *z = nullptr;
      if (getenv("STB_TIMELINE_FILE")) {
        if (!h_tl) HIPCHK(hipMalloc(&h_tl, sizeof(unsigned long long) * 160 * 1280 * 4));
        HIPCHK(hipMemsetAsync(h_tl, 0, sizeof(unsigned long long) * 160 * 1280 * 4, st));
        z = h_tl;
      }
      HIPCHK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_dbg2), &z, sizeof(z), 0, hipMemcpyHostToDevice, st));
    }
#endif
    const dim3 grid((unsigned)cg.B * (unsigned)D);
    X.cnt = g_dot_req ? g_dot_req->cnt : nullptr;
    X.item_ptr = g_dot_req ? g_dot_req->item_ptr : nullptr;
    X.ent_pos = g_dot_req ? g_dot_req->ent_pos : nullptr;
    X.ent_cnt = g_dot_req ? g_dot_req->ent_cnt : nullptr;
    X.nsg = g_dot_req ? g_dot_req->nsg : 0;
    X.dotp = g_dot_req ? g_dot_req->dotp : nullptr;
    if (g_dot_req) g_dot_req->parts_per_table = cg.B * cg.NC;
    const int dot = !g_dot_req ? 0 : (g_dot_req->item_ptr ? 2 : 1);
#define STB_LAUNCH_CHAIN1(PP, NN, FF, DD)                                                                      \
  do {                                                                                                        \
    if (p0)                                                                                                   \
      hipExtLaunchKernelGGL((k_fill_chain<PP, NN, FF, DD>), grid, dim3(64 * (PP + NN + 1 + FF)), 0, st, p0, p1, 0, A, X); \
    else                                                                                                      \
      hipLaunchKernelGGL((k_fill_chain<PP, NN, FF, DD>), grid, dim3(64 * (PP + NN + 1 + FF)), 0, st, A, X);   \
  } while (0)
#define STB_LAUNCH_CHAIN(PP, NN, FF)                 \
  do {                                               \
    if (dot == 2) STB_LAUNCH_CHAIN1(PP, NN, FF, 2);   \
    else if (dot == 1) STB_LAUNCH_CHAIN1(PP, NN, FF, 1); \
    else STB_LAUNCH_CHAIN1(PP, NN, FF, 0);            \
  } while (0)
    const int shape = cg.P * 1000 + cg.NC * 10 + cg.NF;
    // (the summing variants are compiled for the two default block shapes only)
    if (dot != 0 && shape != 2063 && shape != 4101)
      return fail("%s: the fused evaluation needs a default block shape (unset STB_CHAIN_P / _NC / _NF)", who);
    switch (shape) {
      case 2063: STB_LAUNCH_CHAIN(2, 6, 3); break;
      case 4101: STB_LAUNCH_CHAIN(4, 10, 1); break;
      case 1031: STB_LAUNCH_CHAIN1(1, 3, 1, 0); break;
      case 1033: STB_LAUNCH_CHAIN1(1, 3, 3, 0); break;
      case 1061: STB_LAUNCH_CHAIN1(1, 6, 1, 0); break;
      case 1063: STB_LAUNCH_CHAIN1(1, 6, 3, 0); break;
      case 2041: STB_LAUNCH_CHAIN1(2, 4, 1, 0); break;
      case 2043: STB_LAUNCH_CHAIN1(2, 4, 3, 0); break;
      case 2061: STB_LAUNCH_CHAIN1(2, 6, 1, 0); break;
      case 2081: STB_LAUNCH_CHAIN1(2, 8, 1, 0); break;
      case 2083: STB_LAUNCH_CHAIN1(2, 8, 3, 0); break;
      case 4061: STB_LAUNCH_CHAIN1(4, 6, 1, 0); break;
      case 4063: STB_LAUNCH_CHAIN1(4, 6, 3, 0); break;
      case 4081: STB_LAUNCH_CHAIN1(4, 8, 1, 0); break;
      case 4083: STB_LAUNCH_CHAIN1(4, 8, 3, 0); break;
      default: return fail("%s: no chain kernel for %d producers / %d consumers / %d fetchers", who, cg.P, cg.NC, cg.NF);
    }
#undef STB_LAUNCH_CHAIN1
#ifdef STB_STAMPS
    if (getenv("STB_TIMELINE_FILE") && h_tl) {
      HIPCHK(hipStreamSynchronize(st));
      const size_t cnt = (size_t)160 * 1280 * 4;
      unsigned long long *h = (unsigned long long *)malloc(cnt * sizeof(*h));
      HIPCHK(hipMemcpy(h, h_tl, cnt * sizeof(*h), hipMemcpyDeviceToHost));
      FILE *f = fopen(getenv("STB_TIMELINE_FILE"), "w");
      for (int jj = 0; jj < 160; jj++)
        for (int t = 0; t < 1280; t++) {
          unsigned long long *q = h + ((size_t)jj * 1280 + t) * 4;
          if (q[0] | q[1] | q[2] | q[3]) fprintf(f, "%d %d %llu %llu %llu %llu\n", jj, t, q[0], q[1], q[2], q[3]);
        }
      fclose(f);
      free(h);
    }
#endif
#undef STB_LAUNCH_CHAIN
#ifdef STB_STAMPS
    if (getenv("STB_STAMP_FILE") && h_cdbg) {
      HIPCHK(hipStreamSynchronize(st));
      const size_t cnt = (size_t)512 * 16 * 4;
      unsigned long long *h = (unsigned long long *)malloc(cnt * sizeof(*h));
      HIPCHK(hipMemcpy(h, h_cdbg, cnt * sizeof(*h), hipMemcpyDeviceToHost));
      FILE *f = fopen(getenv("STB_STAMP_FILE"), "w");
      for (int jj = 0; jj < 512; jj++)
        for (int w = 0; w < 16; w++) {
          unsigned long long *q = h + ((size_t)jj * 16 + w) * 4;
          if (q[1]) fprintf(f, "%d %d %llu %llu %llu %llu\n", jj, w, q[0], q[1], q[2], q[3]);
        }
      fclose(f);
      free(h);
    }
#endif
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (mode == STB_MODE_PC) {
    const int ncw = (256 - H) / 64;
    const int OW = 64 * ncw;
    hipLaunchKernelGGL(k_s1, dim3((N + 255) / 256 < 64 ? (N + 255) / 256 : 64, D), dim3(256), 0, st, A.a, d_S1, s1_stride, N);
#ifdef STB_STAMPS
    static unsigned long long *h_dbg = nullptr;
    {
      unsigned long long *z = nullptr;
      if (getenv("STB_STAMP_FILE")) {
        if (!h_dbg) HIPCHK(hipMalloc(&h_dbg, sizeof(unsigned long long) * 16 * 512 * 1024));
        HIPCHK(hipMemsetAsync(h_dbg, 0, sizeof(unsigned long long) * 16 * 512 * 1024, st));
        z = h_dbg;
      }
      HIPCHK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_dbg), &z, sizeof(z), 0, hipMemcpyHostToDevice, st));
    }
#endif
    for (int k = 0; k < nlaunch; k++) {
      int n1 = 2 + (k + 1) * R - 1;
      if (n1 > (int)N) n1 = (int)N;
      int ncols = (n1 < (int)M ? n1 : (int)M) - 1;
      if (ncols < 1) ncols = 1;
      dim3 grid((ncols + OW - 1) / OW, D);
      hipEvent_t p0 = nullptr, p1 = nullptr;
      if (g_prof.armed && g_prof.used + 2 <= 2 * 4096) {
        while (g_prof.made < g_prof.used + 2) {
          if (hipEventCreate(&g_prof.ev[g_prof.made]) != hipSuccess) break;
          g_prof.made++;
        }
        if (g_prof.made >= g_prof.used + 2) {
          p0 = g_prof.ev[g_prof.used];
          p1 = g_prof.ev[g_prof.used + 1];
          g_prof.used += 2;
        }
      }
      if (ncw == 3) {
        if (p0) hipExtLaunchKernelGGL((k_fill_pc<3>), grid, dim3(256), 0, st, p0, p1, 0, A, k, P);
        else hipLaunchKernelGGL((k_fill_pc<3>), grid, dim3(256), 0, st, A, k, P);
      } else {
        if (p0) hipExtLaunchKernelGGL((k_fill_pc<2>), grid, dim3(192), 0, st, p0, p1, 0, A, k, P);
        else hipLaunchKernelGGL((k_fill_pc<2>), grid, dim3(192), 0, st, A, k, P);
      }
    }
    HIPCHK(hipGetLastError());
#ifdef STB_STAMPS
    if (getenv("STB_STAMP_FILE") && h_dbg) {
      HIPCHK(hipStreamSynchronize(st));
      size_t cnt = (size_t)16 * 512 * nlaunch;
      unsigned long long *h = (unsigned long long *)malloc(cnt * sizeof(*h));
      HIPCHK(hipMemcpy(h, h_dbg, cnt * sizeof(*h), hipMemcpyDeviceToHost));
      FILE *f = fopen(getenv("STB_STAMP_FILE"), "w");
      for (int k = 0; k < nlaunch; k++)
        for (int jj = 0; jj < 512; jj++)
          for (int w = 0; w < 4; w++) {
            unsigned long long *q = h + (((size_t)k * 512 + jj) * 4 + w) * 4;
            if (q[1]) fprintf(f, "%d %d %d %llu %llu %llu\n", k, jj, w, q[0], q[1], q[2]);
          }
      fclose(f);
      free(h);
    }
#endif
    return 0;
  }
  if (mode == STB_MODE_SPLIT) {
    split_args X;
    X.P = P;
    X.PPL = (R + P - 1) / P;
    if (X.PPL > STB_PPL_MAX) return fail("%s: %d renormalisation periods per launch (max %d); lower STB_FILL_R", who, X.PPL, STB_PPL_MAX);
    X.epbuf = (int *)((char *)A.fe + (size_t)D * 2 * A.W * sizeof(int));
#ifdef STB_STAMPS
    X.stamps = nullptr;
    X.stamp_strips = 512;
    static unsigned long long *g_stamps = nullptr;
    if (getenv("STB_STAMP_FILE")) {
      if (!g_stamps) HIPCHK(hipMalloc(&g_stamps, sizeof(unsigned long long) * 8 * 512 * 2048));
      HIPCHK(hipMemsetAsync(g_stamps, 0, sizeof(unsigned long long) * 8 * 512 * 2048, st));
      X.stamps = g_stamps;
    }
#endif
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return fail("device index %d out of range", dev);
    split_ctx &cx = g_split[dev];
    for (auto &q : cx.aux)
      if (!q) HIPCHK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    int G = env_int("STB_SPLIT_GROUP", 8);  // row-blocks converted per k_logconv launch
    if (G < 1) G = 1;
    if (G > STB_EP_RING / 2) G = STB_EP_RING / 2;
    const int ngroups = (nlaunch + G - 1) / G;
    // events: [g] = recurrence of group g done (on st), [ngroups+g] = conversion of group g done
    for (int k = 0; k < nlaunch; k++) {
      const int g = k / G;
      if (k % G == 0 && g >= 2) {
        // the exponent slots about to be reused belong to group g-2: its conversion must be over
        hipEvent_t e;
        if (split_event(cx, ngroups + g - 2, &e)) return 1;
        HIPCHK(hipStreamWaitEvent(st, e, 0));
      }
      int n1 = 2 + (k + 1) * R - 1;
      if (n1 > (int)N) n1 = (int)N;
      int ncols = (n1 < (int)M ? n1 : (int)M) - 1;
      if (ncols < 1) ncols = 1;
      dim3 grid((ncols + A.Wv - 1) / A.Wv, D);
      hipEvent_t p0 = nullptr, p1 = nullptr;
      if (g_prof.armed && g_prof.used + 2 <= 2 * 4096) {
        while (g_prof.made < g_prof.used + 2) {
          if (hipEventCreate(&g_prof.ev[g_prof.made]) != hipSuccess) break;
          g_prof.made++;
        }
        if (g_prof.made >= g_prof.used + 2) {
          p0 = g_prof.ev[g_prof.used];
          p1 = g_prof.ev[g_prof.used + 1];
          g_prof.used += 2;
        }
      }
#define STB_LAUNCH_REC(CC)                                                               \
  do {                                                                                   \
    if (p0)                                                                              \
      hipExtLaunchKernelGGL((k_rec<CC>), grid, dim3(64), 0, st, p0, p1, 0, A, X, k);     \
    else                                                                                 \
      hipLaunchKernelGGL((k_rec<CC>), grid, dim3(64), 0, st, A, X, k);                   \
  } while (0)
      switch (C) {
        case 1: STB_LAUNCH_REC(1); break;
        case 2: STB_LAUNCH_REC(2); break;
        default: STB_LAUNCH_REC(4); break;
      }
#undef STB_LAUNCH_REC
      if (k % G == G - 1 || k == nlaunch - 1) {
        hipEvent_t ea, eb;
        if (split_event(cx, g, &ea) || split_event(cx, ngroups + g, &eb)) return 1;
        hipStream_t q = cx.aux[g % 3];
        HIPCHK(hipEventRecord(ea, st));
        HIPCHK(hipStreamWaitEvent(q, ea, 0));
        const int ra = 2 + g * G * R;
        const int rb = n1;
        const int cm = (rb < (int)M ? rb : (int)M) - 1;  // columns 2..min(rb,M)
        dim3 cg((unsigned)((cm > 0 ? cm : 1) + 511) / 512, (unsigned)(rb - ra + 1), (unsigned)D);
        hipLaunchKernelGGL(k_logconv, cg, dim3(256), 0, q, A, X, ra, rb);
        HIPCHK(hipEventRecord(eb, q));
      }
    }
    // the caller's stream continues only after every conversion has finished
    for (int g = (ngroups > 3 ? ngroups - 3 : 0); g < ngroups; g++) {
      hipEvent_t e;
      if (split_event(cx, ngroups + g, &e)) return 1;
      HIPCHK(hipStreamWaitEvent(st, e, 0));
    }
    HIPCHK(hipGetLastError());
#ifdef STB_STAMPS
    if (X.stamps) {
      HIPCHK(hipStreamSynchronize(st));
      size_t cnt = (size_t)8 * 512 * nlaunch;
      unsigned long long *h = (unsigned long long *)malloc(cnt * sizeof(*h));
      HIPCHK(hipMemcpy(h, X.stamps, cnt * sizeof(*h), hipMemcpyDeviceToHost));
      FILE *f = fopen(getenv("STB_STAMP_FILE"), "w");
      for (int k = 0; k < nlaunch; k++)
        for (int jj = 0; jj < 512; jj++) {
          unsigned long long *q = h + ((size_t)k * 512 + jj) * 8;
          if (q[0]) fprintf(f, "%d %d %llu %llu %llu %llu %llu\n", k, jj, q[0], q[1], q[2], q[3], q[4]);
        }
      fclose(f);
      free(h);
    }
#endif
    return 0;
  }
  for (int k = 0; k < nlaunch; k++) {
    int n1 = 2 + (k + 1) * R - 1;
    if (n1 > (int)N) n1 = (int)N;
    int ncols = (n1 < (int)M ? n1 : (int)M) - 1;  // owned columns 2..min(n1,M)
    if (ncols < 1) ncols = 1;
    int strips = (ncols + A.Wv - 1) / A.Wv;
    dim3 grid(strips, D);
    switch (C) {
      case 1: launch_fill<1>(A, k, grid, mode, P, st); break;
      case 2: launch_fill<2>(A, k, grid, mode, P, st); break;
      default: launch_fill<4>(A, k, grid, mode, P, st); break;
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int stb_default_variant(void) {
  const int v = env_int("STB_FILL_VARIANT", STB_FILL_SCALED);
  return (v == STB_FILL_LOGDOMAIN || v == STB_FILL_SCALED_STEP || v == STB_FILL_SPLIT || v == STB_FILL_FUSED ||
          v == STB_FILL_PC || v == STB_FILL_CHAIN || v == STB_FILL_CHAINX)
             ? v
             : STB_FILL_SCALED;
}

extern "C" int stb_fill_S(const double *a_host, int D, unsigned N, unsigned M, double *d_tables,
                          uint64_t table_stride, double *d_S1, uint64_t s1_stride, void *d_ws,
                          size_t ws_bytes, int variant, void *stream) {
  STB_ENTRY;
  // STB_FILL_SCALED picks the form by how many table columns are in flight.  Up to about twenty
  // 10^4-column tables the chain form wins (one launch, no halo; measured 1.0 vs 1.3 ms for one
  // table against the split form, 1.3 vs 1.8 ms for eight against the producer/consumer form);
  // beyond that its one block per compute unit is too few waves and the producer/consumer form,
  // which re-launches per 128 rows but keeps six blocks per compute unit, is faster.
  const bool few = (uint64_t)D * M <= 200000;
  const int mode = variant == STB_FILL_LOGDOMAIN ? STB_MODE_LOGDOM
                   : variant == STB_FILL_SCALED_STEP ? STB_MODE_SCALED
                   : variant == STB_FILL_SPLIT ? STB_MODE_SPLIT
                   : variant == STB_FILL_FUSED ? STB_MODE_BFP
                   : variant == STB_FILL_PC ? STB_MODE_PC
                   : variant == STB_FILL_CHAIN ? STB_MODE_CHAIN
                   : variant == STB_FILL_CHAINX ? STB_MODE_CHAINX
                   : (few ? STB_MODE_CHAIN : STB_MODE_PC);
  return fill_common(a_host, D, N, M, d_tables, table_stride, d_S1, s1_stride, d_ws, ws_bytes, mode,
                     (hipStream_t)stream);
}

extern "C" int stb_fill_V(const double *a_host, int D, unsigned N, unsigned M, double *d_vtables,
                          uint64_t vtable_stride, void *d_ws, size_t ws_bytes, void *stream) {
  STB_ENTRY;
  return fill_common(a_host, D, N, M, d_vtables, vtable_stride, nullptr, 0, d_ws, ws_bytes,
                     STB_MODE_VRATIO, (hipStream_t)stream);
}

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
  if (elems & 1) return fail("stb_table_to_float: element count must be even (slabs are)");
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

// ------------------------------------------------------------------------------------------------
// deterministic double-double reductions

struct dd_t {
  double hi, lo;
};

__device__ __forceinline__ void dd_add(dd_t &s, double x) {
  double t = s.hi + x;
  if (isfinite(t)) {
    double bb = t - s.hi;
    s.lo += (s.hi - (t - bb)) + (x - bb);
  }
  s.hi = t;
}
__device__ __forceinline__ void dd_merge(dd_t &s, dd_t o) {
  dd_add(s, o.hi);
  s.lo += o.lo;
}

// reduce one dd per thread over a 256-thread block in a fixed order; result valid in thread 0
__device__ __forceinline__ dd_t block_reduce_dd(dd_t v, dd_t *lds /* [4] */) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    dd_t o;
    o.hi = __shfl_down(v.hi, off, 64);
    o.lo = __shfl_down(v.lo, off, 64);
    dd_merge(v, o);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = lds[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) dd_merge(v, lds[w]);
  }
  __syncthreads();
  return v;
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

// ------------------------------------------------------------------------------------------------
// K3: sweep.  Each block takes a contiguous chunk of pairs and DT discounts; the (n,t) pair is read
// once per DT tables, the row offset computed once, and DT gathers issued.

#define STB_SWEEP_DT 8
#define STB_SWEEP_CHUNK 4096

__global__ __launch_bounds__(256) void k_sweep_partial(const double *tables, uint64_t tstride,
                                                       const double *S1, uint64_t s1stride, int D,
                                                       unsigned N, unsigned M, const uint32_t *n,
                                                       const uint16_t *t, uint64_t G, dd_t *partial,
                                                       int nb) {
  __shared__ dd_t lds[4];
  const int d0 = blockIdx.y * STB_SWEEP_DT;
  const uint64_t g0 = (uint64_t)blockIdx.x * STB_SWEEP_CHUNK;
  const uint64_t g1 = (g0 + STB_SWEEP_CHUNK < G) ? g0 + STB_SWEEP_CHUNK : G;
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

static int sweep_blocks(uint64_t G) { return (int)((G + STB_SWEEP_CHUNK - 1) / STB_SWEEP_CHUNK); }

extern "C" size_t stb_sweep_workspace_bytes(uint64_t G, int D) {
  int nb = sweep_blocks(G);
  if (nb < 1) nb = 1;
  return (size_t)D * nb * sizeof(dd_t) + 256;
}

extern "C" int stb_sweep_S(const double *d_tables, uint64_t table_stride, const double *d_S1,
                           uint64_t s1_stride, int D, unsigned N, unsigned M, const uint32_t *d_n,
                           const uint16_t *d_t, uint64_t G, double *d_out, void *d_ws,
                           size_t ws_bytes, void *stream) {
  STB_ENTRY;
  hipStream_t st = (hipStream_t)stream;
  if (D < 1) return fail("stb_sweep_S: D=%d", D);
  if (ws_bytes < stb_sweep_workspace_bytes(G, D)) return fail("stb_sweep_S: workspace too small");
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
                                                       int nb) {
  __shared__ dd_t lds[4];
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
  if (ws_bytes < stb_terms_workspace_bytes(I, D)) return fail("terms: workspace too small");
  int nb = terms_blocks(I);
  double *d_base = (double *)d_ws;
  dd_t *partial = (dd_t *)((char *)d_ws + align_up((size_t)D * sizeof(double), 256));
  if (base_host)
    HIPCHK(hipMemcpyAsync(d_base, base_host, sizeof(double) * D, hipMemcpyHostToDevice, st));
  if (nb == 0) {
    if (base_host)
      HIPCHK(hipMemcpyAsync(d_out, d_base, sizeof(double) * D, hipMemcpyDeviceToDevice, st));
    else
      HIPCHK(hipMemsetAsync(d_out, 0, sizeof(double) * D, st));
    return 0;
  }
  hipLaunchKernelGGL(k_terms_partial, dim3(nb, D), dim3(256), 0, st, A, d_T, d_bpar, I, partial, nb);
  hipLaunchKernelGGL(k_reduce_final, dim3(D), dim3(256), 0, st, partial, nb, d_out,
                     base_host ? (const double *)d_base : (const double *)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int stb_restaurant_terms(const double *x_host, int D, const uint32_t *d_T,
                                    const double *d_bpar, uint64_t I, double *d_out, void *d_ws,
                                    size_t ws_bytes, void *stream) {
  STB_ENTRY;
  if (D < 1 || D > STB_TERMS_DMAX) return fail("stb_restaurant_terms: D=%d (max %d)", D, STB_TERMS_DMAX);
  terms_args A;
  memset(&A, 0, sizeof(A));
  A.D = D;
  A.mode = 0;
  for (int d = 0; d < D; d++) {
    if (!(x_host[d] > 0)) return fail("stb_restaurant_terms: x=%g", x_host[d]);
    A.x[d] = x_host[d];
    A.p[d] = log(x_host[d]);  // host libm: one scalar per abscissa, same call as samplea.c:66
  }
  return run_terms(A, nullptr, d_T, d_bpar, I, d_out, d_ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int stb_bterms(const double *x_host, int J, double Q, double shape, double apar,
                          const uint32_t *d_T, uint64_t I, double *d_out, void *d_ws,
                          size_t ws_bytes, void *stream) {
  STB_ENTRY;
  if (J < 1 || J > STB_TERMS_DMAX) return fail("stb_bterms: J=%d (max %d)", J, STB_TERMS_DMAX);
  if (!(apar > 0)) return fail("stb_bterms: apar=%g", apar);
  terms_args A;
  double base[STB_TERMS_DMAX];
  memset(&A, 0, sizeof(A));
  A.D = J;
  A.mode = 1;
  for (int j = 0; j < J; j++) {
    if (!(x_host[j] > 0)) return fail("stb_bterms: x=%g", x_host[j]);
    A.x[j] = x_host[j];
    A.q[j] = x_host[j] / apar;
    A.p[j] = lgamma(A.q[j]);                                   // lib/sampleb.c:36
    base[j] = -Q * x_host[j] + (shape - 1) * log(x_host[j]);   // lib/sampleb.c:37
  }
  return run_terms(A, base, d_T, nullptr, I, d_out, d_ws, ws_bytes, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// device-resident group set

struct stb_groups {
  int I;
  uint64_t G;
  unsigned N, M;
  int Dmax;
  uint32_t *d_n, *d_T;
  uint16_t *d_t;
  double *d_bpar;
  double *d_tables, *d_S1, *d_out;  // d_out: [2][Dmax]
  uint64_t tstride;
  void *d_ws_fill, *d_ws_sweep, *d_ws_terms;
  size_t ws_fill, ws_sweep, ws_terms;
  hipStream_t st;
  hipEvent_t ev[4];
  // fused evaluation (stb_groups_aterms with the chain form): occurrence count per table cell, the
  // pairs that do not address a table cell (t = 1, t = n, out of bounds), partial sums of the fill
  unsigned *d_cnt;
  uint32_t *d_n2;
  uint16_t *d_t2;
  uint64_t G2;
  double *d_dotp;
  size_t dotp_elems;
  int fused, fused_ready;
  // sparse form of the fused evaluation: CSR of the occurring cells per (trip, slice) item
  unsigned *d_item_ptr;
  unsigned short *d_ent_pos;
  unsigned *d_ent_cnt;
  unsigned nsg;
  int sparse;
};

// The sweep gathers table[row(n) + t]; pairs arrive in restaurant order, i.e. random in (n,t), and a
// random 8-byte gather moves a whole 64-byte sector.  Sorting the pairs once by (n,t) (they are reused
// for every evaluation of a samplea call and for all D tables of a grid) makes neighbouring threads
// read neighbouring addresses.  The sum is order-independent up to rounding and stays deterministic.
__global__ void k_pack_pairs(const uint32_t *n, const uint16_t *t, uint64_t G, uint64_t *key) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) key[g] = ((uint64_t)n[g] << 16) | t[g];
}
__global__ void k_unpack_pairs(const uint64_t *key, uint64_t G, uint32_t *n, uint16_t *t) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) {
    n[g] = (uint32_t)(key[g] >> 16);
    t[g] = (uint16_t)(key[g] & 0xffff);
  }
}

static int sort_pairs(uint32_t *d_n, uint16_t *d_t, uint64_t G, hipStream_t st) {
  if (G < 2) return 0;
  uint64_t *k0 = nullptr, *k1 = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  int rc = 1;
  do {
    if (pool_malloc((void **)&k0, sizeof(uint64_t) * G) != hipSuccess || pool_malloc((void **)&k1, sizeof(uint64_t) * G) != hipSuccess) {
      fail("sort_pairs: out of device memory");
      break;
    }
    const unsigned blocks = (unsigned)((G + 255) / 256);
    hipLaunchKernelGGL(k_pack_pairs, dim3(blocks), dim3(256), 0, st, d_n, d_t, G, k0);
    if (rocprim::radix_sort_keys(nullptr, tmp_bytes, k0, k1, (size_t)G, 0, 48, st) != hipSuccess) {
      fail("sort_pairs: radix_sort_keys (size query) failed");
      break;
    }
    if (pool_malloc(&tmp, tmp_bytes ? tmp_bytes : 1) != hipSuccess) {
      fail("sort_pairs: out of device memory");
      break;
    }
    if (rocprim::radix_sort_keys(tmp, tmp_bytes, k0, k1, (size_t)G, 0, 48, st) != hipSuccess) {
      fail("sort_pairs: radix_sort_keys failed");
      break;
    }
    hipLaunchKernelGGL(k_unpack_pairs, dim3(blocks), dim3(256), 0, st, k1, G, d_n, d_t);
    if (hipStreamSynchronize(st) != hipSuccess) {
      fail("sort_pairs: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    rc = 0;
  } while (0);
  // (the stream was synchronised above, or nothing was launched on these buffers)
  if (rc != 0) (void)hipStreamSynchronize(st);
  pool_free(k0);
  pool_free(k1);
  pool_free(tmp);
  return rc;
}

extern "C" void stb_groups_free(stb_groups_t *g) {
  STB_ENTRY;
  if (!g) return;
  void *ptrs[] = {g->d_n, g->d_T, g->d_t, g->d_bpar, g->d_tables, g->d_S1, g->d_out,
                  g->d_ws_fill, g->d_ws_sweep, g->d_ws_terms, g->d_cnt, g->d_n2, g->d_t2, g->d_dotp,
                  g->d_item_ptr, g->d_ent_pos, g->d_ent_cnt};
  if (g->st) (void)hipStreamSynchronize(g->st);  // nothing may still be using the buffers
  for (void *p : ptrs) pool_free(p);
  for (auto &e : g->ev)
    if (e) (void)hipEventDestroy(e);
  if (g->st) (void)hipStreamDestroy(g->st);
  free(g);
}

#define GCHK(expr)                                                                            \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);        \
      stb_groups_free(g);                                                                     \
      return nullptr;                                                                         \
    }                                                                                         \
  } while (0)

// occurrence count of every table cell among the pairs (same classification as k_sweep_partial)
__global__ void k_count_pairs(const uint32_t *n, const uint16_t *t, uint64_t G, unsigned N, unsigned M,
                              unsigned *cnt) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const unsigned nn = n[g], tt = t[g];
  if (nn <= 1 || nn == tt || tt <= 1 || nn < tt || tt > M || nn > N) return;
  atomicAdd(&cnt[stb_row_offset(nn, M) + (tt - 2)], 1u);
}

// out[d] += sum of the DOT kernel's partial sums of table d, in a fixed order
__global__ __launch_bounds__(64) void k_dot_reduce(const double *dotp, int parts, double *out) {
  const int d = blockIdx.x, lane = threadIdx.x;
  dd_t acc{0.0, 0.0};
  for (int i = lane; i < parts; i += 64) dd_add(acc, dotp[(size_t)d * parts + i]);
  // lanes in order, on lane 0
  dd_t tot{0.0, 0.0};
  for (int l = 0; l < 64; l++) {
    const double hi = __shfl(acc.hi, l), lo = __shfl(acc.lo, l);
    dd_add(tot, hi);
    dd_add(tot, lo);
  }
  if (lane == 0) out[d] += tot.hi + tot.lo;
}

extern "C" stb_groups_t *stb_groups_create(int I, const int *K, const uint32_t *T,
                                           const uint32_t *nflat, const uint16_t *tflat,
                                           const double *bpar, unsigned N, unsigned M, int Dmax) {
  STB_ENTRY;
  if (stb_device_count() < 1) {
    fail("stb_groups_create: no HIP device (libstb_amd has no CPU path)");
    return nullptr;
  }
  if (Dmax < 1 || Dmax > STB_TERMS_DMAX) {
    fail("stb_groups_create: Dmax=%d (1..%d)", Dmax, STB_TERMS_DMAX);
    return nullptr;
  }
  stb_groups_t *g = (stb_groups_t *)calloc(1, sizeof(*g));
  if (!g) {
    fail("stb_groups_create: out of host memory");
    return nullptr;
  }
  uint64_t G = 0;
  for (int i = 0; i < I; i++) G += (uint64_t)(K[i] > 0 ? K[i] : 0);
  g->I = I;
  g->G = G;
  g->N = N;
  g->M = M;
  g->Dmax = Dmax;
  g->tstride = (stb_table_elems(N, M) + 31) & ~31ull;
  GCHK(hipStreamCreate(&g->st));
  for (auto &e : g->ev) GCHK(hipEventCreate(&e));
  GCHK(pool_malloc((void **)&g->d_n, sizeof(uint32_t) * (G ? G : 1)));
  GCHK(pool_malloc((void **)&g->d_t, sizeof(uint16_t) * (G ? G : 1)));
  GCHK(pool_malloc((void **)&g->d_T, sizeof(uint32_t) * (I > 0 ? I : 1)));
  GCHK(pool_malloc((void **)&g->d_bpar, sizeof(double) * (I > 0 ? I : 1)));
  GCHK(pool_malloc((void **)&g->d_tables, sizeof(double) * g->tstride * Dmax));
  GCHK(pool_malloc((void **)&g->d_S1, sizeof(double) * (size_t)N * Dmax));
  GCHK(pool_malloc((void **)&g->d_out, sizeof(double) * 2 * Dmax));
  g->ws_fill = stb_fill_workspace_bytes(N, M, Dmax);
  g->ws_sweep = stb_sweep_workspace_bytes(G, Dmax);
  g->ws_terms = stb_terms_workspace_bytes((uint64_t)I, Dmax);
  GCHK(pool_malloc((void **)&g->d_ws_fill, g->ws_fill));
  GCHK(pool_malloc((void **)&g->d_ws_sweep, g->ws_sweep));
  GCHK(pool_malloc((void **)&g->d_ws_terms, g->ws_terms));
  if (G) {
    GCHK(hipMemcpy(g->d_n, nflat, sizeof(uint32_t) * G, hipMemcpyHostToDevice));
    GCHK(hipMemcpy(g->d_t, tflat, sizeof(uint16_t) * G, hipMemcpyHostToDevice));
  }
  if (I > 0) {
    GCHK(hipMemcpy(g->d_T, T, sizeof(uint32_t) * I, hipMemcpyHostToDevice));
    GCHK(hipMemcpy(g->d_bpar, bpar, sizeof(double) * I, hipMemcpyHostToDevice));
  }
  if (env_int("STB_SORT_PAIRS", 1) && sort_pairs(g->d_n, g->d_t, G, g->st)) {
    stb_groups_free(g);
    return nullptr;
  }
  g->fused = env_int("STB_ATERMS_FUSED", 1) && N >= 3 && N < (1u << 27);  // set up on first use
  return g;
}

// ---- sparse set-up of the fused evaluation, all on the device -----------------------------------
#define STB_KEY_OTHER 0xfffffffffffffffeull  // a pair that addresses no table cell (t = 1, t = n, out of bounds)
#define STB_KEY_SKIP 0xffffffffffffffffull   // n <= 1: contributes nothing (lib/samplea.c:78)

// key of a pair: (item index << 9) | (row in trip << 6) | column in slice, item = trip * nsg + slice
__global__ void k_item_keys(const uint32_t *n, const uint16_t *t, uint64_t G, unsigned N, unsigned M, unsigned nsg,
                            uint64_t *key, uint32_t *payload) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const unsigned nn = n[g], tt = t[g];
  uint64_t k;
  if (nn <= 1) k = STB_KEY_SKIP;
  else if (nn == tt || tt <= 1 || nn < tt || tt > M || nn > N) k = STB_KEY_OTHER;
  else {
    const unsigned trip = (nn - 3) >> 3, u = (nn - 3) & 7, sg = (tt - 1) >> 6, ln = (tt - 1) & 63;
    k = (((uint64_t)trip * nsg + sg) << 9) | (u << 6) | ln;
  }
  key[g] = k;
  payload[g] = (uint32_t)g;
}

// counts[0] = keys below STB_KEY_OTHER (table cells), counts[1] = keys equal to it
__global__ void k_key_bounds(const uint64_t *key, uint64_t G, uint64_t *counts) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  uint64_t lo = 0, hi = G;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) / 2;
    if (key[mid] < STB_KEY_OTHER) lo = mid + 1;
    else hi = mid;
  }
  const uint64_t a = lo;
  hi = G;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) / 2;
    if (key[mid] <= STB_KEY_OTHER) lo = mid + 1;
    else hi = mid;
  }
  counts[0] = a;
  counts[1] = lo - a;
}

__global__ void k_gather_pairs(const uint32_t *n, const uint16_t *t, const uint32_t *idx, uint64_t cnt, uint32_t *n2,
                               uint16_t *t2) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < cnt) {
    n2[g] = n[idx[g]];
    t2[g] = t[idx[g]];
  }
}

__global__ void k_split_runs(const uint64_t *ukey, const unsigned *runs, unsigned short *pos, unsigned *item) {
  const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < *runs) {
    pos[r] = (unsigned short)(ukey[r] & 511u);
    item[r] = (unsigned)(ukey[r] >> 9);
  }
}

// item_ptr[i] = first run whose item index is >= i
__global__ void k_item_ptr(const unsigned *item, const unsigned *runs, unsigned nitems, unsigned *ptr) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nitems) return;
  unsigned lo = 0, hi = *runs;
  while (lo < hi) {
    const unsigned mid = (lo + hi) / 2;
    if (item[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  ptr[i] = lo;
}

// Returns 0 and sets g->sparse = 1 when the sparse form was built, 0 with g->sparse = 0 when the
// pairs are too dense for it to pay (the caller then builds the count slab), non-zero on error.
static int groups_fused_setup_sparse(stb_groups_t *g) {
  const unsigned N = g->N, M = g->M;
  const uint64_t G = g->G;
  g->sparse = 0;
  if (G == 0 || G >= 0xffffffffull) return 0;
  const unsigned nsg = (M + 63) / 64 + 4;
  const unsigned trips = (N - 2 + 7) / 8;
  const uint64_t nitems64 = (uint64_t)trips * nsg;
  if (nitems64 >= (1ull << 31)) return 0;
  const unsigned nitems = (unsigned)nitems64;
  uint64_t *k0 = nullptr, *k1 = nullptr, *uk = nullptr, *d_counts = nullptr;
  uint32_t *p0 = nullptr, *p1 = nullptr;
  unsigned *cnt = nullptr, *runs = nullptr, *item = nullptr;
  void *tmp = nullptr;
  int rc = 1;
  const unsigned blocks = (unsigned)((G + 255) / 256);
  do {
    if (pool_malloc((void **)&k0, 8 * G) != hipSuccess || pool_malloc((void **)&k1, 8 * G) != hipSuccess ||
        pool_malloc((void **)&p0, 4 * G) != hipSuccess || pool_malloc((void **)&p1, 4 * G) != hipSuccess ||
        pool_malloc((void **)&uk, 8 * G) != hipSuccess || pool_malloc((void **)&cnt, 4 * G) != hipSuccess ||
        pool_malloc((void **)&item, 4 * G) != hipSuccess || pool_malloc((void **)&runs, 64) != hipSuccess ||
        pool_malloc((void **)&d_counts, 64) != hipSuccess) {
      fail("stb_groups_aterms: out of device memory");
      break;
    }
    hipLaunchKernelGGL(k_item_keys, dim3(blocks), dim3(256), 0, g->st, g->d_n, g->d_t, G, N, M, nsg, k0, p0);
    size_t b1 = 0, b2 = 0;
    if (rocprim::radix_sort_pairs(nullptr, b1, k0, k1, p0, p1, (size_t)G, 0, 64, g->st) != hipSuccess) break;
    if (rocprim::run_length_encode(nullptr, b2, k1, (unsigned)G, uk, cnt, runs, g->st) != hipSuccess) break;
    const size_t tmp_bytes = b1 > b2 ? b1 : b2;
    if (pool_malloc(&tmp, tmp_bytes ? tmp_bytes : 1) != hipSuccess) {
      fail("stb_groups_aterms: out of device memory");
      break;
    }
    size_t bb = tmp_bytes;
    if (rocprim::radix_sort_pairs(tmp, bb, k0, k1, p0, p1, (size_t)G, 0, 64, g->st) != hipSuccess) break;
    hipLaunchKernelGGL(k_key_bounds, dim3(1), dim3(1), 0, g->st, k1, G, d_counts);
    uint64_t h_counts[2] = {0, 0};
    if (hipMemcpyAsync(h_counts, d_counts, 16, hipMemcpyDeviceToHost, g->st) != hipSuccess ||
        hipStreamSynchronize(g->st) != hipSuccess)
      break;
    const uint64_t n_in = h_counts[0], n_other = h_counts[1];
    // dense enough that every cell's log might as well be computed: leave it to the count slab
    if (n_in * 3 > stb_table_cells(N, M)) {
      rc = 0;
      break;
    }
    // the pairs outside the table, in their sorted order
    g->G2 = n_other;
    if (pool_malloc((void **)&g->d_n2, 4 * (n_other ? n_other : 1)) != hipSuccess ||
        pool_malloc((void **)&g->d_t2, 2 * (n_other ? n_other : 1)) != hipSuccess) {
      fail("stb_groups_aterms: out of device memory");
      break;
    }
    if (n_other)
      hipLaunchKernelGGL(k_gather_pairs, dim3((unsigned)((n_other + 255) / 256)), dim3(256), 0, g->st, g->d_n, g->d_t,
                         p1 + n_in, n_other, g->d_n2, g->d_t2);
    // distinct cells with their counts
    unsigned h_runs = 0;
    if (n_in) {
      bb = tmp_bytes;
      if (rocprim::run_length_encode(tmp, bb, k1, (unsigned)n_in, uk, cnt, runs, g->st) != hipSuccess) break;
      if (hipMemcpyAsync(&h_runs, runs, 4, hipMemcpyDeviceToHost, g->st) != hipSuccess ||
          hipStreamSynchronize(g->st) != hipSuccess)
        break;
    } else if (hipMemsetAsync(runs, 0, 4, g->st) != hipSuccess) {
      break;
    }
    if (pool_malloc((void **)&g->d_ent_pos, 2 * (size_t)(h_runs ? h_runs : 1)) != hipSuccess ||
        pool_malloc((void **)&g->d_ent_cnt, 4 * (size_t)(h_runs ? h_runs : 1)) != hipSuccess ||
        pool_malloc((void **)&g->d_item_ptr, 4 * ((size_t)nitems + 2)) != hipSuccess) {
      fail("stb_groups_aterms: out of device memory");
      break;
    }
    if (h_runs) {
      hipLaunchKernelGGL(k_split_runs, dim3((h_runs + 255) / 256), dim3(256), 0, g->st, uk, runs, g->d_ent_pos, item);
      if (hipMemcpyAsync(g->d_ent_cnt, cnt, 4 * (size_t)h_runs, hipMemcpyDeviceToDevice, g->st) != hipSuccess) break;
    }
    hipLaunchKernelGGL(k_item_ptr, dim3((nitems + 1 + 255) / 256), dim3(256), 0, g->st, item, runs, nitems, g->d_item_ptr);
    g->dotp_elems = (size_t)g->Dmax * ((size_t)(M + 63) / 64 + 1) * 16;
    if (pool_malloc((void **)&g->d_dotp, sizeof(double) * g->dotp_elems) != hipSuccess) {
      fail("stb_groups_aterms: out of device memory");
      break;
    }
    if (hipStreamSynchronize(g->st) != hipSuccess || hipGetLastError() != hipSuccess) break;
    g->nsg = nsg;
    g->sparse = 1;
    g->fused_ready = 1;
    rc = 0;
  } while (0);
  if (rc != 0) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) fail("stb_groups_aterms: sparse set-up failed: %s", hipGetErrorString(e));
  }
  (void)hipStreamSynchronize(g->st);
  pool_free(k0);
  pool_free(k1);
  pool_free(p0);
  pool_free(p1);
  pool_free(uk);
  pool_free(cnt);
  pool_free(item);
  pool_free(runs);
  pool_free(d_counts);
  pool_free(tmp);
  return rc;
}

// Lazy set-up of the fused evaluation (first call with more than one discount): occurrence count
// per table cell, and the pairs that address no cell.  The pairs come back from the device in their
// sorted order, so the result does not depend on the order the caller supplied them in.
static int groups_fused_setup(stb_groups_t *g) {
  const unsigned N = g->N, M = g->M;
  const uint64_t G = g->G;
  const uint64_t elems = stb_table_elems(N, M);
  HIPCHK(pool_malloc((void **)&g->d_cnt, sizeof(unsigned) * (elems ? elems : 1)));
  HIPCHK(hipMemsetAsync(g->d_cnt, 0, sizeof(unsigned) * (elems ? elems : 1), g->st));
  if (G)
    hipLaunchKernelGGL(k_count_pairs, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, g->st, g->d_n, g->d_t, G, N,
                       M, g->d_cnt);
  uint32_t *hn = (uint32_t *)malloc(sizeof(uint32_t) * (G ? G : 1));
  uint16_t *ht = (uint16_t *)malloc(sizeof(uint16_t) * (G ? G : 1));
  if (!hn || !ht) {
    free(hn);
    free(ht);
    return fail("stb_groups_aterms: out of host memory");
  }
  hipError_t e1 = hipSuccess, e2 = hipSuccess;
  if (G) {
    e1 = hipMemcpyAsync(hn, g->d_n, sizeof(uint32_t) * G, hipMemcpyDeviceToHost, g->st);
    e2 = hipMemcpyAsync(ht, g->d_t, sizeof(uint16_t) * G, hipMemcpyDeviceToHost, g->st);
  }
  if (e1 == hipSuccess && e2 == hipSuccess) e1 = hipStreamSynchronize(g->st);
  uint64_t G2 = 0;
  if (e1 == hipSuccess && e2 == hipSuccess) {
    for (uint64_t q = 0; q < G; q++) {  // compact in place: the pairs outside the table
      const unsigned nn = hn[q], tt = ht[q];
      if (nn > 1 && (nn == tt || tt <= 1 || nn < tt || tt > M || nn > N)) {
        hn[G2] = nn;
        ht[G2] = (uint16_t)tt;
        G2++;
      }
    }
    g->G2 = G2;
    e1 = pool_malloc((void **)&g->d_n2, sizeof(uint32_t) * (G2 ? G2 : 1));
    if (e1 == hipSuccess) e2 = pool_malloc((void **)&g->d_t2, sizeof(uint16_t) * (G2 ? G2 : 1));
    if (e1 == hipSuccess && e2 == hipSuccess && G2) {
      e1 = hipMemcpy(g->d_n2, hn, sizeof(uint32_t) * G2, hipMemcpyHostToDevice);
      e2 = hipMemcpy(g->d_t2, ht, sizeof(uint16_t) * G2, hipMemcpyHostToDevice);
    }
  }
  free(hn);
  free(ht);
  if (e1 != hipSuccess || e2 != hipSuccess) return fail("stb_groups_aterms: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  // partial sums: at most (column blocks of 64) x 16 waves per table
  g->dotp_elems = (size_t)g->Dmax * ((size_t)(M + 63) / 64 + 1) * 16;
  HIPCHK(pool_malloc((void **)&g->d_dotp, sizeof(double) * g->dotp_elems));
  HIPCHK(hipGetLastError());
  g->fused_ready = 1;
  return 0;
}

extern "C" int stb_groups_aterms_timed(stb_groups_t *g, const double *x_host, int D,
                                       double *out_host, float *ms_fill, float *ms_sweep,
                                       float *ms_terms) {
  STB_ENTRY;
  if (!g) return fail("stb_groups_aterms: null group set");
  if (D < 1 || D > g->Dmax) return fail("stb_groups_aterms: D=%d outside 1..%d", D, g->Dmax);
  double h[2 * STB_TERMS_DMAX];
  const int v = stb_default_variant();
  // one discount: the gather over a stored table is cheap and needs no set-up; a grid: fused
  const bool fuse = g->fused && D >= 2 && (v == STB_FILL_SCALED || v == STB_FILL_CHAIN);
  if (fuse && !g->fused_ready) {
    if (env_int("STB_ATERMS_SPARSE", 1) && groups_fused_setup_sparse(g)) return 1;
    if (!g->fused_ready && groups_fused_setup(g)) return 1;
  }
  HIPCHK(hipEventRecord(g->ev[0], g->st));
  if (fuse) {
    // the chain form as a DOT kernel: sum over table cells of count * log S, no table in memory;
    // then the few pairs that address no cell (t = 1 -> S1, t = n -> 0, out of bounds -> -inf)
    dot_request req;
    if (g->sparse) {
      req.item_ptr = g->d_item_ptr;
      req.ent_pos = g->d_ent_pos;
      req.ent_cnt = g->d_ent_cnt;
      req.nsg = g->nsg;
    } else {
      req.cnt = g->d_cnt;
    }
    req.dotp = g->d_dotp;
    g_dot_req = &req;
    const int rc = stb_fill_S(x_host, D, g->N, g->M, g->d_tables, g->tstride, g->d_S1, g->N, g->d_ws_fill,
                              g->ws_fill, STB_FILL_CHAIN, g->st);
    g_dot_req = nullptr;
    if (rc) return 1;
    if ((size_t)D * req.parts_per_table > g->dotp_elems) return fail("stb_groups_aterms: partial-sum buffer too small");
    HIPCHK(hipEventRecord(g->ev[1], g->st));
    if (stb_sweep_S(g->d_tables, g->tstride, g->d_S1, g->N, D, g->N, g->M, g->d_n2, g->d_t2, g->G2,
                    g->d_out, g->d_ws_sweep, g->ws_sweep, g->st))
      return 1;
    hipLaunchKernelGGL(k_dot_reduce, dim3(D), dim3(64), 0, g->st, g->d_dotp, req.parts_per_table, g->d_out);
  } else {
    if (stb_fill_S(x_host, D, g->N, g->M, g->d_tables, g->tstride, g->d_S1, g->N, g->d_ws_fill,
                   g->ws_fill, v, g->st))
      return 1;
    HIPCHK(hipEventRecord(g->ev[1], g->st));
    if (stb_sweep_S(g->d_tables, g->tstride, g->d_S1, g->N, D, g->N, g->M, g->d_n, g->d_t, g->G,
                    g->d_out, g->d_ws_sweep, g->ws_sweep, g->st))
      return 1;
  }
  HIPCHK(hipEventRecord(g->ev[2], g->st));
  if (stb_restaurant_terms(x_host, D, g->d_T, g->d_bpar, (uint64_t)g->I, g->d_out + g->Dmax,
                           g->d_ws_terms, g->ws_terms, g->st))
    return 1;
  HIPCHK(hipEventRecord(g->ev[3], g->st));
  HIPCHK(hipMemcpyAsync(h, g->d_out, sizeof(double) * 2 * g->Dmax, hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  if (stb_fill_status()) return 1;
  for (int d = 0; d < D; d++) out_host[d] = h[g->Dmax + d] + h[d];
  if (ms_fill) HIPCHK(hipEventElapsedTime(ms_fill, g->ev[0], g->ev[1]));
  if (ms_sweep) HIPCHK(hipEventElapsedTime(ms_sweep, g->ev[1], g->ev[2]));
  if (ms_terms) HIPCHK(hipEventElapsedTime(ms_terms, g->ev[2], g->ev[3]));
  return 0;
}

extern "C" int stb_groups_aterms(stb_groups_t *g, const double *x_host, int D, double *out_host) {
  return stb_groups_aterms_timed(g, x_host, D, out_host, nullptr, nullptr, nullptr);
}

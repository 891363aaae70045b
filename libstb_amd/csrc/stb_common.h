// stb_common.h -- what the HIP translation units of libstb_amd share: error plumbing, the guard
// around libc's rand() state, the buffer cache, the arguments of the fill kernels, and the device
// helpers for cell arithmetic, table logs and deterministic double-double sums.
//
//   abi.hip          error text, device selection, buffer cache, memory helpers, layout exports
//   fill.hip         stb_fill_S / stb_fill_V: choice of form, workspace, status, per-launch timing
//   fill_chain.hip   k_fill_chain (default S fill, also the fused aterms sum), k_fillv_chain (V)
//   fill_hb.hip      k_fill_hb (halo-block spine + tile workers: the default fill; floats and V ratios too)
//   grid_hb.hip      k_grid_hb (the fused grid aterms whose walking waves sum their own strips)
//   fill_pc.hip      k_fill_pc (launch-per-128-rows form: many tables, and the fallback), k_s1
//   fill_rows.hip    k_fill_rows: the reference's own operation order (log domain / V ratios)
//   sweep_terms.hip  k_lookup, k_to_float, k_sweep_partial, k_terms_partial, reductions
//   groups.hip       stb_groups_*: device-resident (n,t) pairs and the aterms evaluation
//   tools/ablation/ablation.hip   superseded fill forms, in a library of their own (make -C tools/ablation)
#ifndef STB_COMMON_H
#define STB_COMMON_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "stb_layout.h"
#include "../../include/stb_hip.h"

// ------------------------------------------------------------------------------------------------
// host side

int stb_fail(const char *fmt, ...);  // records the message for stb_last_error(), returns 1

#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return stb_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// The HIP runtime draws from libc's rand() while it initialises (first stream, first module load:
// observed on ROCm 7.2), which would shift the rand() stream the caller's ARMS sampler is about to
// consume (reference lib/arms.c:913-918).  Every entry point that can reach the runtime runs with
// rand()'s state swapped to a private buffer (glibc: rand() and random() share the state that
// initstate/setstate switch) and puts the caller's state back on exit.  That state is one per
// process, so the swap is too: entry points nest and come from several threads at once, hence a
// depth count under a lock that is held for the count only -- NOT for the call -- and a buffer that
// is not on anybody's stack; the state changes hands at depth 0 <-> 1.  (A thread of the caller that
// draws from rand() while another is inside the library draws from the private state: libc's rand()
// has one state per process, and the reference's samplers are single-threaded for the same reason.)
struct stb_rand_guard {
  stb_rand_guard();
  ~stb_rand_guard();
  stb_rand_guard(const stb_rand_guard &) = delete;
  stb_rand_guard &operator=(const stb_rand_guard &) = delete;
};
#define STB_ENTRY stb_rand_guard stb_rand_guard_

// buffer cache (abi.hip): kind 0 device memory, 1 pinned host memory
hipError_t stb_pool_malloc(void **out, size_t bytes, int kind = 0);
bool stb_pool_free(void *p);  // false if p did not come from the cache

// the device this thread's calls go to (stb_set_device / STB_DEVICE), made current
int stb_use_device(void);

// 128-entry {1/c, -log(1/c)} table of the stored log, on the current device
int stb_logtab(const double2 **out);

static inline size_t stb_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int stb_env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  if (!s || !*s) return dflt;
  return atoi(s);
}

// ------------------------------------------------------------------------------------------------
// fill kernels: common arguments

struct fill_args {
  const double *a;    // [D] discounts (device)
  double *tables;     // D slabs (S or V layout)
  uint64_t tstride;   // elements between slabs
  double *S1;         // D vectors of N (S modes only)
  uint64_t s1stride;
  double *fm;         // frontier mantissas / plain values: [D][2][W]   (launch-per-row-block forms)
  int *fe;            // frontier exponents:                 [D][2][W]
  const double2 *lt;  // log table (stb_logtab)
  unsigned W;         // frontier row pitch (>= M+2)
  unsigned N, M;
  int R;              // rows advanced per launch
  int H;              // halo columns (>= R, multiple of C)
  int Wv;             // owned columns per strip = 64*C - H
};

// what a fill left behind to be checked: the header of a one-launch form (null for the forms that
// cannot give up), and what is needed to repeat the fill with k_fill_pc
struct last_fill {
  unsigned *hdr = nullptr;  // [0] ticket, [1] error code, [2] error detail
  fill_args A;
  int D = 0;
  bool s_table = false, can_fall_back = false;
  bool fell_back = false;   // out of stb_fill_status_of: the fill gave up and was repeated with k_fill_pc
  bool stamped = false;     // the header carries the launch's start and end stamps (STB_HDR_T0 / _T1)
  hipStream_t st = nullptr;
};
void stb_fill_last(last_fill *out);       // the calling thread's last fill
int stb_fill_status_of(last_fill *lf);    // waits for it on its stream; 0, or 1 with stb_last_error() set

// per-launch timing (fill.hip): when armed, a launch gets a begin/end event pair
void stb_prof_events(hipEvent_t *e0, hipEvent_t *e1);
#define STB_LAUNCH_SHM(KERN, GRID, BLOCK, SHM, ST, ...)                                 \
  do {                                                                                  \
    hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                          \
    stb_prof_events(&pe0_, &pe1_);                                                      \
    if (pe0_)                                                                           \
      hipExtLaunchKernelGGL(KERN, GRID, BLOCK, SHM, ST, pe0_, pe1_, 0, __VA_ARGS__);    \
    else                                                                                \
      hipLaunchKernelGGL(KERN, GRID, BLOCK, SHM, ST, __VA_ARGS__);                      \
  } while (0)
#define STB_LAUNCH(KERN, GRID, BLOCK, ST, ...) STB_LAUNCH_SHM(KERN, GRID, BLOCK, 0, ST, __VA_ARGS__)

// rows per renormalisation period of the block-floating forms that start a period at 2^-700
int stb_period_rows(unsigned N);

// a GPU shared with other processes (abi.hip): true -> the forms without waits between workgroups are taken;
// stb_note_span: what a one-launch form's stamps say it took against what its geometry should take
bool stb_shared_gpu();
void stb_note_span(double span_ms, double expect_ms, const char *what);
// header words of the one-launch forms that carry the stamps (100 MHz ticks, 64 bits each): start at word 6, end at word 8
#define STB_HDR_T0 6
#define STB_HDR_T1 8

// the forms (each in its own translation unit); all return 0 or stb_fail(...)
struct dot_request {  // set by stb_groups_aterms around its fill: the chain form sums count * log S
  const unsigned *cnt = nullptr;           // dense: occurrence count per cell, table layout
  const unsigned *item_ptr = nullptr;      // sparse: CSR over (trip, slice) items
  const unsigned short *ent_pos = nullptr; //         row-in-trip << 6 | column-in-slice
  const unsigned *ent_cnt = nullptr;       //         occurrence count
  unsigned nsg = 0;                        //         slices per trip in item_ptr
  int col0 = 1;                            //         first column of slice 0: 1 (k_fill_chain), 2 (k_fill_ck); 3: k_fill_hb's tiles;
                                           //         4: k_fill_hb's strips (the spine sums), lists built for geom_*
  int geom_C = 0, geom_R = 0, geom_G = 0;  //         col0 = 4: columns per lane, rows per block, rows per group of the lists
  const unsigned *tile_off = nullptr;      //         col0 = 4: first tile of every strip (device, stb_grid_tile_offsets)
  const unsigned *dense = nullptr;         //         col0 = 4: the listed cells as words per lane (position | count << 13), group after group
  const unsigned *tinfo = nullptr;         //         col0 = 4: per tile, first word / 64 << 6 | words per group (63: the CSR lists)
  const unsigned *jobs = nullptr;          //         col0 = 4: tiles left to helper waves, strip | block << 16, in the order they become ready
  const unsigned *tjob = nullptr;          //         col0 = 4: per tile its place in `jobs`, or 0xffffffff
  const unsigned *parts_extra_dev = nullptr;  // out: device word with further partial sums per table (the grid form's helper jobs), or null
  double *dotp = nullptr;                  // partial sums [D][parts_per_table]
  int parts_per_table = 0;                 // out
  size_t dotp_cap = 0;                     // in: doubles `dotp` holds; a launch whose partial sums would not fit is refused BEFORE anything is queued
  size_t ws_zero = 0;                      // in: bytes at the start of the workspace the caller knows to be zero (no memset then)
  size_t zero_bytes = 0;                   // out: bytes at the start of the workspace the launch needs zero (and dirties)
  int no_s1 = 0;                           // in: nobody reads the S1 vector (column 1 is among the listed cells)
};
void stb_set_dot_request(const dot_request *r);  // for the next stb_fill_S of this thread
size_t stb_chain_workspace(unsigned N, unsigned M, int D);
int stb_launch_chain(fill_args &A, int D, char *ws, size_t ws_left, const dot_request *dot, unsigned **hdr_out,
                     hipStream_t st);
int stb_launch_vchain(fill_args &A, int D, char *ws, size_t ws_left, unsigned **hdr_out, hipStream_t st);
int stb_chain_tuning(unsigned N, unsigned M, int D, int *W_out);  // columns per strip
// checkpointed form (tools/ablation/fill_ck.hip: spine + tile workers, one launch; superseded by the halo-block form and
// absent from the default build: weak, like the other ablation entry points -- test stb_launch_ck before any of them)
bool stb_ck_eligible(unsigned N, unsigned M, int D) __attribute__((weak));
size_t stb_ck_workspace(unsigned N, unsigned M, int D) __attribute__((weak));
int stb_ck_tuning(unsigned N, unsigned M, int D, int *W_out, int *rows_out) __attribute__((weak));  // columns of a wave strip, rows of a tile
int stb_launch_ck(fill_args &A, int D, char *ws, size_t ws_left, const dot_request *dot, unsigned **hdr_out, hipStream_t st)
    __attribute__((weak));
unsigned stb_ck_dot_parts(unsigned N, unsigned M, int D) __attribute__((weak));  // partial sums per table of the summing form
unsigned stb_ck_dot_spine(unsigned N, unsigned M, int D) __attribute__((weak));  // spine workgroups it launches for D tables

int stb_cu_count();  // compute units of the current device
// halo-block form (fill_hb.hip): a spine that walks blocks of rows alone + tile workers, one launch
bool stb_hb_eligible(unsigned N, unsigned M, int D);
bool stb_hb_eligible_out(unsigned N, unsigned M, int D, int out_kind);
size_t stb_hb_workspace(unsigned N, unsigned M, int D);
unsigned stb_hb_spine(unsigned N, unsigned M, int D);  // spine workgroups of a fill of D tables
int stb_hb_tuning(unsigned N, unsigned M, int D, int *W_out, int *rows_out);  // own columns of a strip, rows of a block
// out_kind: 0 log S (double), 1 log S (float), 2 V = S^n_m / S^n_{m-1} (double), 3 V (float); A.tables is the slab of that type
int stb_launch_hb(fill_args &A, int D, char *ws, size_t ws_left, const dot_request *dot, unsigned **hdr_out, hipStream_t st, int out_kind = 0);
struct hb_dot_info {  // the tiles of the summing halo-block form (cell lists: item = record index * NQ + group of 4 rows)
  int R, UC, HC, NB, JW, NQ;   // rows of a block, own columns of a strip, halo columns, blocks, strips, groups per item base
  int G, C;                    // rows of a group, columns per lane
  unsigned n_tiles, n_rec, n_spine;
  const unsigned *rec_off;     // device: first record of strip index s = j + 1 (s = 0: the halo of strip 0)
};
int stb_hb_sum_C(unsigned N, unsigned M, int Dmax);  // columns per lane of the summing form's strips for a set of up to Dmax discounts
int stb_hb_dot_info(unsigned N, unsigned M, int D, hb_dot_info *out, int sum_C);

// the fused grid evaluation in which the walking waves sum their own strips' listed cells (grid_hb.hip)
struct grid_geom {
  int C, P, B, JW, NB, R, HL, U;  // columns per lane, strips per workgroup, workgroups per table, strips, blocks, rows per block, halo / own lanes
  int G, NQ, phases;              // rows per group, groups per block, launches
  int K;                          // every K-th row of a group is staged in LDS
  unsigned n_tiles;               // (strip, block) pairs of a table
  unsigned job_cap;               // tiles of a table at most whose listed cells are left to waves with nothing to do yet (0: none)
  size_t off_cke, off_ckv, off_state_e, off_state_v, zero_bytes, bytes;
  size_t off_jflag, off_jrec_e, off_jrec_v;  // [D][job_cap] "written" words (zeroed), [D][job_cap][64] exponents, [D][job_cap][64 C] significands
  int ok;
};
int stb_grid_geometry(unsigned N, unsigned M, int D, grid_geom *out);
size_t stb_grid_workspace(unsigned N, unsigned M, int D);
unsigned stb_grid_job_cap(int C, int D, unsigned n_tiles, int phases);
int stb_launch_grid(fill_args &A, int D, char *ws, size_t ws_left, const dot_request *dot, unsigned **hdr_out, hipStream_t st);

int stb_launch_pc(fill_args &A, int D, hipStream_t st);
int stb_launch_s1(const fill_args &A, int D, hipStream_t st);
// The discounts of a fill on their way to A.a (device): a copy of a few bytes from pageable host memory is a 4.4 us copy
// kernel of its own behind a staging buffer; up to 64 discounts travel as kernel arguments instead -- of k_prep when the
// launcher has one (stb_a_take), else of a kernel of their own (stb_a_flush; no-ops when nothing is on its way).
struct stb_a64 { double v[64]; };
void stb_a_defer(const double *a_host, int D);
bool stb_a_take(stb_a64 *out, int *D_out);
int stb_a_flush(const fill_args &A, hipStream_t st);
int stb_launch_prep(const fill_args &A, int D, void *ws, size_t zero_bytes, hipStream_t st);  // k_s1 + a zeroed workspace (+ the discounts on their way), one launch
#define STB_ROWS_LOGDOM 1
#define STB_ROWS_VRATIO 2
int stb_launch_rows(fill_args &A, int D, int C, int what, hipStream_t st);
// superseded forms (ablation.hip); absent from the default build
#define STB_HAVE_ABLATION_SYM stb_ablation_fill
extern "C" int stb_ablation_fill(fill_args &A, int D, int variant, char *ws, size_t ws_left, unsigned **hdr_out,
                                 hipStream_t st) __attribute__((weak));
extern "C" size_t stb_ablation_workspace(unsigned N, unsigned M, int D) __attribute__((weak));

// ------------------------------------------------------------------------------------------------
// device side
#if defined(__HIPCC__)

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
// rotated instead: lane 0 receives lane 63's value (DPP wave_ror:1)
__device__ __forceinline__ double wave_ror1(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x13C, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x13C, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shr1(double v, double fill) {
  int lo = wave_shr1(__double2loint(v), __double2loint(fill));
  int hi = wave_shr1(__double2hiint(v), __double2hiint(fill));
  return __hiloint2double(hi, lo);
}

// the 32-bit LDS address of a __shared__ object, and a store of two doubles to `addr` + 8*OFF and
// `addr` + 8*(OFF+1) whose sources may be any two register pairs
typedef __attribute__((address_space(3))) double stb_lds_double;
__device__ __forceinline__ unsigned lds_addr_of(double *p) { return (unsigned)(uintptr_t)(stb_lds_double *)p; }
template <int OFF>
__device__ __forceinline__ void lds_store2(unsigned addr, double x, double y) {
  static_assert(OFF >= 0 && OFF + 1 <= 255, "ds_write2_b64 offsets are 8 bits");
  asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(x), "v"(y), "n"(OFF), "n"(OFF + 1)
               : "memory");
}

// A store of one double per lane at (wave-uniform base) + (per-lane unsigned byte offset): the base
// goes to scalar registers and the address costs no vector instruction (the compiler's own choice for
// base[lane_index] is one or two 64-bit vector adds per store).  The base passes through an s_mov
// inside the statement: a vector-memory instruction that reads a scalar register within five cycles of
// a VECTOR instruction writing it (v_readfirstlane) gets the old value, the compiler does not look
// into an asm statement for that hazard, and a scalar instruction in between is interlocked.
__device__ __forceinline__ void store_sbase(const void *sbase, unsigned byte_off, double val) {
  unsigned long long base_copy;
  asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx2 %1, %2, %0"
               : "=&s"(base_copy)
               : "v"(byte_off), "v"(val), "s"(sbase)
               : "memory");
}
// high word of the double with the mantissa of `hi` and the exponent of 1.0: (hi & 0xfffff) | 0x3ff00000
// in one instruction (v_bfi_b32 takes one scalar operand: the 1.0 pattern comes in a vector register)
__device__ __forceinline__ int mantissa_of_one(int hi, int one_hi_vgpr) {
  int r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(0x000fffff), "v"(hi), "v"(one_hi_vgpr));
  return r;
}

// log(v 2^ep) from the bits of v: exponent field + 7 leading mantissa bits index a 128-entry table
// {1/c, -log(1/c)} (held in LDS by the callers), then a degree-5 polynomial in r = z/c - 1,
// |r| < 2^-8 (the construction used by table-driven libm logs).  Absolute error a few 1e-16.
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

// LDS-only barrier: waits for this wave's LDS traffic, NOT for its global stores (a __syncthreads()
// would add s_waitcnt vmcnt(0) and stall every consumer on its stores' round trip once per trip)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// significands of the producer/consumer and chain forms start a period at 2^-PC_BIAS * [0.5,1)
#define PC_BIAS 700
#define STB_EZ (-(1 << 28))  // exponent standing for an exact zero in a frontier

// ---- deterministic double-double sums ----
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
// reduce one dd per thread over a block of up to 256 threads in a fixed order; valid in thread 0
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
#endif  // __HIPCC__

#endif

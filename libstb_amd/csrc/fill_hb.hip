// fill_hb.hip -- the halo-block form of the S-table fill: ONE launch; a spine that walks the rows
// without talking to anybody inside a block of rows, and tile workers that do everything else.
//
// Replaces the table part of S_remake_part's double-S branch (reference lib/stable.c:321-388):
//   S^n_m = (n-1-m a) S^{n-1}_m + S^{n-1}_{m-1},   stored as log S^n_m for 2 <= m <= min(n-1, M).
//
// Why another form.  In k_fill_ck a spine wave hands the last column of its strip to its right
// neighbour row by row: the post, the look at the left neighbour's counter and the bookkeeping of a trip
// cost a lone wave more than the arithmetic of the row (42-49 ns a row of which 17-26 are the recurrence),
// and that row time, times N, is the floor of a fill.  Information moves one column to the right per
// row, so a wave that starts a block of R rows with R extra columns on its left -- a halo, copied from
// its left neighbour's state at the start of the block -- needs nothing from anybody until the block
// ends: the halo goes wrong from the left, one column per row, and the wave's own columns stay exact.
//
//   spine    a wave owns U = 64 - HL lanes (C adjacent columns each, block-floating, see k_fill_chain) of
//            one table for all rows; lanes 0..HL-1 are the halo, HL = R / C.  Per block: renormalise,
//            hand the rightmost HL lanes to the right neighbour (LDS inside a workgroup; the record below
//            across workgroups, polled by the right workgroup's fetcher wave), store the own lanes as the
//            block's RECORD (significands + lane exponents, HBM), take the halo from the left, then R rows
//            of 2 DPP moves, 1 multiply, C fma and C adds each, and nothing else.
//   workers  every other wave of the grid.  A worker takes a tile (table d, strip j, block b) from a
//            ticket, loads the strip's record of the block and -- for its halo lanes -- the left strip's,
//            recomputes the R rows in registers exactly as the spine does, and converts and stores the
//            own lanes' cells row by row.  Tiles are independent of each other and of the spine's pace.
//
// Lanes hold aligned groups of C table elements (element e = m - 2): a lane's cells are one aligned
// 8*C-byte store.  Column 1 (the S1 vector, not a table column) is the last element of group -1, i.e. of
// the last halo lane of strip 0, whose halo needs no neighbour (column 0 is identically zero): strip 0
// computes it along.  The state before block b is row 1 + b R; row 1 is S^1_1 = 1.
//
// Round 6: what lies between two blocks' rows on a spine wave (hand-over, record, halo, renormalisation) was ~125
// instructions and ~840 cycles beside the rows' ~1500; the block loop below (HB_LEAN) does it in ~60 and ~440 -- nothing
// masked, one LDS word per lane for all signalling, the halo asked for a record's stores ahead of its use, the
// renormalisation shift taken four rows before the block's end.  MEASUREMENTS.md section R6.1.
//
// A record word is its own flag: 0 means "not written yet" (significands, which are never negative, travel
// with the sign bit set, exponents carry an offset); records are written with write-through stores and read with L1-bypassing loads.
// Every wait is bounded; on expiry the waiter records an error in the header and everybody runs to the
// end (stb_fill_status repeats the fill with k_fill_pc).

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "fill_chain.h"

#ifndef HB_NW
#define HB_NW 8          // waves per workgroup: up to HB_PMAX spine waves and a fetcher
#endif
#define HB_PMAX 7
#define HB_SLOTS 8       // ring of hand-overs between two spine waves of a workgroup (blocks); looked after every 4th block
#define HB_FSLOTS 8      // ring of hand-overs from the fetcher to spine wave 0
#define HB_MAXHL 32      // halo lanes at most
// ... per strip shape: blocks have at most 48 rows with 2 or 4 columns per lane (+ a lane for the V table), 32 with one -- what
// the hand-over rings in LDS are sized by (4 columns: 23 KB instead of 57)
__host__ __device__ constexpr int hb_mhl(int C) { return C == 1 ? HB_MAXHL : (C == 2 ? 25 : (C == 3 ? 17 : 13)); }
#ifndef HB_NW_DOT4
#define HB_NW_DOT4 14
#endif                   // waves per workgroup of the summing form with 4 columns per lane: its tile workers are what a grid of 8-28
                         // discounts waits for, and a workgroup has its compute unit alone (LDS) -- four more of them on it
#define HB_EOFF32 (1u << 30)
#define HB_WRITTEN 0x8000000000000000ull  // a record's significands (never negative) travel with the sign bit set
#define HB_DOT_GR 8        // rows of a group of the summing form: its listed cells are looked up together, one per lane and pass
                           // (4 rows: 12 passes a tile, each with its LDS round trips exposed -- a worker wave has one
                           // partner on its SIMD -- and a quarter of the lanes at work; 8 rows: 6 fuller passes)
#define HB_DOT_NQ 6        // groups in a block at most (48 rows)
#define HB_MAXCNT 64        // ticket counters at most
#define HB_CNT0 64          // word of the header the ticket counters start at (one per 32 words)
#define HB_HDR_BYTES (256 + HB_MAXCNT * 128)
#define HB_RECOFF_LDS 512   // strips + 2 whose record offsets are copied to LDS
#define HB_ORDER_LDS 8192  // tiles of a table whose order list is copied to LDS (32 KB)
#define HB_PROG_STRIDE 32  // words between two strips' progress words: a line each (a strip's waiting workers poll
                           // the word its spine wave writes; 125 strips' words in four lines made those lines the
                           // busiest of the chip and the spine's stores to them slow: its store queue filled up)
#define HB_SPIN 48         // looks at a neighbour's counter, ~0.1 us apart, before the out-of-line wait
// timeline stamps of the spine: the 100 MHz wall clock, or (diagnostic build -DHB_TL_CYCLES) the shader clock,
// which tells a slower clock from waiting
#ifdef HB_TL_CYCLES
#define HB_STAMP() ((unsigned long long)clock64())
#else
#define HB_STAMP() ((unsigned long long)wall_clock64())
#endif
#ifndef HB_LTX
#define HB_LTX 0           // 1: the storing kernels hold the log table 16 times in LDS (no bank conflicts), 0: once.  Measured
                           // (MI355X, tools/ab_lib.py, N = M = 10^4, medians): 8 tables 0.772 / 0.813 ms with either, 16 tables
                           // 1.446 against 1.444, one table 0.332 against 0.325 (32 KB more to set up per workgroup): the
                           // storing workers wait for their stores, not for LDS -- left off
#endif
#ifndef HB_DIAG
#define HB_DIAG 0          // diagnostic builds: 1 the spine stores no records, 2 it stores every record twice, 8 the workers store nothing, 16 summing workers look nothing up, 32 ... and stage nothing (results wrong)
#endif

struct hb_args {
  unsigned *hdr;               // [0] role ticket, [1] error code, [2] error detail, [3] tile ticket, [4] spine waves through
  unsigned long long *ck_v;    // [D][n_rec][U*C]  records: significands of the own lanes at the start of a block
  unsigned *ck_e;              // [D][n_rec][U]    ... and their lane exponents + HB_EOFF32
  unsigned *progress;          // [D][JW][HB_PROG_STRIDE]  blocks whose record a spine wave has written (scheduling hint)
  const unsigned *order;       // [n_order] j | b << 16 (split: j | quarter << 14 | b << 16), in the order the tiles become ready
  const unsigned *rec_off;     // [JW + 2]  first record of strip s (s = j + 1; s = 0: the halo of strip 0), per table
  unsigned n_rec;              // records per table
  int D, B, JW, NB;            // tables, spine workgroups per table, strips per table, blocks
  int P, R, HL, U;             // spine waves per workgroup, rows per block, halo lanes, own lanes
  unsigned n_tiles;            // per table
  unsigned n_order;            // entries of `order`: the tiles, those of a strip's last `split` blocks four times (a quarter of the rows each)
  int split;                   // see hb_order_list; 0: no tile is split (and j has 16 bits in `order`)
  unsigned n_spine;            // spine workgroups in all (B * D)
  unsigned long long timeout;  // wall_clock64 ticks a wait may last
  int poll_nap;                // s_sleep argument between two polls of a fetcher
  int nap_block;               // s_sleep argument of a worker per block its inputs are away
  int diag;                    // STB_HB_DIAG: 1 the workers wait for the whole spine
  int n_cnt;                   // ticket counters: a multiple of D
  int order_lds;               // 1: the tile order fits the dynamic LDS segment
  int spare_work;              // 1: waves of a spine workgroup that have no strip work on tiles meanwhile
  int doze;                    // 1: a spine wave sleeps until its left neighbour in the workgroup reaches its first block
  // DOT kernels (aterms without a table, lib/samplea.c:68-80): the cells that occur among the (n,t) pairs, grouped
  // per item = (record index of the tile) * HB_DOT_NQ + (group of HB_DOT_GR rows of the block), and where the sums go
  const unsigned *item_ptr;        // [n_rec * HB_DOT_NQ + 1] first entry of every item
  const unsigned short *ent_pos;   // row-in-group << 8 | element of the wave (halo included) of each occurring cell
  const unsigned *ent_cnt;         // its occurrence count
  double *dotp;                    // [D][n_tiles] sum of count * log S per tile
  unsigned long long *dbg;     // STB_HB_TIMELINE: wall-clock stamps, table 0: [JW][NB + 2] spine (start, block starts, end),
                               // then [n_tiles][4] workers (claimed, inputs loaded, done, hardware id)
};

typedef double hb_double2 __attribute__((ext_vector_type(2)));

// aligned 16-byte store at (wave-uniform base) + (per-lane byte offset); see store_sbase
__device__ __forceinline__ void hb_store16(const void *sbase, unsigned byte_off, double x, double y) {
  unsigned long long base_copy;
  hb_double2 v2 = {x, y};
  asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, %0\n\ts_nop 1"
               : "=&s"(base_copy)
               : "v"(byte_off), "v"(v2), "s"(sbase)
               : "memory");
}

// two record words with one write-through store (what __hip_atomic_store at agent scope compiles to, 16 bytes wide)
__device__ __forceinline__ void hb_store_wt16(unsigned long long *p, unsigned long long a, unsigned long long b) {
  typedef unsigned long long hb_u64x2 __attribute__((ext_vector_type(2)));
  const hb_u64x2 v2 = {a, b};
  // (s_nop: a store of more than 8 bytes reads its data a cycle or two after it issues, and the compiler does
  // not look into an asm statement for the VALU write of those registers that may follow)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v2) : "memory");
}

// the record of a spine block: C words per lane (each with the sign bit set) behind `p`, written through, by the lanes of
// `mask` only -- the mask goes into exec and comes out again inside the statement: no branch round the stores (the
// compiler's own masked region moved them out of line, two taken branches a block on the lone wave's path)
template <int C>
__device__ __forceinline__ void hb_store_record(unsigned long long mask, unsigned long long *p, const double (&v)[C]) {
  typedef unsigned long long hb_u64x2 __attribute__((ext_vector_type(2)));
  unsigned long long saved;
  if constexpr (C == 1) {
    const unsigned long long a = (unsigned long long)__double_as_longlong(v[0]) | HB_WRITTEN;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx2 %2, %3, off sc1\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "s"(mask), "v"(p), "v"(a) : "memory");
  } else if constexpr (C == 2) {
    const hb_u64x2 a = {(unsigned long long)__double_as_longlong(v[0]) | HB_WRITTEN, (unsigned long long)__double_as_longlong(v[1]) | HB_WRITTEN};
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx4 %2, %3, off sc1\n\ts_nop 0\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "s"(mask), "v"(p), "v"(a) : "memory");
  } else if constexpr (C == 3) {
    const hb_u64x2 a = {(unsigned long long)__double_as_longlong(v[0]) | HB_WRITTEN, (unsigned long long)__double_as_longlong(v[1]) | HB_WRITTEN};
    const unsigned long long c = (unsigned long long)__double_as_longlong(v[2]) | HB_WRITTEN;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx4 %2, %3, off sc1\n\tglobal_store_dwordx2 %2, %4, off offset:16 sc1\n\ts_nop 0\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "s"(mask), "v"(p), "v"(a), "v"(c) : "memory");
  } else {
    static_assert(C == 4, "columns per lane");
    const hb_u64x2 a = {(unsigned long long)__double_as_longlong(v[0]) | HB_WRITTEN, (unsigned long long)__double_as_longlong(v[1]) | HB_WRITTEN};
    const hb_u64x2 c = {(unsigned long long)__double_as_longlong(v[2]) | HB_WRITTEN, (unsigned long long)__double_as_longlong(v[3]) | HB_WRITTEN};
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx4 %2, %3, off sc1\n\tglobal_store_dwordx4 %2, %4, off offset:16 sc1\n\ts_nop 0\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "s"(mask), "v"(p), "v"(a), "v"(c) : "memory");
  }
}

// ---- the log of a block-floating cell, eight cells at a time, stage-major (as in k_fill_chain) ----
// (LTX: the table is held 16 times, entry e of copy c at lt[e * 16 + c], and a lane reads copy lane & 15: the sixteen lanes
// a 16-byte LDS read serves per cycle then sit on sixteen different slots whatever their entries -- no bank conflicts,
// where 64 random entries of ONE 2 KB table cost a storing worker 56 % of its LDS cycles, profiles/r03_sq_counters_fill8.txt)
template <bool LTX>
__device__ __forceinline__ void hb_logs8(const double (&x)[8], const int (&ep8)[8], const double2 *lt, int one_hi, double (&val)[8]) {
  double z[8], kf[8], r[8], pl[8];
  double2 tt[8];
  const int l15 = LTX ? (int)(threadIdx.x & 15) : 0;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int e = (__double2hiint(x[u]) >> 13) & 127;
    tt[u] = lt[LTX ? (e << 4) | l15 : e];
  }
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int hi = __double2hiint(x[u]);
    z[u] = __hiloint2double(mantissa_of_one(hi, one_hi), __double2loint(x[u]));
    kf[u] = (double)((int)((hi >> 20) & 0x7ff) - 1023 + ep8[u]);
  }
#pragma unroll
  for (int u = 0; u < 8; u++) r[u] = fma(z[u], tt[u].x, -1.0);
#pragma unroll
  for (int u = 0; u < 8; u++) pl[u] = fma(r[u], 0.2, -0.25);
#pragma unroll
  for (int u = 0; u < 8; u++) pl[u] = fma(r[u], pl[u], 1.0 / 3.0);
#pragma unroll
  for (int u = 0; u < 8; u++) pl[u] = fma(r[u], pl[u], -0.5);
#pragma unroll
  for (int u = 0; u < 8; u++) pl[u] = fma(r[u], pl[u], 1.0);
#pragma unroll
  for (int u = 0; u < 8; u++) val[u] = fma(kf[u], 0.693147180559945309417, fma(r[u], pl[u], tt[u].y));
}

// renormalise a lane: the largest of its C significands (none is negative) back to 2^-PC_BIAS * [0.5,1)
template <int C>
__device__ __forceinline__ void hb_renorm(double (&v)[C], int &ep) {
  double vmax = v[0];
#pragma unroll
  for (int i = 1; i < C; i++) vmax = fmax(vmax, v[i]);
  const int sh = (vmax != 0.0) ? __builtin_amdgcn_frexp_exp(vmax) + PC_BIAS : 0;
#pragma unroll
  for (int i = 0; i < C; i++) v[i] = ldexp(v[i], -sh);
  ep += sh;
}

// ... in two steps: the shift (from the lane as it stands) and its application (possibly to the lane a few rows later: any
// shift that keeps the significands in range is as good as any other -- the arithmetic is scale-free and the logs are taken
// from exponent field + lane exponent together)
template <int C>
__device__ __forceinline__ int hb_renorm_shift(const double (&v)[C]) {
  // (v_max_f64 as written: fmax() first quiets its operands, an instruction each -- nothing here is a NaN)
  double vmax = v[0];
#pragma unroll
  for (int i = 1; i < C; i++) asm("v_max_f64 %0, %1, %2" : "=v"(vmax) : "v"(vmax), "v"(v[i]));
  int sh = (vmax != 0.0) ? __builtin_amdgcn_frexp_exp(vmax) + PC_BIAS : 0;
  asm volatile("" : "+v"(sh));  // (here, not where it is used)
  return sh;
}
template <int C>
__device__ __forceinline__ void hb_renorm_apply(double (&v)[C], int &ep, int sh) {
#pragma unroll
  for (int i = 0; i < C; i++) v[i] = ldexp(v[i], -sh);
  ep += sh;
}

// eight rows of the recurrence: lane l takes the last column of lane l - 1 (lane 0: nothing), scaled
// by s = 2^(exponent of lane l-1 - exponent of lane l), frozen for the block
template <int C>
__device__ __forceinline__ void hb_row(double (&v)[C], double (&coef)[C], double s) {
  const double t0 = wave_shr1_zero(v[C - 1]) * s;
#pragma unroll
  for (int i = C - 1; i >= 1; i--) v[i] = fma(coef[i], v[i], v[i - 1]);
  v[0] = fma(coef[0], v[0], t0);
#pragma unroll
  for (int i = 0; i < C; i++) coef[i] += 1.0;
}

// first block of strip j: the state before it has nothing in the strip's own columns (2 + j U C and up)
__host__ __device__ static inline int hb_first_block(int j, int UC, int R) { return (int)(((long long)j * UC) / R); }

// OUT: what the tile workers store -- 0 log S^n_m as double (S_remake_part, lib/stable.c:321-388), 1 the same narrowed to
// float (S_FLOAT, lib/stable.c:389-449: all arithmetic in double, only the stored value is a float), 2 the ratio
// V^n_m = S^n_m / S^n_{m-1} as double (S_UVTABLE, lib/stable.c:451-482: a block-floating cell and its left neighbour are
// one division away from it), 3 that ratio as float (lib/stable.c:483-537).
#define HB_LIKELY(x) __builtin_expect(!!(x), 1)
#define HB_UNLIKELY(x) __builtin_expect(!!(x), 0)
#ifndef HB_LEAN
#define HB_LEAN 1  // 0: the block top as rounds 3-5 had it (a masked region per step, two waits for LDS; kept for A/B runs: make variant DEFS=-DHB_LEAN=0)
#endif
#define HB_EB (HB_LEAN ? HB_EOFF32 : 0u)
#ifndef HB_LEAN_MASK
#define HB_LEAN_MASK 0  // 1: the lean top's hand-over and halo touch LDS with their own lanes only (masked regions again, less LDS work): measured, see MEASUREMENTS R6.1
#endif
#ifndef HB_ABL
#define HB_ABL 0  // timing-only builds (results wrong): the spine leaves out 1 the halo read, 2 the ring store, 4 the renormalisation, 8 the record, 16 the progress word
#endif
template <int C, int DOT, int OUT = 0>
__global__ __launch_bounds__(64 * ((DOT != 0 && C >= 3) ? HB_NW_DOT4 : HB_NW), (DOT != 0 && C >= 3) ? 1 : 2) void k_fill_hb(fill_args A, hb_args X) {
  static_assert(C == 1 || C == 2 || C == 4 || (C == 3 && DOT != 0), "columns per lane (3: the summing form only)");
  static_assert(OUT == 0 || (DOT == 0 && C >= 2), "only the storing fill of 2 or 4 columns per lane narrows or divides");
  constexpr bool VT = (OUT & 2) != 0, FL = (OUT & 1) != 0;
  // a worker lane's groups of adjacent elements (see the workers; a float row of 4 columns per lane is one 16-byte store)
  constexpr int NG = (C == 4 && DOT == 0 && !FL) ? 2 : 1, CG = C / NG;
  constexpr bool LTX = (DOT == 0) && (HB_LTX != 0);  // the storing kernels hold the log table 16 times (see hb_logs8)
  __shared__ double2 lt[LTX ? 128 * 16 : 128];
  // what a spine wave hands to its right neighbour at the start of a block: its rightmost HL lanes
  constexpr int NW = (DOT != 0 && C >= 3) ? HB_NW_DOT4 : HB_NW, MHL = hb_mhl(C);
  // Ring r (r = 0 .. P) is what spine wave r READS its halo from: ring 0 is filled by the fetcher (the left workgroup's last
  // strip), ring w + 1 by spine wave w; ring HB_PMAX + 1 takes the writes of lanes that hand nothing over (no lane is
  // masked off for a hand-over).  c_posted[r]: blocks below it are in ring r; c_taken[r]: blocks below it are in wave r's
  // registers; c_huge: what a lane with nothing to check looks at.
  constexpr int SLOTV = MHL * C, RINGV = HB_SLOTS * SLOTV, RINGE = HB_SLOTS * MHL;
  __shared__ __attribute__((aligned(16))) double rgv[(HB_PMAX + 2) * RINGV];
  __shared__ int rge[(HB_PMAX + 2) * RINGE];
  __shared__ int c_posted[HB_NW + 1], c_taken[HB_NW + 1], c_huge, c_never, s_abort, s_awake;
  static_assert(HB_FSLOTS == HB_SLOTS, "one ring shape");
  __shared__ unsigned s_ticket;
  // dynamic segment.  Summing form: per wave four rows of the wave's 64 C significands; then, in both forms,
  // the tile order when it fits (a ticket then costs no dependent global load).
  extern __shared__ __attribute__((aligned(16))) double hb_dyn[];
  unsigned *s_order = reinterpret_cast<unsigned *>(hb_dyn + (DOT ? (size_t)NW * 4 * 64 * C : 0));
  __shared__ int w_se[DOT ? NW : 1][64];
  __shared__ int s_done[HB_MAXCNT];  // ticket counters this workgroup has found exhausted
  __shared__ unsigned s_recoff[HB_RECOFF_LDS];  // first record of every strip (a tile's record costs no global load)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_ticket = atomicAdd(X.hdr, 1u);
  if constexpr (LTX) {
    for (int i = tid; i < 128 * 16; i += blockDim.x) lt[i] = A.lt[i >> 4];
  } else {
    if (tid < 128) lt[tid] = A.lt[tid];
  }
  const bool order_in_lds = X.order_lds != 0;
  if (order_in_lds)
    for (unsigned i = tid; i < X.n_order; i += blockDim.x) s_order[i] = X.order[i];
  for (int i = tid; i < HB_MAXCNT; i += blockDim.x) s_done[i] = 0;
  const bool recoff_in_lds = X.JW + 2 <= HB_RECOFF_LDS;
  if (recoff_in_lds)
    for (int i = tid; i < X.JW + 2; i += blockDim.x) s_recoff[i] = X.rec_off[i];
  __syncthreads();
  auto rec_off = [&](int sidx) -> unsigned { return recoff_in_lds ? s_recoff[sidx] : X.rec_off[sidx]; };
  const unsigned ticket = s_ticket;
  const unsigned N = A.N, M = A.M;
  const int R = X.R, HL = X.HL, U = X.U, NB = X.NB, P = X.P;
  const int UC = U * C;
  // (when the launch started, for the host's look at what it took: stb_note_span)
  if (ticket == 0 && tid == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(X.hdr + STB_HDR_T0), (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  if (ticket < X.n_spine) {
    // =========================================================================================
    // spine workgroup: strips j*P .. j*P + P - 1 of table d
    const int j = (int)(ticket / (unsigned)X.D);
    const int d = (int)(ticket % (unsigned)X.D);
    const int jw0 = j * P;
    if (tid < HB_NW) {
      const int jw = jw0 + tid;
      const int b0 = (jw < X.JW) ? hb_first_block(jw, UC, R) : NB;
      c_posted[tid + 1] = b0;  // hand-overs for blocks below it are nobody's business
      c_taken[tid] = b0;
    }
    if (tid == 0) {
      c_posted[0] = hb_first_block(jw0, UC, R);  // (the fetcher's)
      c_taken[HB_NW] = 0x7fffffff;
      c_huge = 0x7fffffff;
      c_never = -0x7fffffff;
      s_abort = 0;
      s_awake = (j == 0) ? 1 : 0;
    }
    __syncthreads();
    const unsigned who = (unsigned)(j | (d << 16));
    bool aborted = false;
    // (neighbours reach a block's end within a fraction of a microsecond of each other: a short wait is
    // spun out here -- the out-of-line wait is a call, and a call waits for every store under way)
    auto wait_ge = [&](const int *cnt, int need, unsigned code) {
      if (aborted) return;
      for (int k = 0; k < HB_SPIN; k++) {
        if (lds_peek(cnt) >= need) {
          asm volatile("" ::: "memory");  // (what the counter guards is read or written after it)
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!chain_wait_slow(cnt, need, &s_abort, X.hdr, X.timeout, code, who, 1)) aborted = true;
      asm volatile("" ::: "memory");
    };
    const size_t tab_rec = (size_t)d * X.n_rec;

    if (wave < P) {
      // ================= spine waves =================
      const int w = wave;
      const int jw = jw0 + w;
      if (jw < X.JW) {
        const int b0 = hb_first_block(jw, UC, R);
        const double a = A.a[d];
        const int m0 = 2 + (jw * U - HL + lane) * C;  // first column of the lane (may be <= 0 in the halo of strip 0)
        // (the coefficient n - 1 - m a of the next row is carried along, + 1 a row: the workers start every tile
        // from the closed form instead, which differs from this in the last bits at most)
        double v[C], coef[C];
#pragma unroll
        for (int i = 0; i < C; i++) {
          v[i] = 0.0;
          coef[i] = (double)(1 + b0 * R) - (double)(m0 + i) * a;
        }
        int ep = 1 + PC_BIAS + (int)HB_EB;  // (HB_EB: what the lean spine's exponents carry, in registers and in the rings: a record's own offset)
        if (jw == 0 && lane == HL - 1) v[C - 1] = ldexp(1.0, -1 - PC_BIAS);  // row 1: S^1_1 = 1
        const bool has_next = (w + 1 < P) && (jw + 1 < X.JW);
        const int *left_cnt = &c_posted[w];
        unsigned long long *dbg = (X.dbg && d == 0 && lane == 0) ? X.dbg + (size_t)jw * (NB + 2) : nullptr;
        // own records: strip index jw + 1, blocks from b0; the halo of strip 0: strip index 0, blocks from 0
        // (nothing is loaded from global memory inside the block loop: a load is waited for with vmcnt(0), i.e.
        // together with every write-through store of the records still under way)
        const bool own = lane >= HL;
        const size_t rec_base = own ? tab_rec + rec_off(jw + 1) - (size_t)b0 : tab_rec + rec_off(0);
        const int slot = own ? lane - HL : lane + U - HL;
        unsigned long long *rec_v = X.ck_v + (rec_base * U + slot) * C;
        unsigned *rec_e = X.ck_e + rec_base * U + slot;
        unsigned *prog = X.progress + ((size_t)d * X.JW + jw) * HB_PROG_STRIDE;
        double s = 1.0;
#ifdef HB_TL_FINE
        unsigned long long *fdbg = dbg ? X.dbg + (size_t)X.JW * (NB + 2) + (size_t)X.n_order * 4 + (size_t)jw * NB * 8 : nullptr;
#ifndef HB_FINE_SET
#define HB_FINE_SET 0x7f   // which of the seven stamps are made (a stamp waits for LDS and costs ~100 cycles: fewer stamps, truer blocks)
#endif
#define HB_FINE(k) if (((HB_FINE_SET) >> (k)) & 1) { if (fdbg) fdbg[(size_t)b * 8 + (k)] = HB_STAMP(); }
#else
#define HB_FINE(k)
#endif
#if HB_LEAN
        // ---- the lean block top (round 6).  A block's 48 rows are ~1500 cycles of a lone wave; what the compiler made of
        // the hand-over, the record and the halo between two blocks' rows was ~125 instructions and ~800 cycles more (masked
        // regions with their branches, three waits for LDS, address products).  Here nothing is masked and nothing is
        // multiplied: every lane writes its significands to a ring (the lanes that hand nothing over into a spare one),
        // ONE word per lane carries both counters (lane 63: blocks posted to the right, lane 62: blocks taken from the left
        // -- the latter one block late, which the rings' depth pays for), ONE word per lane is looked at (lane 63: what the
        // left neighbour has posted, lane 61: what the right one has taken) and compared with the lane's own counter, and the
        // halo is asked for straight after the hand-over, a record's stores ahead of its use.  Waves of a workgroup settle a
        // round trip apart (the first look of a wave that is too close fails once and its wait puts it there).
        const bool hand = lane >= U;
        const bool halo = lane < HL && jw > 0;
        const int wv_idx = hand ? (w + 1) * RINGV + (lane - U) * C : (HB_PMAX + 1) * RINGV + (lane % MHL) * C;
        const int we_idx = hand ? (w + 1) * RINGE + (lane - U) : (HB_PMAX + 1) * RINGE + (lane % MHL);
        const int rv_idx = w * RINGV + (lane < HL ? lane : 0) * C;
        const int re_idx = w * RINGE + (lane < HL ? lane : 0);
        int *const post_p = lane == 63 ? &c_posted[w + 1] : (lane == 62 ? &c_taken[w] : &rge[(HB_PMAX + 1) * RINGE + (lane % MHL)]);
        const int *chk_p = (lane == 63 && jw > 0) ? &c_posted[w] : ((lane == 61 && has_next) ? &c_taken[w + 1] : (lane == 60 ? &c_never : &c_huge));
        // (lane 63: b + 1 once block b is posted; lane 62: b, the blocks taken; lane 61: b - 6, what the right neighbour must
        // have taken before the NEXT block's hand-over goes into its slot of the ring of 8; these are never halo lanes)
        // (every other lane counts like lane 63: the halo lanes' b + 1 is the strip's progress word, see the record)
        int vcnt = lane == 62 ? b0 - 1 : (lane == 61 ? b0 - 7 : b0);
        // records: the own lanes' (every lane's in strip 0, whose halo is exact and stands for a left neighbour); the other
        // lanes of a strip write the progress word with the store that carries the exponents
        const bool rec_lane = own || jw == 0;
        const unsigned long long rec_mask = __ballot(rec_lane);
        unsigned long long *rvp = rec_v + (size_t)b0 * (size_t)(U * C);
        unsigned *rep = rec_lane ? rec_e + (size_t)b0 * (size_t)U : prog;
        const unsigned rep_step = rec_lane ? (unsigned)U : 0u;
        const bool tl_on = X.dbg != nullptr;
        auto spine_loop = [&](auto r48) {
        constexpr bool R48 = decltype(r48)::value;
        int b = b0;
        for (;;) {
          HB_FINE(1);
          const int so = b & (HB_SLOTS - 1);
          // ---- hand-over to the right: significands and exponent of every lane, then the counters ----
          if (!HB_LEAN_MASK || hand) {
            double *dst = &rgv[wv_idx + so * SLOTV];
            if constexpr (C == 1) {
              dst[0] = v[0];
            } else if constexpr (C == 3) {
              dst[0] = v[0];
              dst[1] = v[1];
              dst[2] = v[2];
            } else {
#pragma unroll
              for (int i = 0; i < C; i += 2) *reinterpret_cast<hb_double2 *>(dst + i) = hb_double2{v[i], v[i + 1]};
            }
            rge[we_idx + so * MHL] = ep;
          }
          asm volatile("" ::: "memory");
          vcnt += 1;
          __hip_atomic_store(post_p, vcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
          // ---- the halo is asked for: the counters first (LDS serves a wave's requests in order) ----
          int seen = __hip_atomic_load(chk_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
          double hv[C];
          int he = 0;
          if constexpr (HB_LEAN_MASK != 0) {
#pragma unroll
            for (int i = 0; i < C; i++) hv[i] = 0.0;
          }
          if (!HB_LEAN_MASK || lane < HL) {
            const double *src = &rgv[rv_idx + so * SLOTV];
            if constexpr (C == 1) {
              hv[0] = src[0];
            } else if constexpr (C == 3) {
              hv[0] = src[0];
              hv[1] = src[1];
              hv[2] = src[2];
            } else {
#pragma unroll
              for (int i = 0; i < C; i += 2) {
                const hb_double2 t2 = *reinterpret_cast<const hb_double2 *>(src + i);
                hv[i] = t2.x;
                hv[i + 1] = t2.y;
              }
            }
            he = rge[re_idx + so * MHL];
          }
          HB_FINE(2);
          // ---- the record of the block ----
          hb_store_record<C>(rec_mask, rvp, v);
          rvp += U * C;
          __hip_atomic_store(rep, rec_lane ? (unsigned)ep : (unsigned)vcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          rep += rep_step;
          if (HB_UNLIKELY(jw == 0)) {
            if (lane == 0) __hip_atomic_store(prog, (unsigned)(b + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          HB_FINE(3);
          // ---- is the halo there, and has the right neighbour room for the next hand-over?  (Not in a strip's first block,
          // whose lane 60 looks at a word that says no.) ----
#ifdef HB_TL_FINE
          if (fdbg) fdbg[(size_t)b * 8 + 7] = 101;  // (there at the first look)
#endif
          if (HB_UNLIKELY(__any(seen < vcnt))) {
#ifdef HB_TL_FINE
            if (fdbg) fdbg[(size_t)b * 8 + 7] = 99;
#endif
            if (b == b0) {
              // Everything is set up -- and the first block's hand-over and record are out: they are the wave as it stands,
              // nothing of the halo -- before the wave dozes until the fetcher has the first halo: a strip can never make up
              // for a late start -- its neighbours walk at the same pace -- so what a workgroup loses at its start is added to
              // the end of the fill, once per workgroup of the table (tools/hop_hb.py).
              while (!lds_peek(&s_awake) && !lds_peek(&s_abort)) __builtin_amdgcn_s_sleep(2);
              if (w > 0 && X.doze)
                while (lds_peek(left_cnt) < b0 + 1 && !lds_peek(&s_abort)) __builtin_amdgcn_s_sleep(8);
              __builtin_amdgcn_s_setprio(3);
              if (dbg) dbg[0] = HB_STAMP();
              if (lane == 60) chk_p = &c_huge;
            }
            // (one LDS round trip per look: the counters and, behind them, the halo they guard -- a strip's first hand-over
            // decides how far behind its left neighbour it walks for the rest of the fill)
            bool there = false;
            const double *src = &rgv[rv_idx + so * SLOTV];
            for (int k = 0; k < HB_SPIN && !aborted; k++) {
              const int seen2 = __hip_atomic_load(chk_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              asm volatile("" ::: "memory");
#pragma unroll
              for (int i = 0; i < C; i++) hv[i] = src[i];
              he = rge[re_idx + so * MHL];
              if (!__any(seen2 < vcnt)) {
                there = true;
                break;
              }
            }
            if (!there) {
              if (jw > 0) wait_ge(left_cnt, b + 1, 0x100u);
              if (has_next) wait_ge(&c_taken[w + 1], b - 6, 0x400u);
#pragma unroll
              for (int i = 0; i < C; i++) hv[i] = src[i];
              he = rge[re_idx + so * MHL];
            }
          }
#pragma unroll
          for (int i = 0; i < C; i++) v[i] = halo ? hv[i] : v[i];
          ep = halo ? he : ep;
          if (HB_UNLIKELY(tl_on)) {
            if (dbg) dbg[1 + b] = HB_STAMP();
          }
          HB_FINE(4);
          // ---- R rows alone ----
          {
            const int dl = wave_shr1(ep, ep) - ep;
            s = ldexp(1.0, min(max(dl, -1100), 220));
          }
          HB_FINE(5);
          // (the next block's renormalisation shift is taken from the lane four rows before the block's end -- the rows hide
          // its dependent steps; what the significands grow by in four rows, at most 2^(4 (2 log2 N + 1)), is inside the
          // slack stb_period_rows leaves -- and applied after the last row: two ldexp and an add on the block top's path)
          int sh;
          if constexpr (R48) {
#pragma unroll
            for (int u = 0; u < 44; u++) hb_row<C>(v, coef, s);
            sh = hb_renorm_shift<C>(v);
#pragma unroll
            for (int u = 44; u < 48; u++) hb_row<C>(v, coef, s);
          } else {
            for (int r = 0; r < R; r += 8) {
#pragma unroll
              for (int u = 0; u < 8; u++) hb_row<C>(v, coef, s);
            }
            sh = hb_renorm_shift<C>(v);
          }
          HB_FINE(6);
          if (HB_UNLIKELY(++b >= NB)) break;
          HB_FINE(0);
          hb_renorm_apply<C>(v, ep, sh);
        }
        };
        // (the 48 rows of a block in line: a taken branch costs a lone wave most of a row)
        if (HB_LIKELY(R == 48))
          spine_loop(std::true_type{});
        else
          spine_loop(std::false_type{});
#else
        // what a block begins with and needs no halo for: the hand-over to the right neighbour in the workgroup and the
        // block's record, both of the wave as it stands
        auto block_top = [&](const int b) {
          // ---- the rightmost HL lanes, for the right neighbour in this workgroup ----
            if (has_next && (HB_ABL & 2)) lds_post(&c_posted[w + 1], b + 1);
            if (HB_LIKELY(has_next) && !(HB_ABL & 2)) {
              // (the ring holds 8 blocks: every 4th block it is made sure that the right neighbour has taken all
              // but the last 4, which covers this block and the next three)
              if (HB_UNLIKELY((b & 3) == 0 || b == b0)) wait_ge(&c_taken[w + 1], b - 4, 0x400u);
              if (lane >= U) {
                double *dst = &rgv[(w + 1) * RINGV + (b & (HB_SLOTS - 1)) * SLOTV + (lane - U) * C];
#pragma unroll
                for (int i = 0; i < C; i++) dst[i] = v[i];
                rge[(w + 1) * RINGE + (b & (HB_SLOTS - 1)) * MHL + lane - U] = ep;
              }
              lds_post(&c_posted[w + 1], b + 1);
            }
            HB_FINE(2);
            // ---- the record of the block: the own lanes as they stand before it (the halo lanes of strip 0 are
            // exact: they go to strip index 0 at the place a left neighbour's rightmost lanes would have).
            // (A publisher wave that takes the row from LDS and stores it in the spine's stead was tried: the spine
            // got slower, 36 against 32.5 ns a row alone and 53 against 48 beside eight tables' workers.) ----
            if (HB_LIKELY(own || jw == 0) && !(HB_DIAG & 1) && !(HB_ABL & 8)) {
              unsigned long long *dst = rec_v + (size_t)b * (size_t)(U * C);
              if constexpr (C == 1) {
                __hip_atomic_store(dst, (unsigned long long)__double_as_longlong(v[0]) | HB_WRITTEN, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
              } else if constexpr (C == 3) {
                hb_store_wt16(dst, (unsigned long long)__double_as_longlong(v[0]) | HB_WRITTEN, (unsigned long long)__double_as_longlong(v[1]) | HB_WRITTEN);
                __hip_atomic_store(dst + 2, (unsigned long long)__double_as_longlong(v[2]) | HB_WRITTEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              } else {
#pragma unroll
                for (int i = 0; i < C; i += 2)
                  hb_store_wt16(dst + i, (unsigned long long)__double_as_longlong(v[i]) | HB_WRITTEN,
                                (unsigned long long)__double_as_longlong(v[i + 1]) | HB_WRITTEN);
              }
              __hip_atomic_store(rec_e + (size_t)b * (size_t)U, (unsigned)ep + HB_EOFF32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0 && !(HB_ABL & 16)) __hip_atomic_store(prog, (unsigned)(b + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // Everything is set up (the loads above included) -- and the first block's top done: its record and hand-over are
        // the wave as it stands, nothing of the halo -- before the wave dozes until the fetcher has the first
        // halo: a strip can never make up for a late start -- its neighbours walk at the same pace -- so what a
        // workgroup loses at its start is added to the end of the fill, once per workgroup of the table
        // (tools/hop_hb.py: a hop is 0.8 us of hand-over and as much again of never-recovered start).
        block_top(b0);
        while (!lds_peek(&s_awake) && !lds_peek(&s_abort)) __builtin_amdgcn_s_sleep(2);
        // ... and until its left neighbour in the workgroup is about to hand over the first halo: a wave that spins on
        // the counter inside the block loop (every 64 cycles) takes issue slots from the spine wave it shares a SIMD
        // with (waves w and w + 4 of a workgroup) for as many blocks as the diagonal is away
        if (w > 0 && X.doze)
          while (lds_peek(left_cnt) < b0 + 1 && !lds_peek(&s_abort)) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_s_setprio(3);
        if (dbg) dbg[0] = HB_STAMP();
        for (int b = b0; b < NB; b++) {
          HB_FINE(0);
          if (HB_LIKELY(b > b0)) {
            if (!(HB_ABL & 4)) hb_renorm<C>(v, ep);
            HB_FINE(1);
            block_top(b);
          }
          HB_FINE(3);
          // ---- the halo: the left neighbour's rightmost HL lanes as they stand before the block ----
          if (jw > 0 && (HB_ABL & 1)) lds_post(&c_taken[w], b + 1);
          if (HB_LIKELY(jw > 0) && !(HB_ABL & 1)) {
            // (the counter and the data are asked for together -- LDS serves a wave's requests in order, so data
            // read after a counter that says "there" is there -- and only if the counter says "not yet" is it
            // waited for and the data read again: one LDS round trip instead of two.  Asking at the top of the block,
            // so that the round trip passes behind the hand-over and the record above, gains nothing: measured 0.316 ms
            // either way for one table -- the round trip is not what a block waits for)
            double hv[C];
            int he = 0;
            const double *src = &rgv[w * RINGV + (b & (HB_SLOTS - 1)) * SLOTV + lane * C];
            const int *srce = &rge[w * RINGE + (b & (HB_SLOTS - 1)) * MHL + lane];
            const int seen = aborted ? 0x7fffffff : lds_peek(left_cnt);
#ifdef HB_TL_FINE
            if (fdbg) fdbg[(size_t)b * 8 + 7] = (unsigned long long)(unsigned)(seen - b + 100);
#endif
            asm volatile("" ::: "memory");
            if (lane < HL) {
#pragma unroll
              for (int i = 0; i < C; i++) hv[i] = src[i];
              he = *srce;
            }
            if (HB_UNLIKELY(seen < b + 1)) {
              wait_ge(left_cnt, b + 1, 0x100u);
              if (lane < HL) {
#pragma unroll
                for (int i = 0; i < C; i++) hv[i] = src[i];
                he = *srce;
              }
            }
            if (lane < HL) {
#pragma unroll
              for (int i = 0; i < C; i++) v[i] = hv[i];
              ep = he;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the slot is in registers: it may be written again)
            lds_post(&c_taken[w], b + 1);
          }
          if (dbg) dbg[1 + b] = HB_STAMP();
          HB_FINE(4);
          // ---- R rows alone ----
          {
            const int dl = wave_shr1(ep, ep) - ep;
            s = ldexp(1.0, min(max(dl, -1100), 220));
          }
          HB_FINE(5);
          // (a taken branch costs a lone wave ~28 cycles, most of a row: tools/ubench/rowvar.hip, 35.5 cycles a row in a
          // loop of 8 rows, 31.0 with the 48 rows of a block in line)
          if (HB_LIKELY(R == 48)) {
#pragma unroll
            for (int u = 0; u < 48; u++) hb_row<C>(v, coef, s);
          } else {
            for (int r = 0; r < R; r += 8) {
#pragma unroll
              for (int u = 0; u < 8; u++) hb_row<C>(v, coef, s);
            }
          }
          HB_FINE(6);
        }
#endif
        if (lane == 0) __hip_atomic_store(prog, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (dbg) dbg[NB + 1] = HB_STAMP();
        if (lane == 0) {
          atomicAdd(X.hdr + 4, 1u);  // (spine waves that are through: diagnostics only)
          atomicMax(reinterpret_cast<unsigned long long *>(X.hdr + STB_HDR_T1), (unsigned long long)wall_clock64());  // (... and when)
        }
        __builtin_amdgcn_s_setprio(0);
      }
    } else if (wave == P && j > 0) {
      // ================= fetcher: the left workgroup's last strip's records -> LDS, for spine wave 0 =================
      // groups of 16 (or 32) lanes poll consecutive blocks: a poll takes longer than a block
      const int gl = (HL <= 16) ? 16 : 32;
      const int grp = lane / gl, sub = lane % gl, ngrp = 64 / gl;
      const bool act = sub < HL;
      const int bL0 = hb_first_block(jw0 - 1, UC, R);
      const size_t rec_left = tab_rec + rec_off(jw0) - (size_t)bL0;  // (strip jw0 - 1 has strip index jw0)
      const int slot = sub + U - HL;
      int bb = hb_first_block(jw0, UC, R);  // blocks below it are delivered
      // (the left strip writes records from its own first block on; the first one that arrives wakes the spine
      // waves, and is delivered with the same look that found it)
      bool woke = false;
      unsigned long long t_begin = 0;
      bool timing = false;
      unsigned idle = 0;
#ifdef HB_TL_FINE
      // (diagnostic build: when the look that found block b complete was sent and when it was back, per block of this
      // workgroup's first strip -- behind the spine's fine stamps: [JW][NB][2])
      unsigned long long *hopdbg = (X.dbg && d == 0) ? X.dbg + (size_t)X.JW * (NB + 2) + (size_t)X.n_order * 4 + (size_t)X.JW * NB * 8 + (size_t)jw0 * NB * 2 : nullptr;
#endif
      while (bb < NB) {
        const int mb = bb + grp;
        const bool want = act && mb < NB;
        unsigned long long bv[C];
        unsigned be = 1;
#ifdef HB_TL_FINE
        const unsigned long long t_sent = HB_STAMP();
#endif
#pragma unroll
        for (int i = 0; i < C; i++) bv[i] = 1;
        if (want) {
          const unsigned long long *src = X.ck_v + ((rec_left + mb) * U + slot) * C;
#pragma unroll
          for (int i = 0; i < C; i++) bv[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          be = __hip_atomic_load(X.ck_e + (rec_left + mb) * U + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int tk = lds_peek(&c_taken[0]);
        bool have = be != 0;
#pragma unroll
        for (int i = 0; i < C; i++) have = have && bv[i] != 0;
        const unsigned long long miss = ~__ballot(have);
        // leading groups that are complete, in range, and whose ring slot spine wave 0 has read
        int k = miss ? (int)(__builtin_ctzll(miss) / gl) : ngrp;
        k = min(k, NB - bb);
        k = min(k, tk + HB_FSLOTS - bb);
        if (k > 0) {
#ifdef HB_TL_FINE
          if (hopdbg && sub == 0 && grp < k && mb < NB) {
            hopdbg[(size_t)mb * 2] = t_sent;
            hopdbg[(size_t)mb * 2 + 1] = HB_STAMP();
          }
#endif
          if (want && grp < k) {
            double *dst = &rgv[(mb & (HB_FSLOTS - 1)) * SLOTV + sub * C];
#pragma unroll
            for (int i = 0; i < C; i++) dst[i] = __longlong_as_double((long long)(bv[i] & ~HB_WRITTEN));
            rge[(mb & (HB_FSLOTS - 1)) * MHL + sub] = (int)(be - HB_EOFF32 + HB_EB);
          }
          bb += k;
          lds_post(&c_posted[0], bb);
          if (!woke) {
            woke = true;
            lds_post(&s_awake, 1);
          }
          timing = false;
          idle = 0;
          continue;
        }
        // (until the first record has come the looks follow each other as fast as they return: what is lost before a
        // strip's first block is lost for the whole fill -- 0.272 against 0.274 ms with a pause of 0.25 us between them)
        if (woke)
          for (int i = 0; i < X.poll_nap; i++) __builtin_amdgcn_s_sleep(1);
        if ((++idle & 31) != 0 && X.timeout != 0) continue;
        if (!timing) {
          timing = true;
          t_begin = wall_clock64();
        }
        const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (err != 0 || lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin >= X.timeout) {
          if (lane == 0) {
            __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (err == 0) {
              __hip_atomic_store(X.hdr + 2, who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(X.hdr + 1, 0x900u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          lds_post(&c_posted[0], 0x7fffffff);  // release the spine: it runs on with stale halos
          lds_post(&s_awake, 1);
          break;
        }
      }
    }
    // Whoever is through with its part of the spine works on tiles -- but only once every workgroup of the
    // grid has started (each takes a ticket when it does): while spine workgroups are still waiting for a
    // free compute unit, this one must give its place up, or its waves would sit on tiles of strips whose
    // spine cannot start.  Waves without a part (no strip, no fetching) sleep until the spine is through:
    // a worker on the spine's compute unit takes issue slots from it.
    // (X.spare_work: how many of them at most -- the others sleep until the spine is through, like all of them in a storing fill)
    if (wave - P <= X.spare_work && X.spare_work && (wave > P || (wave == P && j == 0) || (wave < P && jw0 + wave >= X.JW))) {
      // (tunable: the spare waves work on tiles from the start, once every workgroup of the grid is running)
      unsigned spins = 0;
      while (__hip_atomic_load(X.hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < 200000u)
        __builtin_amdgcn_s_sleep(64);
    } else if (wave > P || (wave == P && j == 0) || (wave < P && jw0 + wave >= X.JW)) {
      const int last = min(P, X.JW - jw0) - 1;
      unsigned spins = 0;
      while (__hip_atomic_load(X.progress + ((size_t)d * X.JW + jw0 + last) * HB_PROG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0x7fffffffu &&
             !lds_peek(&s_abort)) {
        __builtin_amdgcn_s_sleep(127);
        if ((++spins & 1023u) == 0 && __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      }
    }
    if (__hip_atomic_load(X.hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) return;
  }

  // =========================================================================================
  // tile workers (every wave for itself)
  {
    int one_hi = 0x3ff00000;
    asm volatile("" : "+v"(one_hi));
    if (X.diag & 1) {
      // diagnostic: the workers start when every spine wave is through (what the tiles cost with the chip to themselves)
      unsigned spins = 0;
      while (__hip_atomic_load(X.hdr + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(X.JW * X.D) && ++spins < 400000u)
        __builtin_amdgcn_s_sleep(100);
    }
    // Tickets.  ONE counter hands out 88 million tickets a second whoever asks (tools/ubench/ticket.hip): eight
    // tables of 10^4 columns are 41 000 tiles, half a millisecond of tickets alone.  So there is a counter per
    // table -- and per interleaved part of the order list when the tables are few -- each on a line of its own;
    // a wave starts at the counter its number gives it and moves on when one is exhausted (and tells its
    // workgroup, so that the other waves do not ask there again).
    // (more than HB_MAXCNT tables share counters: table d asks counter d mod Dg)
    const unsigned NC = (unsigned)X.n_cnt, Dg = min((unsigned)X.D, (unsigned)HB_MAXCNT), S = NC / Dg;
    unsigned cur = ((unsigned)blockIdx.x * NW + (unsigned)wave) % NC, misses = 0;
    for (;;) {
      if (lds_peek(&s_done[cur])) {
        if (++misses >= NC) break;
        cur = (cur + 1 == NC) ? 0 : cur + 1;
        continue;
      }
      const unsigned grp = cur % Dg, part = cur / Dg;          // tiles part, part + S, part + 2 S, .. of the order list
      const unsigned Tg = ((unsigned)X.D - grp + Dg - 1) / Dg;  // tables grp, grp + Dg, ..
      const unsigned mine = (X.n_order + S - 1 - part) / S * Tg;  // how many tickets this counter hands out
      unsigned k = 0;
      if (lane == 0) k = atomicAdd(X.hdr + HB_CNT0 + cur * HB_PROG_STRIDE, 1u);
      k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
      if (k >= mine) {
        if (lane == 0) __hip_atomic_store(&s_done[cur], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (++misses >= NC) break;
        cur = (cur + 1 == NC) ? 0 : cur + 1;
        continue;
      }
      misses = 0;
      const int d = (int)(grp + Dg * (k % Tg));
      const unsigned oi = (k / Tg) * S + part;
      const unsigned ord = (unsigned)__builtin_amdgcn_readfirstlane((int)(order_in_lds ? s_order[oi] : X.order[oi]));
      const int jw = (int)(ord & (X.split ? 0x3fffu : 0xffffu)), b = (int)(ord >> 16);
      // The tiles of a strip's last blocks come as four tickets, a quarter of the rows each (the rows before it are walked
      // without logs and stores, an eighth of a tile's cost): what the fill waits for at its very end is the last tiles'
      // latency -- 8-10 us a whole tile, all other workers idle -- not throughput.
      const bool quartered = X.split != 0 && b >= NB - X.split;
      const int r_lo = quartered ? (int)((ord >> 14) & 3u) * (R >> 2) : 0, r_hi = quartered ? r_lo + (R >> 2) : R;
      const unsigned who = (unsigned)jw | ((unsigned)d << 16);
      unsigned long long *wdbg = (X.dbg && d == 0 && lane == 0) ? X.dbg + (size_t)X.JW * (NB + 2) + (size_t)oi * 4 : nullptr;
      if (wdbg) wdbg[0] = wall_clock64();
      // ---- the tile's inputs: the strip's record of the block (own lanes) and the left strip's (halo lanes).
      // Everything is asked for at once, without looking at the spine's progress first: what has been written
      // is non-zero.  Only when something is missing is the progress word read, to sleep about as long as the
      // missing blocks take. ----
      // A worker lane holds NG groups of CG adjacent elements.  With 4 elements per spine lane the two groups
      // lie a kilobyte apart -- elements 2 l, 2 l + 1 and 128 + 2 l, 129 + 2 l of the wave's 256 -- so that each
      // of a row's two store instructions covers ONE contiguous kilobyte (8 whole lines) instead of 64 pieces
      // of 16 bytes 32 apart: half the requests to the memory side, which is what a compute unit's store rate
      // is made of (tools/ubench/wcap.hip: 129 against 65 GB/s per compute unit).  A group takes its
      // exponent from the spine lane its elements belong to.
      const size_t tab_rec = (size_t)d * X.n_rec;
      const int bO = hb_first_block(jw, UC, R), bL = (jw > 0) ? hb_first_block(jw - 1, UC, R) : 0;
      const size_t recO = tab_rec + rec_off(jw + 1) + (size_t)(b - bO), recL = tab_rec + rec_off(jw) + (size_t)(b - bL);
      const unsigned long long *ckv[NG];
      const unsigned *cke[NG];
      bool own[NG];
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const int L = (NG == 2) ? g * 32 + (lane >> 1) : lane;  // the spine lane of the group
        const int i0 = (NG == 2) ? 2 * (lane & 1) : 0;          // ... and the group's first element in it
        own[g] = L >= HL;
        const size_t rec = own[g] ? recO : recL;
        const int slot = own[g] ? L - HL : L + U - HL;
        ckv[g] = X.ck_v + (rec * U + slot) * C + i0;
        cke[g] = X.ck_e + rec * U + slot;
      }
      const unsigned *prog = X.progress + ((size_t)d * X.JW + jw) * HB_PROG_STRIDE;
      // summing form: first entry of each of the tile's groups of four rows (lane q: group q; lane NQ: the end)
      unsigned ip = 0;
      int qe = 0;  // groups below it have occurring cells
      unsigned short pp[DOT ? HB_DOT_NQ : 1];
      unsigned cc[DOT ? HB_DOT_NQ : 1];
      if constexpr (DOT != 0) {
        const int NQ = R / HB_DOT_GR;
        const unsigned tix = (unsigned)(recO - tab_rec);
        if (lane <= NQ) ip = X.item_ptr[(size_t)tix * HB_DOT_NQ + lane];
        const unsigned ipn = (unsigned)__shfl_down((int)ip, 1);
        const unsigned long long hm = __ballot(lane < NQ && ipn != ip);
        qe = hm ? 64 - (int)__builtin_clzll(hm) : 0;
        if (qe == 0) {  // none of the tile's cells occurs: nothing to walk, nothing to wait for
          if (lane == 0) X.dotp[(size_t)d * X.n_tiles + (tix - (unsigned)NB)] = 0.0;
          continue;
        }
      }
      double v[NG][CG], coef[NG][CG];
      int ep[NG];
      bool ok = true;
      {
        unsigned spins = 0, tries = 0;
        unsigned long long t_begin = 0;
        for (;;) {
          unsigned long long bv[NG][CG];
          unsigned be[NG];
#pragma unroll
          for (int g = 0; g < NG; g++) {
#pragma unroll
            for (int i = 0; i < CG; i++) bv[g][i] = __hip_atomic_load(ckv[g] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            be[g] = __hip_atomic_load(cke[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          bool have = true;
#pragma unroll
          for (int g = 0; g < NG; g++) {
            have = have && be[g] != 0;
#pragma unroll
            for (int i = 0; i < CG; i++) have = have && bv[g][i] != 0;
          }
          if (__all(have)) {
#pragma unroll
            for (int g = 0; g < NG; g++) {
#pragma unroll
              for (int i = 0; i < CG; i++) v[g][i] = __longlong_as_double((long long)(bv[g][i] & ~HB_WRITTEN));
              ep[g] = (int)(be[g] - HB_EOFF32);
            }
            break;
          }
          // Not there yet.  Thousands of waves may be waiting like this while a table's first rows are walked,
          // and every look at a record is 64 lanes' worth of requests to the memory side: wait on the strip's
          // progress word alone -- one request per look, a line of its own -- sleeping about as long as the
          // blocks still missing take, then read the records again.
          for (;;) {
            const unsigned done = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done >= (unsigned)(b + 1)) break;
            // (a strip that has not started yet -- done = 0 -- will start at its first block bO, not at block 0: counting
            // from 0, a worker with one of a right-hand strip's first tiles slept through 32 blocks' time and then, finding
            // the strip still not started, through 32 more -- the fill's last tiles were noticed 13-20 us after their
            // records were written, profiles/r05_tail_before.txt)
            const int missing = min((int)((unsigned)(b + 1) - (done ? done : (unsigned)bO)) + (done ? 0 : 2), 32);
            for (int i = 0; i < missing; i++)
              for (int q = 0; q < X.nap_block; q++) __builtin_amdgcn_s_sleep(8);
            if ((++spins & 3u) != 0 && X.timeout != 0) continue;
            if (t_begin == 0) t_begin = wall_clock64();
            const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (err != 0 || (unsigned long long)wall_clock64() - t_begin >= X.timeout) {
              if (err == 0 && lane == 0) {
                __hip_atomic_store(X.hdr + 2, who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(X.hdr + 1, 0xA00u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              ok = false;
              break;
            }
          }
          if (!ok) break;
          // (the progress word may overtake the record's words by a little, and the left strip's record is
          // not covered by it: look again shortly; the same clock bounds this)
          __builtin_amdgcn_s_sleep(8);
          if ((++tries & 63u) != 0 && X.timeout != 0) continue;
          if (t_begin == 0) t_begin = wall_clock64();
          const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (err != 0 || (unsigned long long)wall_clock64() - t_begin >= X.timeout) {
            if (err == 0 && lane == 0) {
              __hip_atomic_store(X.hdr + 2, who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(X.hdr + 1, 0xA00u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ok = false;
            break;
          }
        }
      }
      if (!ok) break;
      if (wdbg) wdbg[1] = wall_clock64();
      const double a = A.a[d];
      const int mE0 = 2 + (jw * U - HL) * C;  // the column of the wave's first element (halo included)
      const double n1 = (double)(1 + b * R);
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const int m0 = mE0 + ((NG == 2) ? g * 128 + 2 * lane : lane * C);
#pragma unroll
        for (int i = 0; i < CG; i++) coef[g][i] = n1 - (double)(m0 + i) * a;
      }
      // what a group's first element takes from the element to its left, which lives in lane l - 1 -- or,
      // for the second group of lane 0, in the first group of lane 63 -- under that lane's exponent
      double s[NG], z1 = 0.0;
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const int dl = wave_shr1(ep[g], ep[g]) - ep[g];
        s[g] = (lane == 0) ? 0.0 : ldexp(1.0, min(max(dl, -1100), 220));
      }
      if constexpr (NG == 2) {
        const int e63 = __builtin_amdgcn_readlane(ep[0], 63);
        if (lane == 0) z1 = ldexp(1.0, min(max(e63 - ep[1], -1100), 220));
      }
      if constexpr (DOT != 0) {
        // ---- the tile as a sum: no logs but those of the cells that occur, nothing stored.  Of every group of eight
        // rows every other one goes to LDS (this wave's own area; eight waves staging every row keep a compute unit's
        // LDS store path busy twice as long as its SIMDs take to walk), then every lane looks one occurring cell up
        // -- a cell of a row in between is one step of the recurrence away from the staged row above it; the
        // first 64 cells of every group's list were asked for when the tile was taken. ----
        constexpr int WS = 64 * C;
        double *stage = hb_dyn + (size_t)wave * (4 * WS);
        int *se = &w_se[DOT ? wave : 0][0];
        // the first 64 cells of every group's list are asked for now, all at once
#pragma unroll
        for (int q = 0; q < HB_DOT_NQ; q++) {
          pp[q] = 0;
          cc[q] = 0;
          if (q < qe) {
            const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)ip, q), b1 = (unsigned)__builtin_amdgcn_readlane((int)ip, q + 1);
            if (b0 + lane < b1) {
              pp[q] = X.ent_pos[b0 + lane];
              cc[q] = X.ent_cnt[b0 + lane];
            }
          }
        }
        se[lane] = ep[0];
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < HB_DOT_NQ; q++) {
          if (q < qe) {
            const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)ip, q), b1 = (unsigned)__builtin_amdgcn_readlane((int)ip, q + 1);
#pragma unroll
            for (int u = 0; u < HB_DOT_GR; u++) {
              const double t0 = wave_shr1_zero(v[0][C - 1]) * s[0];
#pragma unroll
              for (int i = C - 1; i >= 1; i--) v[0][i] = fma(coef[0][i], v[0][i], v[0][i - 1]);
              v[0][0] = fma(coef[0][0], v[0][0], t0);
#pragma unroll
              for (int i = 0; i < C; i++) coef[0][i] += 1.0;
              if ((u & 1) == 0 && b0 != b1 && !(HB_DIAG & 32)) {  // (a group none of whose cells occurs is only walked)
                if constexpr (C == 4) {
                  *reinterpret_cast<hb_double2 *>(stage + (u >> 1) * WS + lane * 4) = hb_double2{v[0][0], v[0][1]};
                  *reinterpret_cast<hb_double2 *>(stage + (u >> 1) * WS + lane * 4 + 2) = hb_double2{v[0][2], v[0][3]};
                } else if constexpr (C == 2) {
                  *reinterpret_cast<hb_double2 *>(stage + (u >> 1) * WS + lane * 2) = hb_double2{v[0][0], v[0][1]};
                } else if constexpr (C == 3) {
                  double *st3 = stage + (u >> 1) * WS + lane * 3;
                  st3[0] = v[0][0];
                  st3[1] = v[0][1];
                  st3[2] = v[0][2];
                } else {
                  stage[(u >> 1) * WS + lane] = v[0][0];
                }
              }
            }
            if (b0 != b1 && !(HB_DIAG & 16)) {
              unsigned kk = b0 + lane, pos = pp[q], cnt = cc[q];
              for (;;) {
                if (kk < b1) {
                  const int cw = (int)(pos & 255u), r = (int)(pos >> 8);
                  const double *row = stage + (r >> 1) * WS;
                  const int ln = cw / C;
                  const int e = se[ln];
                  double x = row[cw];
                  // (rows 1 and 3 of the group: n - 1 - m a times the cell above, plus the cell above and to the left
                  // -- under the left lane's exponent where that is another lane: the walk's own operations)
                  const int dl = (ln > 0 ? se[ln - 1] : e) - e;
                  const double xl = row[cw - 1 < 0 ? 0 : cw - 1] * (((cw % C) == 0) ? ldexp(1.0, min(max(dl, -1100), 220)) : 1.0);
                  const double c1 = fma(-(double)(mE0 + cw), a, (double)(1 + b * R + q * HB_DOT_GR + r));
                  if (r & 1) x = fma(c1, x, xl);
                  const double val = bfp_log(x, e, lt);
                  acc += (double)cnt * val;
                }
                if (kk - lane + 64 >= b1) break;  // (wave-uniform)
                kk += 64;
                pos = cnt = 0;
                if (kk < b1) {
                  pos = X.ent_pos[kk];
                  cnt = X.ent_cnt[kk];
                }
              }
            }
          }
        }
        // fixed-shape tree over the wave: the same bits on every run, whoever computed the tile
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) X.dotp[(size_t)d * X.n_tiles + ((unsigned)(recO - tab_rec) - (unsigned)NB)] = acc;
      } else {
      double *table = A.tables + (uint64_t)d * A.tstride;
      float *tablef = reinterpret_cast<float *>(A.tables) + (uint64_t)d * A.tstride;
      const unsigned e0 = (unsigned)(jw * UC);  // first own element of the strip
      // (the lane offset counts from the wave's first halo element: the base may lie before the row)
      const unsigned voff = (unsigned)(lane * CG) * (FL ? 4u : 8u);
      unsigned n = 2u + (unsigned)(b * R);  // the row the next step produces
      // the S table's rows start at n = 3 and hold m = 2 .. min(n - 1, M); the V table's at n = 2, m = 2 .. min(n, M)
      constexpr unsigned NMIN = VT ? 2u : 3u;
      auto rlen = [&](unsigned nn) { return VT ? stb_vrow_len(nn, M) : stb_row_len(nn, M); };
      auto rpitch = [&](unsigned nn) { return VT ? stb_vrow_pitch(nn, M) : stb_row_pitch(nn, M); };
      uint64_t roff = VT ? stb_vrow_offset(n, M) : stb_row_offset(n, M);
      constexpr int RS = 8 / C;  // rows converted together: eight cells in flight
      int ep8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) ep8[u] = ep[(NG == 2) ? ((u >> 1) & 1) : 0];
      // (ratios: what a group's first element takes from its left neighbour in the NEXT row is also the denominator of
      // its ratio in THIS row: carried from row to row)
      double tc[NG];
      if constexpr (VT) {
        if constexpr (NG == 1) {
          tc[0] = wave_shr1_zero(v[0][CG - 1]) * s[0];
        } else {
          const double ra = wave_ror1(v[0][1]), rb = wave_shr1_zero(v[1][1]);
          tc[0] = ra * s[0];
          tc[NG - 1] = fma(ra, z1, rb * s[1]);
        }
      }
      for (int r = 0; r < r_hi; r += RS) {
        double x[8], val[8], xl[VT ? 8 : 1];
#pragma unroll
        for (int u = 0; u < RS; u++) {
          if constexpr (VT) {
            if constexpr (NG == 1) {
#pragma unroll
              for (int i = CG - 1; i >= 1; i--) v[0][i] = fma(coef[0][i], v[0][i], v[0][i - 1]);
              v[0][0] = fma(coef[0][0], v[0][0], tc[0]);
              tc[0] = wave_shr1_zero(v[0][CG - 1]) * s[0];
            } else {
              v[0][1] = fma(coef[0][1], v[0][1], v[0][0]);
              v[0][0] = fma(coef[0][0], v[0][0], tc[0]);
              v[1][1] = fma(coef[1][1], v[1][1], v[1][0]);
              v[1][0] = fma(coef[1][0], v[1][0], tc[NG - 1]);
              const double ra = wave_ror1(v[0][1]), rb = wave_shr1_zero(v[1][1]);
              tc[0] = ra * s[0];
              tc[NG - 1] = fma(ra, z1, rb * s[1]);
            }
#pragma unroll
            for (int g = 0; g < NG; g++)
#pragma unroll
              for (int i = 0; i < CG; i++) xl[u * C + g * CG + i] = (i == 0) ? tc[g] : v[g][i - 1];
          } else if constexpr (NG == 1) {
            const double t0 = wave_shr1_zero(v[0][CG - 1]) * s[0];
#pragma unroll
            for (int i = CG - 1; i >= 1; i--) v[0][i] = fma(coef[0][i], v[0][i], v[0][i - 1]);
            v[0][0] = fma(coef[0][0], v[0][0], t0);
          } else {
            const double ra = wave_ror1(v[0][1]), rb = wave_shr1_zero(v[1][1]);
            const double ta = ra * s[0];
            const double tb = fma(ra, z1, rb * s[1]);
            v[0][1] = fma(coef[0][1], v[0][1], v[0][0]);
            v[0][0] = fma(coef[0][0], v[0][0], ta);
            v[1][1] = fma(coef[1][1], v[1][1], v[1][0]);
            v[1][0] = fma(coef[1][0], v[1][0], tb);
          }
#pragma unroll
          for (int g = 0; g < NG; g++)
#pragma unroll
            for (int i = 0; i < CG; i++) {
              coef[g][i] += 1.0;
              x[u * C + g * CG + i] = v[g][i];
            }
        }
        // (rows none of whose cells lies in the strip's own columns, and rows outside the table, are only walked)
        const unsigned nl = n + RS - 1;
        if (r >= r_lo && nl >= NMIN && n <= N && e0 < rlen(min(nl, N))) {
          if constexpr (VT) {
            // V^n_m = S^n_m / S^n_{m-1}: both under the lane's exponent (cells right of the diagonal: 0 / 0 or x / 0 --
            // they land in the row's slack, which nobody reads)
#pragma unroll
            for (int q = 0; q < 8; q++) val[q] = x[q] / xl[q];
          } else {
            hb_logs8<LTX>(x, ep8, lt, one_hi, val);
          }
#pragma unroll
          for (int u = 0; u < RS; u++) {
            const unsigned nu = n + u;
            if (nu >= NMIN && nu <= N && e0 < rlen(nu)) {
              const double *rp = table + roff + e0 - (size_t)(HL * C);
              if constexpr (FL) {
                // (the arithmetic was double; the stored value is a float: two or four of them per lane, one store)
                const float *rpf = tablef + roff + e0 - (size_t)(HL * C);
                const double p0 = __hiloint2double(__float_as_int((float)val[u * C + 1]), __float_as_int((float)val[u * C]));
                if constexpr (C == 2) {
                  if (own[0]) store_sbase(rpf, voff, p0);
                } else {
                  const double p1 = __hiloint2double(__float_as_int((float)val[u * C + 3]), __float_as_int((float)val[u * C + 2]));
                  if (own[0]) hb_store16(rpf, voff, p0, p1);
                }
              } else if constexpr ((HB_DIAG & 8) != 0) {
                asm volatile("" ::"v"(val[u * C]), "v"(val[u * C + C - 1]));
              } else if constexpr (C == 1) {
                if (own[0]) store_sbase(rp, voff, val[u]);
              } else if constexpr (C == 2) {
                if (own[0]) hb_store16(rp, voff, val[u * 2], val[u * 2 + 1]);
              } else {
                if (own[0]) hb_store16(rp, voff, val[u * 4], val[u * 4 + 1]);
                if (own[1]) hb_store16(rp + 128, voff, val[u * 4 + 2], val[u * 4 + 3]);
              }
            }
            roff += rpitch(nu);
          }
        } else {
#pragma unroll
          for (int u = 0; u < RS; u++) roff += rpitch(n + u);
        }
        n += RS;
      }
      }
      if (wdbg) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        wdbg[2] = wall_clock64();
        wdbg[3] = (unsigned long long)hw | ((unsigned long long)(xcc & 15u) << 32);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side

struct hb_geom {
  int C, P, B, JW, NB, R, HL, U;
  unsigned n_tiles, n_rec;
  size_t off_prog, off_cke, off_ckv, zero_bytes, bytes;
  bool ok;
};

int stb_cu_count() {  // compute units of the current device (256 on an MI355X)
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
    cus = 256;
  return cus;
}

// vt: the V table, whose rows include the diagonal (columns up to min(N, M) instead of min(N - 1, M))
// (sum_C: 0 a storing fill; 2 or 4: a summing fill whose cell lists are laid out for strips of that many columns per lane)
static hb_geom hb_geometry(unsigned N, unsigned M, int D, int sum_C = 0, bool vt = false) {
  const bool summing = sum_C != 0;
  hb_geom g;
  memset(&g, 0, sizeof(g));
  g.ok = false;
  if (N < 3 || M < 2 || D < 1 || N >= (1u << 20)) return g;
  const int cus = stb_cu_count();
  // 2 columns per lane walk faster (26 against 32.5 ns a row) but need 2.6 times the spine waves of 4 (80
  // against 208 own columns a strip): 2 while the spine waves of all tables fit on ~80 compute units at one per
  // SIMD.  (MI355X, tools/ab_ck.py: N = M = 10^4, 1 table 0.36 against 0.46-0.50 ms, 2 tables 0.46 against 0.49,
  // 4 tables 0.78 against 0.61; N = M = 4000, 3 tables 0.198 against 0.206, 8 tables 0.31 against 0.25.)
  {
    const unsigned cmax0 = vt ? ((M < N) ? M : N) : ((M < N - 1) ? M : N - 1);
    const uint64_t waves2 = (uint64_t)D * ((cmax0 - 1 + 79) / 80);
    g.C = stb_env_int("STB_HB_C", waves2 <= (uint64_t)cus * 5 / 4 ? 2 : 4);  // (320 on 256 compute units)
  }
  if (g.C != 1 && g.C != 2 && g.C != 4) g.C = 2;
  // (a summing fill's cell lists are laid out for one strip shape, whatever the number of discounts: stb_hb_sum_C)
  if (summing) g.C = (sum_C == 2 || sum_C == 3) ? sum_C : 4;
  g.P = 4;  // (one spine wave per SIMD: two on one slow each other by a third; see below)
  // a block is a renormalisation period (or less): rows in eights, halo lanes R / C <= 32
  int Pc = stb_period_rows(N);
  const int Penv = stb_env_int("STB_FILL_P", 0);
  if (Penv > 0 && Penv < Pc) Pc = Penv;
  int R = stb_env_int("STB_HB_ROWS", 48);
  if (R > Pc) R = Pc;
  if (R > (hb_mhl(g.C) - 1) * g.C) R = (hb_mhl(g.C) - 1) * g.C;
  R = R / 8 * 8;
  if (g.C == 3) R = R / 24 * 24;  // (halo lanes R / 3)
  if (R < 8) return g;
  g.R = R;
  // (halo: R columns -- a halo column k from the left is exact for k rows, so the own columns are through all R rows of
  // a block.  The V table divides an own cell by its left neighbour IN THE SAME ROW, which for a strip's first own
  // column is the last halo column: one lane more keeps that one exact in the block's last row too.)
  g.HL = R / g.C + (vt ? 1 : 0);
  if (g.HL > hb_mhl(g.C)) return g;
  g.U = 64 - g.HL;
  const int UC = g.U * g.C;
  const unsigned cmax = vt ? ((M < N) ? M : N) : ((M < N - 1) ? M : N - 1);  // columns 2..cmax hold stored cells: elements 0 .. cmax - 2
  g.JW = (int)((cmax - 1 + UC - 1) / UC);
  if (g.JW < 1) g.JW = 1;
  // Strips per spine workgroup.  4 -- a spine wave per SIMD -- while the spine decides; once a storing fill's spine
  // workgroups would take more than 100 compute units, the tile workers decide (the spine is through long before
  // them), and 7 strips a workgroup -- 0.56 of the compute units for a spine a third slower -- serve them
  // better: 8 tables of 10^4 columns 0.825-0.845 against 0.84-0.865 ms, 12 tables 1.20 against 1.25, 16 tables
  // 1.39 against 1.43, but 6 tables 0.72 against 0.645, 4 tables 0.65 against 0.53, and the fused grid of 8
  // discounts, whose tiles store nothing, 0.83 against 0.65.
  {
    const int b4 = (g.JW + 3) / 4;
    // (a summing fill's tiles are cheap: its spine keeps a wave per SIMD while all its workgroups fit on the chip)
    // (230 and 100 on the 256 compute units of an MI355X)
    g.P = stb_env_int(summing ? "STB_HB_DOT_P" : "STB_HB_P", ((int64_t)b4 * D > (summing ? cus * 9 / 10 : cus * 2 / 5)) ? 7 : 4);
    if (g.P < 1 || g.P > HB_PMAX) g.P = 4;
  }
  g.B = (g.JW + g.P - 1) / g.P;
  g.NB = (int)((N - 1 + R - 1) / R);  // the state before block b is row 1 + b R
  if (g.JW >= 65535 || g.NB >= 65536) return g;
  uint64_t nt = 0;
  for (int j = 0; j < g.JW; j++) {
    const int b0 = hb_first_block(j, UC, R);
    if (b0 >= g.NB) return g;  // (cannot happen: column 2 + j U C <= N - 1)
    nt += (uint64_t)(g.NB - b0);
  }
  if (nt >= (1ull << 31)) return g;
  g.n_tiles = (unsigned)nt;
  g.n_rec = g.n_tiles + (unsigned)g.NB;
  size_t o = HB_HDR_BYTES;
  g.off_prog = o;
  o += stb_align_up((size_t)D * g.JW * HB_PROG_STRIDE * sizeof(unsigned), 256);
  g.off_cke = o;
  o += stb_align_up((size_t)D * g.n_rec * g.U * sizeof(unsigned), 256);
  g.off_ckv = o;
  o += stb_align_up((size_t)D * g.n_rec * g.U * g.C * 8, 256);
  g.zero_bytes = o;
  g.bytes = o;
  g.ok = true;
  return g;
}

bool stb_hb_eligible(unsigned N, unsigned M, int D) { return hb_geometry(N, M, D).ok; }
bool stb_hb_eligible_out(unsigned N, unsigned M, int D, int out_kind) {  // ... storing floats or V ratios: 2 or 4 columns per lane
  const hb_geom g = hb_geometry(N, M, D, false, (out_kind & 2) != 0);
  return g.ok && (out_kind == 0 || g.C >= 2);
}
// spine workgroups a fill of D tables launches (4 strips each, or 7 once there would be more than 100)
unsigned stb_hb_spine(unsigned N, unsigned M, int D) {
  const hb_geom g = hb_geometry(N, M, D);
  return g.ok ? (unsigned)g.B * (unsigned)D : 0xffffffffu;
}

int stb_hb_tuning(unsigned N, unsigned M, int D, int *W_out, int *rows_out) {
  const hb_geom g = hb_geometry(N, M, D);
  if (W_out) *W_out = g.U * g.C;
  if (rows_out) *rows_out = g.R;
  return g.ok ? 0 : 1;
}

size_t stb_hb_workspace(unsigned N, unsigned M, int D) {
  const hb_geom g0 = hb_geometry(N, M, D);
  if (!g0.ok) return 0;
  // (the strip shape and the block length are tunables: room for the most records any of them needs --
  // blocks of 8 rows, strips of 32 lanes -- is too much to ask for; size for the shapes the defaults and
  // the tests use, and let stb_launch_hb refuse what does not fit)
  size_t need = g0.bytes;
  static const int shapes[4] = {1, 2, 3, 4};
  static const int rows[4] = {16, 24, 32, 48};
  int Pc = stb_period_rows(N);
  for (int c : shapes)
    for (int r0 : rows) {
      int R = std::min(std::min(r0, Pc), (hb_mhl(c) - 1) * c) / 8 * 8;
      if (c == 3) R = R / 24 * 24;
      if (R < 8) continue;
      const int HL = R / c + 1, U = 64 - HL, UC = U * c;  // (the V table's strips: a halo lane more, see hb_geometry)
      const unsigned cmax = (M < N) ? M : N;  // (the V table's: one column more than the S table's)
      const size_t JW = (cmax - 1 + UC - 1) / UC, NB = (N - 1 + R - 1) / R;
      size_t nrec = NB;
      for (size_t j = 0; j < JW; j++) {
        const size_t b0 = (size_t)hb_first_block((int)j, UC, R);
        nrec += (b0 < NB) ? NB - b0 : 0;
      }
      const size_t b = 1024 + HB_HDR_BYTES + (size_t)D * JW * 4 * HB_PROG_STRIDE + (size_t)D * nrec * U * (4 + 8 * c) + 1024;
      if (b > need) need = b;
    }
  return need + 256;
}

// the order in which the tiles of a table become ready, as j | b << 16 -- the spine writes the record of
// block b of strip j at about b R r + (j / P) L + (j % P) lag -- and the first record of every strip
struct hb_order_entry {
  int dev;
  unsigned N, M;
  int C, P, R, NB, JW, U, split;
  unsigned n_order;
  int r_ns, L_ns, lag_ns;
  unsigned *d_buf;  // [JW + 2] rec_off, then [n_tiles] order
  unsigned long long used;  // the look-up that last handed it out
};
static std::mutex g_hb_mu;
static std::vector<hb_order_entry> g_hb_orders;
static unsigned long long g_hb_lookups = 0;

// split: the tiles of every strip's last `split` blocks are listed four times, with a quarter of the rows each (storing
// fills of blocks of 48 rows, 2 or 4 columns per lane, fewer than 2^14 strips; 0 for everything else)
static int hb_order_list(const hb_geom &g, unsigned N, unsigned M, const unsigned **rec_off, const unsigned **order, int split = 0,
                         unsigned *n_order_out = nullptr) {
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  const int r_ns = stb_env_int("STB_HB_ORDER_R", 30), L_ns = stb_env_int("STB_HB_ORDER_L", 3000),
            lag_ns = stb_env_int("STB_HB_ORDER_LAG", 300);
  std::lock_guard<std::mutex> lock(g_hb_mu);
  g_hb_lookups++;
  for (hb_order_entry &e : g_hb_orders)
    if (e.dev == dev && e.N == N && e.M == M && e.C == g.C && e.P == g.P && e.R == g.R && e.NB == g.NB && e.JW == g.JW && e.U == g.U &&
        e.split == split && e.r_ns == r_ns && e.L_ns == L_ns && e.lag_ns == lag_ns) {
      e.used = g_hb_lookups;
      *rec_off = e.d_buf;
      *order = e.d_buf + g.JW + 2;
      if (n_order_out) *n_order_out = e.n_order;
      return 0;
    }
  struct item {
    long key;
    unsigned code;
  };
  const int UC = g.U * g.C;
  std::vector<unsigned> buf((size_t)g.JW + 2);
  std::vector<item> v;
  v.reserve(g.n_tiles + (size_t)3 * split * g.JW);
  unsigned off = 0;
  buf[0] = off;  // strip index 0: the halo of strip 0, blocks 0 .. NB - 1
  off += (unsigned)g.NB;
  for (int j = 0; j < g.JW; j++) {
    const int b0 = hb_first_block(j, UC, g.R);
    buf[j + 1] = off;
    off += (unsigned)(g.NB - b0);
    for (int b = b0; b < g.NB; b++) {
      item it;
      it.key = (long)b * g.R * r_ns + (long)(j / g.P) * L_ns + (long)(j % g.P) * lag_ns;
      it.code = (unsigned)j | ((unsigned)b << 16);
      v.push_back(it);
      if (split && b >= g.NB - split)
        for (unsigned q = 1; q < 4; q++) {
          it.code = (unsigned)j | (q << 14) | ((unsigned)b << 16);
          v.push_back(it);
        }
    }
  }
  buf[g.JW + 1] = off;
  if (off != g.n_rec || (!split && v.size() != g.n_tiles)) return stb_fail("stb_fill_S: record count %u != %u", off, g.n_rec);
  std::stable_sort(v.begin(), v.end(), [](const item &x, const item &y) { return x.key < y.key; });
  buf.resize((size_t)g.JW + 2 + v.size());
  for (size_t i = 0; i < v.size(); i++) buf[(size_t)g.JW + 2 + i] = v[i].code;
  hb_order_entry e;
  e.dev = dev;
  e.N = N;
  e.M = M;
  e.C = g.C;
  e.P = g.P;
  e.R = g.R;
  e.NB = g.NB;
  e.JW = g.JW;
  e.U = g.U;  // (the V table's strips have a halo lane more than the S table's: other first blocks for the same N, M)
  e.split = split;
  e.n_order = (unsigned)v.size();
  e.r_ns = r_ns;
  e.L_ns = L_ns;
  e.lag_ns = lag_ns;
  e.d_buf = nullptr;
  HIPCHK(hipMalloc((void **)&e.d_buf, buf.size() * sizeof(unsigned) + 16));
  HIPCHK(hipMemcpy(e.d_buf, buf.data(), buf.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  // (Shapes come and go: the table is kept small, the least recently used entry goes.  A caller holds the pointers it
  // was handed without this lock until its kernel is launched -- another host thread may be in here meanwhile -- so an
  // entry handed out within the last 32 look-ups is never freed: the table grows instead.)
  e.used = g_hb_lookups;
  if (g_hb_orders.size() >= 48) {
    size_t old = g_hb_orders.size();
    for (size_t i = 0; i < g_hb_orders.size(); i++)
      if (g_hb_orders[i].used + 32 < g_hb_lookups && (old == g_hb_orders.size() || g_hb_orders[i].used < g_hb_orders[old].used)) old = i;
    if (old < g_hb_orders.size()) {
      (void)hipFree(g_hb_orders[old].d_buf);
      g_hb_orders.erase(g_hb_orders.begin() + (long)old);
    }
  }
  g_hb_orders.push_back(e);
  *rec_off = e.d_buf;
  *order = e.d_buf + g.JW + 2;
  if (n_order_out) *n_order_out = e.n_order;
  return 0;
}

// what the builder of a summing fill's cell lists has to know: the shape of the tiles and where a strip's
// records start (device array of JW + 2 words; a tile's record index is its item base)
// columns per lane of a summing fill's strips for a set that is evaluated with up to Dmax discounts: 2 -- the faster walk,
// 22 against 33 ns a row, which is what a grid of a few discounts waits for -- while the spine waves of Dmax tables get a
// SIMD each with room to spare for the tile workers (MI355X, kernel ms, 2 against 4 columns: N = 10^4, 2 discounts 0.296
// against 0.331, 4: 0.337 against 0.333; N = 4000, 2: 0.156 against 0.164, 3: 0.137 against 0.145); 4 beyond
int stb_hb_sum_C(unsigned N, unsigned M, int Dmax) {
  // 2 columns per lane while the spine waves of Dmax tables get a SIMD each with room to spare; 3 -- strips of 144 columns
  // behind 16 halo lanes, a row of 9 instructions instead of 11 -- while theirs fit on ~140 compute units; 4 beyond.
  // (MI355X, kernel ms with 2 / 3 / 4 columns, N = M = 10^4: 3 discounts 0.283 / 0.301 / 0.320, 4: 0.338 / 0.301 / 0.323,
  // 5: 0.347 / 0.305 / 0.327, 6: - / 0.305 / 0.331, 8: 0.59 / 0.310 / 0.327, 9-10: - / 0.335 / 0.336, 12: - / 0.96 / 0.354;
  // N = 4000: 8 discounts 0.153 / 0.136 / 0.144, 16: - / 0.150 / 0.156.)
  const unsigned cmax = (M < N - 1) ? M : N - 1;
  const uint64_t D = (uint64_t)(Dmax > 0 ? Dmax : 1), cus = (uint64_t)stb_cu_count();
  const uint64_t waves2 = D * ((cmax - 1 + 79) / 80), waves3 = D * ((cmax - 1 + 143) / 144);
  const int c = stb_env_int("STB_HB_DOT_C", waves2 <= cus * 3 / 2 ? 2 : (waves3 <= cus * 9 / 4 ? 3 : 4));
  return (c == 2 || c == 3) ? c : 4;
}

int stb_hb_dot_info(unsigned N, unsigned M, int D, hb_dot_info *out, int sum_C) {
  const hb_geom g = hb_geometry(N, M, D, (sum_C == 2 || sum_C == 3) ? sum_C : 4);
  if (!g.ok || g.R % HB_DOT_GR != 0 || g.R / HB_DOT_GR > HB_DOT_NQ) return 1;
  const unsigned *rec_off = nullptr, *order = nullptr;
  if (hb_order_list(g, N, M, &rec_off, &order)) return 1;
  out->R = g.R;
  out->UC = g.U * g.C;
  out->HC = g.HL * g.C;
  out->NB = g.NB;
  out->JW = g.JW;
  out->NQ = HB_DOT_NQ;
  out->G = HB_DOT_GR;
  out->C = g.C;
  out->n_tiles = g.n_tiles;
  out->n_rec = g.n_rec;
  out->n_spine = (unsigned)g.B * (unsigned)D;
  out->rec_off = rec_off;
  return 0;
}

int stb_launch_hb(fill_args &A, int D, char *ws, size_t ws_left, const dot_request *dot, unsigned **hdr_out, hipStream_t st, int out_kind) {
  const unsigned N = A.N, M = A.M;
  if (out_kind < 0 || out_kind > 3 || (out_kind && dot)) return stb_fail("stb_fill: output kind %d", out_kind);
  const hb_geom g = hb_geometry(N, M, D, dot ? ((dot->geom_C == 2 || dot->geom_C == 3) ? dot->geom_C : 4) : 0, (out_kind & 2) != 0);
  if (!g.ok) return stb_fail("stb_fill_S: the halo-block form does not take N=%u M=%u D=%d", N, M, D);
  if (g.bytes > ws_left) return stb_fail("stb_fill_S: workspace too small for the halo-block form (%zu > %zu)", g.bytes, ws_left);
  if (dot && (!dot->item_ptr || dot->col0 != 3))
    return stb_fail("stb_fill_S: the halo-block form sums over cell lists built for its tiles");
  if (dot && (g.R % HB_DOT_GR != 0 || g.R / HB_DOT_GR > HB_DOT_NQ)) return stb_fail("stb_fill_S: blocks of %d rows do not suit the summing form", g.R);
  hb_args X;
  memset(&X, 0, sizeof(X));
  X.hdr = (unsigned *)ws;
  X.progress = (unsigned *)(ws + g.off_prog);
  X.ck_e = (unsigned *)(ws + g.off_cke);
  X.ck_v = (unsigned long long *)(ws + g.off_ckv);
  X.n_rec = g.n_rec;
  X.D = D;
  X.B = g.B;
  X.JW = g.JW;
  X.NB = g.NB;
  X.P = g.P;
  X.R = g.R;
  X.HL = g.HL;
  X.U = g.U;
  X.n_tiles = g.n_tiles;
  X.n_spine = (unsigned)g.B * (unsigned)D;
  X.timeout = (unsigned long long)stb_env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;  // wall_clock64: 100 MHz
  X.poll_nap = stb_env_int("STB_HB_POLL_NAP", 4);
  if (X.poll_nap < 1) X.poll_nap = 1;
  // a waiting worker sleeps this many x 512 cycles per block its tile is away: about half of what the spine takes
  // for a block (48 rows x 22 ns with 2 columns per lane, x 32-45 ns with 4), so that it looks again well before
  // the tile is due -- longer and it oversleeps (one table: 0.33 ms with 6 against 0.31 with 3; a table of 4000
  // rows 0.166 against 0.148), much shorter and the looks of thousands of waiting waves get in the spine's way
  X.nap_block = stb_env_int("STB_HB_NAP_BLOCK", g.C >= 3 ? (dot ? 6 : 4) : 3);
  X.diag = stb_env_int("STB_HB_DIAG", 0);
  // (the waves of a spine workgroup that have no strip work on tiles from the start in the SUMMING form, whose tile workers
  // are what it waits for from 8 discounts on -- N = M = 10^4, kernel ms with / without: 3 discounts at N = 4000 0.125 / 0.134,
  // 8: 0.348 / 0.361, 12: 0.414 / 0.438, 16: 0.542 / 0.582, 24: 0.822 / 0.843, 1-6: alike -- not in the storing fills, where
  // they take issue slots from a spine that decides: 8 tables 0.74-0.80 against 0.80)
  X.spare_work = stb_env_int("STB_HB_SPARE", dot ? 64 : 0);
  // (measured, MI355X, N = M = 10^4: one table 0.365 ms with it against 0.321 without -- a late start is never made up,
  // and every strip's is up to 512 cycles late; 8 tables 0.74-0.80 against 0.80: off unless asked for)
  X.doze = stb_env_int("STB_HB_DOZE", 0);
  if (dot) {
    X.item_ptr = dot->item_ptr;
    X.ent_pos = dot->ent_pos;
    X.ent_cnt = dot->ent_cnt;
    X.dotp = dot->dotp;
    const_cast<dot_request *>(dot)->parts_per_table = (int)g.n_tiles;
    if (dot->dotp_cap && (size_t)D * g.n_tiles > dot->dotp_cap) return stb_fail("stb_groups_aterms: partial-sum buffer too small");
  }
  // (quartered last tiles: see the workers; the V table's blocks may be a row group out of step with a quarter -- 48 rows are
  // 12 groups of 4 or 24 of 2 either way)
  // Only where the spine decides (strips of 2 columns per lane: a table or two): with 8 tables the workers are busy to the end
  // and the rows walked twice cost more than the shorter tail saves (MI355X, N = M = 10^4, ms: one table, floats 0.289 against
  // 0.302, V 0.303 against 0.314, N = 4000 0.134 against 0.137; 8 tables 0.826-0.850 against 0.811; the last 4, 6 or 10 blocks: alike).
  X.split = (!dot && g.R == 48 && g.C >= 2 && g.JW < (1 << 14)) ? std::min(stb_env_int("STB_HB_SPLIT", g.C == 2 ? 6 : 0), g.NB) : 0;
  if (X.split < 0) X.split = 0;
  X.n_order = g.n_tiles;
  if (hb_order_list(g, N, M, &X.rec_off, &X.order, X.split, &X.n_order)) return 1;
  const char *tl_file = getenv("STB_HB_TIMELINE");
#ifdef HB_TL_FINE
  const size_t dbg_words = (size_t)g.JW * (g.NB + 2) + (size_t)X.n_order * 4 + (size_t)g.JW * g.NB * 8 + (size_t)g.JW * g.NB * 2;
#else
  const size_t dbg_words = (size_t)g.JW * (g.NB + 2) + (size_t)X.n_order * 4;
#endif
  if (tl_file && *tl_file) {
    HIPCHK(hipMalloc((void **)&X.dbg, dbg_words * 8));
    HIPCHK(hipMemsetAsync(X.dbg, 0, dbg_words * 8, st));
  }
  {
    const bool need_zero = !dot || dot->ws_zero < g.zero_bytes;
    const bool need_s1 = !(dot && dot->no_s1) && !(out_kind & 2);  // (the V table has no column 1)
    if (need_zero && need_s1 && !(g.zero_bytes & 15) && !((uintptr_t)ws & 15) && stb_env_int("STB_HB_PREP", 1)) {
      if (stb_launch_prep(A, D, ws, g.zero_bytes, st)) return 1;
    } else {
      if (stb_a_flush(A, st)) return 1;
      if (need_zero) HIPCHK(hipMemsetAsync(ws, 0, g.zero_bytes, st));
      if (need_s1 && stb_launch_s1(A, D, st)) return 1;
    }
  }
  if (dot) const_cast<dot_request *>(dot)->zero_bytes = g.zero_bytes;
  *hdr_out = X.hdr;
  // every workgroup is generic: the first B*D tickets walk the spine, the others work on tiles
  const int cus = stb_cu_count();
  // A storing fill takes the whole chip: idle workers cost nothing any more (they wait on progress words that have
  // a line each, and ask counters of their own for tiles).  A summing fill's tiles are cheap and its spine decides:
  // ~D N / 250 worker workgroups keep up with it, and fewer waiting waves leave it the memory side (2 discounts:
  // 0.36 ms with 126 workgroups against 0.38 with 256).
  const int per_cu = stb_env_int("STB_HB_WG_PER_CU", 1);
  unsigned grid = (unsigned)(cus * per_cu);
  const unsigned min_workers = (unsigned)stb_env_int("STB_HB_MIN_WORKERS", 48);
  if (dot) {
    const uint64_t fed = (uint64_t)D * N / 250;
    const uint64_t want = (uint64_t)X.n_spine + (fed > min_workers ? fed : min_workers);
    if (want < grid) grid = (unsigned)want;
  }
  if (grid < X.n_spine + min_workers) grid = X.n_spine + min_workers;
  if (stb_env_int("STB_HB_GRID", 0) > 0) grid = (unsigned)stb_env_int("STB_HB_GRID", 0);  // (diagnostic: spine alone)
  // the tile order goes to LDS when it fits beside the rest (a storing kernel's ~80 KB; a summing kernel's staging
  // rows leave no room for it)
  X.order_lds = (!dot && X.n_order <= HB_ORDER_LDS) ? 1 : 0;
  {
    // at least 8 counters when the tables are few; every counter's part of the order list keeps hundreds of tiles
    const int Dg = D < HB_MAXCNT ? D : HB_MAXCNT;
    int Sn = stb_env_int("STB_HB_TICKET_PARTS", Dg >= 8 ? 1 : (8 + Dg - 1) / Dg);
    while (Sn > 1 && ((unsigned)Sn * 64 > X.n_order || Sn * Dg > HB_MAXCNT)) Sn--;
    if (Sn < 1) Sn = 1;
    X.n_cnt = Sn * Dg;
  }
  if (X.n_cnt < 1 || X.n_cnt > HB_MAXCNT) return stb_fail("stb_fill_S: %d ticket counters", X.n_cnt);
  if (dot) {
    const int nw = g.C >= 3 ? HB_NW_DOT4 : HB_NW;
    const size_t shm = (size_t)nw * 4 * 64 * g.C * sizeof(double);
    switch (g.C) {
      case 4: STB_LAUNCH_SHM((k_fill_hb<4, 1>), dim3(grid), dim3(64 * nw), shm, st, A, X); break;
      case 3: STB_LAUNCH_SHM((k_fill_hb<3, 1>), dim3(grid), dim3(64 * nw), shm, st, A, X); break;
      case 2: STB_LAUNCH_SHM((k_fill_hb<2, 1>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
      default: return stb_fail("stb_fill_S: no summing halo-block kernel for %d columns per lane", g.C);
    }
  } else {
    const size_t shm = X.order_lds ? (size_t)X.n_order * sizeof(unsigned) : 0;
    if (out_kind && g.C == 1) return stb_fail("stb_fill: the halo-block form narrows or divides with 2 or 4 columns per lane");
    switch (g.C * 4 + out_kind) {
      case 4: STB_LAUNCH_SHM((k_fill_hb<1, 0>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
      case 8: STB_LAUNCH_SHM((k_fill_hb<2, 0>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
      case 9: STB_LAUNCH_SHM((k_fill_hb<2, 0, 1>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
      case 10: STB_LAUNCH_SHM((k_fill_hb<2, 0, 2>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
      case 11: STB_LAUNCH_SHM((k_fill_hb<2, 0, 3>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
      case 16: STB_LAUNCH_SHM((k_fill_hb<4, 0>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
      case 17: STB_LAUNCH_SHM((k_fill_hb<4, 0, 1>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
      case 18: STB_LAUNCH_SHM((k_fill_hb<4, 0, 2>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
      default: STB_LAUNCH_SHM((k_fill_hb<4, 0, 3>), dim3(grid), dim3(64 * HB_NW), shm, st, A, X); break;
    }
  }
  HIPCHK(hipGetLastError());
  if (X.dbg) {
    HIPCHK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(dbg_words);
    HIPCHK(hipMemcpy(h.data(), X.dbg, dbg_words * 8, hipMemcpyDeviceToHost));
    (void)hipFree(X.dbg);
    FILE *f = fopen(tl_file, "wb");
    if (f) {
      const int hd[8] = {g.JW, g.NB, (int)X.n_order, g.C, g.P, g.R, g.U, D | (X.split << 16)};
      fwrite(hd, sizeof(int), 8, f);
      std::vector<unsigned> ord(X.n_order);
      HIPCHK(hipMemcpy(ord.data(), X.order, (size_t)X.n_order * sizeof(unsigned), hipMemcpyDeviceToHost));
      fwrite(ord.data(), sizeof(unsigned), X.n_order, f);
      fwrite(h.data(), 8, dbg_words, f);
      fclose(f);
    }
  }
  return 0;
}

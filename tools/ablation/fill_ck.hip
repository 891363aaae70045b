// fill_ck.hip -- the checkpointed form of the S-table fill: ONE launch, a latency chain that only
// carries the recurrence, and tile workers that do everything else.
//
// Replaces the table part of S_remake_part's double-S branch (reference lib/stable.c:321-388):
//   S^n_m = (n-1-m a) S^{n-1}_m + S^{n-1}_{m-1},   stored as log S^n_m for 2 <= m <= min(n-1, M).
//
// Why another form.  Rows are strictly sequential, so some wave has to walk all N rows of the first
// columns, one dependent step per row: that chain (N x row time + strips x hand-off) is the floor of
// a fill.  In k_fill_chain the wave that walks it also feeds, through an LDS ring, the waves that turn
// its cells into logs and store them, and those waves live in the same workgroup: the strip at
// column 1 converts 1.6 x the average strip's cells on one compute unit, and its producer runs at
// the pace its consumers leave it.  Here the two jobs are separated:
//
//   spine    P waves of a workgroup walk 64*C-column wave strips of one table for all rows, C adjacent
//            columns per lane in block-floating form (see k_fill_chain), and do NOTHING but the
//            recurrence: per row 2 DPP moves, 1 multiply, C fma, C adds and one 8-byte LDS post of
//            the strip's last column.  What they hand on: (i) that last column per row, to the next
//            spine wave through LDS and -- by the workgroup's publisher wave -- to HBM as 8-byte
//            granules (the edge stream of the strip); (ii) every RB trips (a block, ~96 rows) the
//            whole row of significands and lane exponents (a checkpoint), straight to HBM.
//   workers  every other wave of the grid.  A worker takes a tile (table d, wave strip jw, block b)
//            from a ticket, waits until the spine has passed it, loads the tile's checkpoint and the
//            left strip's edge stream for its rows, and recomputes the tile's ~96 rows x 64*C columns
//            in registers -- the same fma sequence -- converting every row to logs and storing it as it
//            goes.  No LDS ring, no waits inside a tile, no neighbours: tiles are independent, so the
//            log work and the table's HBM traffic are spread evenly over the chip whatever the
//            shape of the table (a triangle) or the number of tables.
//
// Column 1 (S^n_1 = Gamma(n-a)/Gamma(1-a), the S1 vector) is not a table column.  Wave strips start
// at column 2 -- element 0 of a row -- so a lane's C cells are one aligned 8*C-byte store and a
// wave's row segment is whole 128-byte lines.  The left input of column 2 comes from k_col1, a
// prefix product over n written in the edge-stream format of a virtual strip "-1" before the fill
// starts: every strip, the first included, has a left neighbour and is treated alike.
//
// Everything handed from one workgroup to another is self-flagging 8-byte (or 4-byte) granules
// written with write-through stores and read with L1-bypassing loads: 0 means "not written yet"
// (an exact zero travels as -0.0; exponents carry an offset).  The counters that order the tiles
// (progress per strip) are hints for scheduling only.  Every wait is bounded; on expiry the waiter
// records an error in the header and everybody runs to the end (stb_fill_status repeats the fill
// with k_fill_pc).

#include <mutex>
#include <type_traits>
#include <vector>

#include "fill_chain.h"

#define CK_U CH_U      // rows per trip
#define CK_RE 32       // trips in the rings between spine waves and from the fetcher (power of two)
#define CK_NW 8        // waves per workgroup
#define CK_EOFF32 (1u << 30)
#define CK_MAXRB 16    // trips per block at most (worker edge buffer: 128 rows)
#define CK_ORDER_LDS 8192  // tiles of a table whose order list is copied to LDS (32 KB)
#ifndef CK_DIAG
#define CK_DIAG 0      // diagnostic builds (results wrong): 1 no edge posts, 2 no left inputs, 4 no left counter, 8 no progress post
#endif
#ifndef CK_LOOK
#define CK_LOOK 4      // the row of a trip after which the next trip's left inputs and the left counter are read
#endif

struct ck_args {
  unsigned *hdr;               // [0] role ticket, [1] error code, [2] error detail, [3] tile ticket; zeroed per fill
  unsigned long long *edge_v;  // [D][JW+1][EV]  last column of a wave strip, indexed by row (strip 0: column 1)
  unsigned long long *edge_e;  // [D][JW+1][NP]  its lane exponent + CH_EOFF, indexed by trip + 1
  unsigned long long *ck_v;    // [D][JW][NBK][64*C]  checkpoint: significands at the start of a block
  unsigned *ck_e;              // [D][JW][NBK][64]    ... and lane exponents + CK_EOFF32
  unsigned *progress;          // [D][JW]  blocks a spine wave has finished (scheduling hint)
  unsigned *cu_busy;           // [4096]   spine waves walking on a compute unit (key: XCC id, SE/SH/CU id), or null
  const unsigned *order;       // [n_tiles] jw | b << 16, in the order the tiles become ready
  uint64_t EV, NP;
  int D, B, JW, NBK;           // tables, spine workgroups per table, wave strips per table, blocks
  int TP, RB, G;               // trips per period, trips per block, trips in all (rows 3 .. 2 + 8 G)
  unsigned n_tiles;            // per table
  unsigned n_spine;            // spine workgroups in all (B * D)
  unsigned long long timeout;  // wall_clock64 ticks a wait may last
  int poll_nap;
  int pub_nap;                 // s_sleep argument between two looks of the publisher
  unsigned tp_magic;           // ceil(2^32 / TP): trip / TP = umulhi(trip, tp_magic)
  int spare_work;              // 1: the spare waves of a spine workgroup work as tile workers meanwhile
  int diag;                    // STB_CK_DIAG: 1 the workers wait for the whole spine
  // DOT kernels (aterms without a table, lib/samplea.c:68-80): the cells that occur among the (n,t) pairs,
  // grouped per (trip, 64-column slice counted from column 2) item, and where the sums go
  const unsigned *item_ptr;        // [trips * nsg + 1] first entry of every item
  const unsigned short *ent_pos;   // row-in-trip << 6 | column-in-slice of each occurring cell
  const unsigned *ent_cnt;         // its occurrence count
  unsigned nsg;                    // slices per trip in item_ptr
  double *dotp;                    // [D][n_tiles] sum of count * log S per tile
  unsigned long long *dbg;     // STB_CK_TIMELINE: wall-clock stamps, table 0: [JW][NBK + 2] spine (start, block ends, end),
                               // then [n_tiles][4] workers (claimed, inputs loaded, done, hardware id)
};

__device__ __forceinline__ int ck_first_trip(int c) { return (c <= 3) ? 0 : (c - 3) / CK_U; }

typedef double ck_double2 __attribute__((ext_vector_type(2)));

// aligned 16-byte store at (wave-uniform base) + (per-lane byte offset); see store_sbase
__device__ __forceinline__ void store_sbase16(const void *sbase, unsigned byte_off, double x, double y) {
  unsigned long long base_copy;
  ck_double2 v2 = {x, y};
  asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, %0\n\ts_nop 1"
               : "=&s"(base_copy)
               : "v"(byte_off), "v"(v2), "s"(sbase)
               : "memory");
}

// 8-byte LDS store at `addr` + OFF bytes (a statement the compiler neither splits nor predicates)
template <int OFF>
__device__ __forceinline__ void lds_store1(unsigned addr, double x) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(x), "n"(OFF) : "memory");
}

// ---- the log of a block-floating cell, eight cells at a time, stage-major (as in k_fill_chain) ----
template <int NCELL>
__device__ __forceinline__ void ck_logs(const double (&x)[NCELL], int myep, const double2 *lt, int one_hi,
                                        double (&val)[NCELL]) {
  double z[NCELL], kf[NCELL], r[NCELL], pl[NCELL];
  double2 tt[NCELL];
#pragma unroll
  for (int u = 0; u < NCELL; u++) tt[u] = lt[(__double2hiint(x[u]) >> 13) & 127];
#pragma unroll
  for (int u = 0; u < NCELL; u++) {
    const int hi = __double2hiint(x[u]);
    z[u] = __hiloint2double(mantissa_of_one(hi, one_hi), __double2loint(x[u]));
    kf[u] = (double)((int)((hi >> 20) & 0x7ff) - 1023 + myep);
  }
#pragma unroll
  for (int u = 0; u < NCELL; u++) r[u] = fma(z[u], tt[u].x, -1.0);
#pragma unroll
  for (int u = 0; u < NCELL; u++) pl[u] = fma(r[u], 0.2, -0.25);
#pragma unroll
  for (int u = 0; u < NCELL; u++) pl[u] = fma(r[u], pl[u], 1.0 / 3.0);
#pragma unroll
  for (int u = 0; u < NCELL; u++) pl[u] = fma(r[u], pl[u], -0.5);
#pragma unroll
  for (int u = 0; u < NCELL; u++) pl[u] = fma(r[u], pl[u], 1.0);
#pragma unroll
  for (int u = 0; u < NCELL; u++) val[u] = fma(kf[u], 0.693147180559945309417, fma(r[u], pl[u], tt[u].y));
}

// renormalise a lane: the largest of its C significands back to 2^-PC_BIAS * [0.5,1)
template <int C>
__device__ __forceinline__ void ck_renorm(double (&v)[C], int &ep) {
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

template <int C, int P, int DOT>
__global__ __launch_bounds__(64 * CK_NW, 4) void k_fill_ck(fill_args A, ck_args X) {
  constexpr int U = CK_U, RE = CK_RE;
  constexpr int WS = 64 * C;  // columns of a wave strip
  static_assert(C == 1 || C == 2 || C == 4, "columns per lane");
  static_assert(P >= 1 && P <= 4, "spine waves per workgroup");
  __shared__ double2 lt[128];
  // spine workgroups
  __shared__ __attribute__((aligned(16))) double xedge[4][RE * U];  // last column of spine wave w: row r at entry r mod 256
  __shared__ int xexp[4][16];                                       // its lane exponent per period (ring of 16)
  __shared__ __attribute__((aligned(16))) double edge_in[RE * U];   // what the fetcher delivers to spine wave 0
  __shared__ int edge_e[RE];
  __shared__ __attribute__((aligned(16))) double pad8[64 + RE * U + 8];  // where lanes 0..62 of a masked post go
  __shared__ int pad4[64 + RE + 8];
  __shared__ int prod_done[4], pub_done[4], edge_ready, s_abort, s_awake;
  __shared__ int post_pad[4][64];
  __shared__ unsigned s_ticket;
  // workers: the left inputs of a tile, per wave
  __shared__ __attribute__((aligned(16))) double w_le[CK_NW][CK_MAXRB * U];
  __shared__ int w_lee[CK_NW][CK_MAXRB];
  // dynamic segment.  Storing form: the tile order, when it fits (a ticket then costs no dependent global
  // load).  Summing form: per wave a trip's 8 rows x WS significands, and the first 64 cells of every trip's
  // list of the tile in hand (position, count); see ck_dyn_lds().
  extern __shared__ __attribute__((aligned(16))) double ck_dyn[];
  unsigned *s_order = reinterpret_cast<unsigned *>(ck_dyn);
  double *ck_stage = ck_dyn;
  unsigned *ck_ecnt = reinterpret_cast<unsigned *>(ck_dyn + (size_t)CK_NW * U * WS);
  unsigned short *ck_epos = reinterpret_cast<unsigned short *>(ck_ecnt + (size_t)CK_NW * CK_MAXRB * 64);
  __shared__ int w_se[DOT ? CK_NW : 1][64];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_ticket = atomicAdd(X.hdr, 1u);
  if (tid < 128) lt[tid] = A.lt[tid];
  for (int i = tid; i < RE * U; i += blockDim.x) edge_in[i] = 0.0;
  const bool order_in_lds = DOT == 0 && X.n_tiles <= CK_ORDER_LDS;
  if (order_in_lds)
    for (unsigned i = tid; i < X.n_tiles; i += blockDim.x) s_order[i] = X.order[i];
  __syncthreads();
  const unsigned ticket = s_ticket;
  unsigned *cu_busy = nullptr;
  if (X.cu_busy) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    cu_busy = X.cu_busy + (((xcc & 15u) << 8) | ((hw >> 8) & 0xffu));
  }
  const unsigned N = A.N, M = A.M;
  const int TP = X.TP, RB = X.RB, G = X.G;

  if (ticket < X.n_spine) {
    // =========================================================================================
    // spine workgroup: wave strips j*P .. j*P + P - 1 of table d
    const int j = (int)(ticket / (unsigned)X.D);
    const int d = (int)(ticket % (unsigned)X.D);
    const int jw0 = j * P;
    const int g0b = ck_first_trip(2 + jw0 * WS);
    if (tid < 4) {
      const int jw = jw0 + tid;
      const int g0 = (jw < X.JW) ? ck_first_trip(2 + jw * WS) : G;
      prod_done[tid] = g0;
      pub_done[tid] = g0;
    }
    if (tid == 0) {
      edge_ready = g0b;
      s_abort = 0;
      s_awake = (j == 0) ? 1 : 0;
    }
    __syncthreads();
    const unsigned who = (unsigned)(j | (d << 16));
    bool aborted = false;
    auto wait_ge = [&](const int *cnt, int need, unsigned code, int nap) {
      if (aborted || lds_peek(cnt) >= need) {
        asm volatile("" ::: "memory");  // (what the counter guards is read or written after it)
        return;
      }
      if (!chain_wait_slow(cnt, need, &s_abort, X.hdr, X.timeout, code, who, nap)) aborted = true;
      asm volatile("" ::: "memory");
    };
    auto doze = [&]() {
      while (!lds_peek(&s_awake) && !lds_peek(&s_abort)) __builtin_amdgcn_s_sleep(16);
    };
    const size_t strip0 = (size_t)d * (X.JW + 1);  // index of the virtual strip (column 1) of table d

    if (wave < P) {
      // ================= spine waves =================
      doze();
      __builtin_amdgcn_s_setprio(3);
      auto spine = [&](auto wc) {
        constexpr int w = decltype(wc)::value;
        const int jw = jw0 + w;
        if (jw >= X.JW) return;
        const int c0w = 2 + jw * WS;
        const int g0w = ck_first_trip(c0w);
        const double a = A.a[d];
        const int cl = c0w + lane * C;
        double v[C], coef[C];
#pragma unroll
        for (int i = 0; i < C; i++) {
          const int c = cl + i;
          v[i] = (c == 2) ? ldexp(1.0, -1 - PC_BIAS) : 0.0;  // row 2: S^2_2 = 1, nothing to its right
          coef[i] = (double)(2 + g0w * U) - (double)c * a;   // n - 1 - c a for the first row of trip g0w
        }
        double s = 1.0;
        int ep = 1 + PC_BIAS;
        int p = g0w / TP, tin = g0w - p * TP;
        int bnext = g0w / RB + 1;  // the next block boundary is trip bnext * RB
        unsigned long long *dbg = (X.dbg && d == 0 && lane == 0) ? X.dbg + (size_t)jw * (X.NBK + 2) : nullptr;
        if (dbg) dbg[0] = wall_clock64();
        if (cu_busy && lane == 0) atomicAdd(cu_busy, 1u);
        const bool has_next = (w < P - 1) && (jw + 1 < X.JW);
        const int *left_cnt = (w == 0) ? &edge_ready : &prod_done[w > 0 ? w - 1 : 0];
        const int *next_cnt = &prod_done[(w < P - 1) ? w + 1 : w];
        // the ring the left inputs come from
        const double *left_ring = (w == 0) ? edge_in : &xedge[w > 0 ? w - 1 : 0][0];
        const int *left_exp = &xexp[w > 0 ? w - 1 : 0][0];
        // masked posts: lane 63 writes the ring, lane 0 the counter, the other lanes scratch
        const unsigned edge_base = (lane == 63) ? lds_addr_of(&xedge[w][0]) : lds_addr_of(&pad8[lane]);
        int *exp_base = (lane == 63) ? &xexp[w][0] : &pad4[lane];
        int *post_addr = (lane == 0) ? &prod_done[w] : &post_pad[w][lane];
        unsigned long long *ckv = X.ck_v + (((size_t)d * X.JW + jw) * X.NBK) * (64 * C) + lane * C;
        unsigned *cke = X.ck_e + (((size_t)d * X.JW + jw) * X.NBK) * 64 + lane;
        unsigned *prog = X.progress + (size_t)d * X.JW + jw;
        int n_left;
        auto load_left = [&](double(&x)[U], int g) {
          if (w == 0) {
            // (the fetcher stores a row at the entry of the row it feeds: aligned)
            const ck_double2 *src = reinterpret_cast<const ck_double2 *>(left_ring + (g & (RE - 1)) * U);
#pragma unroll
            for (int q = 0; q < U / 2; q++) {
              const ck_double2 t2 = src[q];
              x[2 * q] = t2.x;
              x[2 * q + 1] = t2.y;
            }
          } else {
            // the spine wave to the left stores row r at entry r mod 256: the input of row u is entry 8 g + u - 1
            x[0] = left_ring[(g * U - 1) & (RE * U - 1)];
#pragma unroll
            for (int u = 1; u < U; u++) x[u] = left_ring[(g & (RE - 1)) * U + u - 1];
          }
        };
        // A lone wave pays ~20 cycles for an LDS store and 35-50 for a branch on a vector compare
        // (tools/ubench/spine.hip: 16 ns of arithmetic a row, 9 for a ds_write_b64 a row, 8 for three tests
        // a trip), so: two rows go out with one ds_write2_b64, and everything that can stop a trip -- the
        // left neighbour not far enough (its counter is read with the next trip's inputs at row CK_LOOK),
        // a period or block boundary, the ring guard every 8th trip -- sits behind ONE branch.
        auto trip = [&](int g, double(&e)[U], double(&en)[U]) {
          const int stop = (int)(n_left < g + 1) | (int)(tin == 0) | (int)((g & 7) == 0) | (int)(g == g0w);
          if (__builtin_expect(stop != 0, 0)) {
            if (n_left < g + 1) {
              wait_ge(left_cnt, g + 1, 0x100u, 1);
              asm volatile("" ::: "memory");
              load_left(e, g);
            }
            if ((g & 7) == 0) {
              if (has_next) wait_ge(next_cnt, g - RE + 9, 0x400u, 1);
              wait_ge(&pub_done[w], g - RE + 8, 0x500u, 1);
            }
            if (g == g0w || tin == 0) {
              if (g != g0w) {
                if (g == bnext * RB) {
                  // ---- checkpoint: the row as it stands before trip g, for the worker of block bnext ----
                  unsigned long long *dst = ckv + (size_t)bnext * (64 * C);
#pragma unroll
                  for (int i = 0; i < C; i++) {
                    unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
                    if ((b << 1) == 0) b = CH_NEGZERO;
                    __hip_atomic_store(dst + i, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  }
                  __hip_atomic_store(cke + (size_t)bnext * 64, (unsigned)ep + CK_EOFF32, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
                  if (lane == 0) __hip_atomic_store(prog, (unsigned)bnext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  if (dbg) dbg[bnext] = wall_clock64();
                  bnext++;
                }
                ck_renorm<C>(v, ep);
              }
              // freeze the scale of the cross-lane input for the period (bounds: see k_fill_pc)
              int el;
              if (w == 0) {
                el = edge_e[g & (RE - 1)];
              } else {
                el = left_exp[p & 15];
                // the row above the first row of a period was produced under the previous period's exponent
                if (tin == 0 && p > 0) e[0] = ldexp(e[0], left_exp[(p - 1) & 15] - el);
              }
              int dl = wave_shr1(ep, ep) - ep;
              if (lane == 0) dl = el - ep;
              s = ldexp(1.0, min(max(dl, -1100), 220));
              exp_base[p & 15] = ep;
            }
          }
          const unsigned wa = edge_base + (unsigned)((g & (RE - 1)) * U * 8);
          double hold = 0.0;  // (the even row of a pair, until its odd row is there)
          auto row = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const double t0 = wave_shr1(v[C - 1], e[u]) * s;
#pragma unroll
            for (int i = C - 1; i >= 1; i--) v[i] = fma(coef[i], v[i], v[i - 1]);
            v[0] = fma(coef[0], v[0], t0);
#pragma unroll
            for (int i = 0; i < C; i++) coef[i] += 1.0;
            if constexpr (!(CK_DIAG & 1)) {
              if constexpr ((u & 1) == 0) hold = v[C - 1];
              else lds_store2<u - 1>(wa, hold, v[C - 1]);
            }
            if constexpr (u == CK_LOOK) {
              // what the next trip needs from the left is read now, under this trip's arithmetic
              if constexpr (!(CK_DIAG & 4)) n_left = lds_peek(left_cnt);
              asm volatile("" ::: "memory");
              if constexpr (!(CK_DIAG & 2)) load_left(en, g + 1);
            }
          };
          static_assert(U == 8, "rows of a trip");
          row(std::integral_constant<int, 0>{});
          row(std::integral_constant<int, 1>{});
          row(std::integral_constant<int, 2>{});
          row(std::integral_constant<int, 3>{});
          row(std::integral_constant<int, 4>{});
          row(std::integral_constant<int, 5>{});
          row(std::integral_constant<int, 6>{});
          row(std::integral_constant<int, 7>{});
          asm volatile("" ::: "memory");
          if constexpr (!(CK_DIAG & 8)) __hip_atomic_store(post_addr, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
          if (++tin == TP) {
            tin = 0;
            p++;
          }
        };
        double ea[U], eb[U];
        n_left = lds_peek(left_cnt);
        asm volatile("" ::: "memory");
        load_left(ea, g0w);
        int g = g0w;
        for (; g + 1 < G; g += 2) {
          trip(g, ea, eb);
          trip(g + 1, eb, ea);
        }
        if (g < G) trip(g, ea, eb);
        if (lane == 0) __hip_atomic_store(prog, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (dbg) dbg[X.NBK + 1] = wall_clock64();
        if (lane == 0) atomicAdd(X.hdr + 4, 1u);  // (spine waves that are through: diagnostics only)
        if (cu_busy && lane == 0) atomicSub(cu_busy, 1u);
      };
      if (P == 1 || wave == 0) spine(std::integral_constant<int, 0>{});
      else if (wave == 1) spine(std::integral_constant<int, (P >= 2) ? 1 : 0>{});
      else if (wave == 2) spine(std::integral_constant<int, (P >= 3) ? 2 : 0>{});
      else spine(std::integral_constant<int, (P >= 4) ? 3 : 0>{});
      __builtin_amdgcn_s_setprio(0);
    } else if (wave == P) {
      // ================= fetcher: the left strip's edge stream -> LDS, for spine wave 0 =================
      const unsigned long long *ev_in = X.edge_v + (strip0 + jw0) * X.EV;      // strip jw0 - 1 (virtual for j = 0)
      const unsigned long long *ee_in = X.edge_e + (strip0 + jw0) * X.NP + 1;  // [t] for trip t >= -1
      struct edge_poll {
        unsigned long long va, e1, e0;
        int tb, nt;
      };
      const int ka = lane >> 3;
      int t = g0b;  // trips below it are delivered
      if (j > 0) {
        // the left strip publishes from its own first trip on: wait, dozing, for the exponent granule of
        // a trip shortly before the first one needed here, then wake the workgroup
        const unsigned long long *probe = ee_in + max(g0b - 6, 0);
        unsigned long long t_begin = 0;
        unsigned spins = 0;
        while (__hip_atomic_load(probe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(16);
          if ((++spins & 255u) != 0 && X.timeout != 0) continue;
          if (t_begin == 0) t_begin = wall_clock64();
          if (__hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || lds_peek(&s_abort) ||
              (unsigned long long)wall_clock64() - t_begin >= X.timeout)
            break;  // (the main loop below gives up properly)
        }
        lds_post(&s_awake, 1);
      }
      int pd = lds_peek(&prod_done[0]);
      auto issue = [&](edge_poll &q) {
        int lim = pd + RE - 1;  // trips below it may be written: their ring slots were read by the spine
        if (lim > G) lim = G;
        q.tb = t;
        q.nt = max(0, min(8, lim - t));
        q.va = __hip_atomic_load(ev_in + 2 + t * U + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q.e1 = __hip_atomic_load(ee_in + t + ka, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q.e0 = __hip_atomic_load(ee_in + t + ka - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto settle = [&](const edge_poll &q) {
        const unsigned long long miss = ~__ballot(q.va != 0 && q.e1 != 0 && q.e0 != 0);
        const int nr = min(q.nt, miss ? (int)(__builtin_ctzll(miss) >> 3) : 8);
        const int cur = t;
        if (q.tb + nr <= cur) return false;
        const int ex = (int)(long long)(q.e1 - CH_EOFF);
        double xa = __longlong_as_double((long long)q.va);
        if (ka < nr && q.tb + ka >= cur) {
          if ((lane & 7) == 0) {
            xa = ldexp(xa, (int)(long long)(q.e0 - CH_EOFF) - ex);
            edge_e[(q.tb + ka) & (RE - 1)] = ex;
          }
          edge_in[((q.tb + ka) & (RE - 1)) * U + (lane & 7)] = xa;
        }
        t = q.tb + nr;
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(&edge_ready, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        return true;
      };
      auto pause = [&]() {
        if (X.poll_nap >= 4) __builtin_amdgcn_s_sleep(4);
        else if (X.poll_nap == 3) __builtin_amdgcn_s_sleep(3);
        else if (X.poll_nap == 2) __builtin_amdgcn_s_sleep(2);
        else __builtin_amdgcn_s_sleep(1);
      };
      constexpr int FK = 2;
      edge_poll q[FK];
      unsigned long long t_begin = 0;
      bool timing = false;
      unsigned idle = 0;
      while (t < G) {
#pragma unroll
        for (int i = 0; i < FK; i++) {
          issue(q[i]);
          if (i == FK - 1) pd = lds_peek(&prod_done[0]);
          pause();
        }
        bool any = false;
#pragma unroll
        for (int i = 0; i < FK; i++) any = settle(q[i]) || any;
        if (any) {
          timing = false;
          idle = 0;
          continue;
        }
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
          lds_post(&edge_ready, 0x7fffffff);  // release the spine: it runs on with stale edges
          break;
        }
      }
    } else if (wave == P + 1) {
      // ================= publisher: the spine waves' last columns -> HBM =================
      // 16 lanes per spine wave: lanes 0..7 the rows of a trip, lane 8 its exponent
      doze();
      const int grp = lane >> 4, sub = lane & 15;
      const int jw = jw0 + grp;
      const bool mine = grp < P && jw < X.JW && sub <= U;
      int tw = mine ? ck_first_trip(2 + jw * WS) : G;
      // (nobody reads the last strip's edge: its trips are only released)
      const bool stores = mine && jw + 1 < X.JW;
      unsigned long long *ev_out = X.edge_v + (strip0 + 1 + (mine ? jw : 0)) * X.EV;
      unsigned long long *ee_out = X.edge_e + (strip0 + 1 + (mine ? jw : 0)) * X.NP + 1;
      unsigned idle = 0;
      unsigned long long t_begin = 0;
      bool timing = false;
      while (__any(tw < G)) {
        const int pdw = (tw < G) ? lds_peek(&prod_done[grp & 3]) : 0;
        asm volatile("" ::: "memory");
        const bool go = tw < G && tw < pdw;
        if (go) {
          if (stores) {
            unsigned long long b;
            unsigned long long *dst;
            if (sub < U) {
              b = (unsigned long long)__double_as_longlong(xedge[grp & 3][(tw & (RE - 1)) * U + sub]);
              if ((b << 1) == 0) b = CH_NEGZERO;
              dst = ev_out + 3 + tw * U + sub;
            } else {
              const int pw = (TP == 1) ? tw : (int)__umulhi((unsigned)tw, X.tp_magic);
              b = (unsigned long long)((long long)xexp[grp & 3][pw & 15] + (long long)CH_EOFF);
              dst = ee_out + tw;
            }
            __hip_atomic_store(dst, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          tw++;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ring entries are in registers
          if (sub == 0) __hip_atomic_store(&pub_done[grp & 3], tw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (__any(go)) {
          idle = 0;
          timing = false;
          continue;
        }
        // (a trip takes the spine ~0.3 us: looking more often only takes issue slots from it)
        if (X.pub_nap >= 8) __builtin_amdgcn_s_sleep(8);
        else if (X.pub_nap >= 4) __builtin_amdgcn_s_sleep(4);
        else if (X.pub_nap >= 2) __builtin_amdgcn_s_sleep(2);
        else __builtin_amdgcn_s_sleep(1);
        if ((++idle & 63) != 0) continue;
        if (lds_peek(&s_abort)) {
          // the spine runs to its end whatever happens: keep releasing its ring without the clock
          continue;
        }
        if (!timing) {
          timing = true;
          t_begin = wall_clock64();
        } else if (X.timeout != 0 && (unsigned long long)wall_clock64() - t_begin >= 16 * X.timeout) {
          break;  // (cannot happen unless a spine wave died)
        }
      }
    } else if (!X.spare_work) {
      // spare waves: asleep while the spine walks (they would take issue slots from it), workers afterwards
      while (lds_peek(&prod_done[0]) < G && !lds_peek(&s_abort)) __builtin_amdgcn_s_sleep(127);
    }
    // Whoever is through with its part of the spine works on tiles -- but only once every workgroup of the
    // grid has started (each takes a ticket when it does): while spine workgroups are still waiting for a
    // free compute unit, this one must give its place up, or its waves would sit on tiles of strips whose
    // spine cannot start.
    if (__hip_atomic_load(X.hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) return;
  }

  // =========================================================================================
  // tile workers (every wave for itself)
  {
    const int wslot = wave;
    double *le = &w_le[wslot][0];
    int *lee = &w_lee[wslot][0];
    int one_hi = 0x3ff00000;
    asm volatile("" : "+v"(one_hi));
    const unsigned total = X.n_tiles * (unsigned)X.D;
    if (X.diag & 1) {
      // diagnostic: the workers start when every spine wave is through (what the tiles cost with the chip to themselves)
      unsigned spins = 0;
      while (__hip_atomic_load(X.hdr + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(X.JW * X.D) && ++spins < 400000u)
        __builtin_amdgcn_s_sleep(100);
    }
    for (;;) {
      unsigned k = 0;
      if (cu_busy) {
        // A spine wave runs at the pace of a lone wave and anything else that issues on its compute unit
        // slows it (8 tables: 1.5 ms with the spare waves of the spine's workgroups working, 1.2 without):
        // while one walks here, this wave sleeps.  The other compute units carry two workgroups of workers.
        unsigned spins = 0;
        while (__hip_atomic_load(cu_busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __builtin_amdgcn_s_sleep(127);
          if ((++spins & 63u) == 0 && __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        }
      }
      if (lane == 0) k = atomicAdd(X.hdr + 3, 1u);
      k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
      if (k >= total) break;
      const int d = (int)(k % (unsigned)X.D);
      const unsigned oi = k / (unsigned)X.D;
      const unsigned ord = (unsigned)__builtin_amdgcn_readfirstlane((int)(order_in_lds ? s_order[oi] : X.order[oi]));
      const int jw = (int)(ord & 0xffffu), b = (int)(ord >> 16);
      const int c0w = 2 + jw * WS;
      const int g0w = ck_first_trip(c0w);
      const int gs = max(b * RB, g0w), ge = min((b + 1) * RB, G);
      const int nt = ge - gs, nrows = nt * U;
      const unsigned who = (unsigned)jw | ((unsigned)d << 16);
      unsigned long long *wdbg =
          (X.dbg && d == 0 && lane == 0) ? X.dbg + (size_t)X.JW * (X.NBK + 2) + (size_t)(k / (unsigned)X.D) * 4 : nullptr;
      if (wdbg) wdbg[0] = wall_clock64();
      // ---- the tile's inputs: left edges (rows 2 + 8 gs ..) and the checkpoint.  Everything is asked for at
      // once, without looking at the spine's progress first: what has been written is non-zero.  Only when
      // something is missing is the progress word read, to sleep about as long as the missing blocks take. ----
      const size_t sleft = (size_t)d * (X.JW + 1) + jw;  // strip jw - 1 (+1: the virtual strip is index 0)
      const unsigned long long *ev = X.edge_v + sleft * X.EV + 2 + gs * U;
      const unsigned long long *ee = X.edge_e + sleft * X.NP + 1;
      const bool fresh = (gs == g0w);  // the strip's first tile starts from the empty row
      const unsigned long long *ckv = X.ck_v + (((size_t)d * X.JW + jw) * X.NBK + b) * (64 * C) + lane * C;
      const unsigned *cke = X.ck_e + (((size_t)d * X.JW + jw) * X.NBK + b) * 64 + lane;
      const unsigned *prog = X.progress + (size_t)d * X.JW + jw;
      double v[C], coef[C];
      int ep = 1 + PC_BIAS;
      bool ok = true;
      {
        unsigned spins = 0;
        unsigned long long t_begin = 0;
        const bool two = nrows > 64;
        const int k1 = 64 + lane;
        const bool act0 = lane < nrows, act1 = two && k1 < nrows;
        const int kc0 = act0 ? lane : 0, kc1 = act1 ? k1 : 0;
        const int tc0 = gs + (kc0 >> 3), tc1 = gs + (kc1 >> 3);
        for (;;) {
          const unsigned long long va0 = __hip_atomic_load(ev + kc0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long e10 = __hip_atomic_load(ee + tc0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long e00 = __hip_atomic_load(ee + tc0 - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned long long va1 = 1, e11 = 1, e01 = 1, bv[C];
          unsigned be = 1;
          if (two) {
            va1 = __hip_atomic_load(ev + kc1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            e11 = __hip_atomic_load(ee + tc1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            e01 = __hip_atomic_load(ee + tc1 - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int i = 0; i < C; i++) bv[i] = 1;
          if (!fresh) {
#pragma unroll
            for (int i = 0; i < C; i++) bv[i] = __hip_atomic_load(ckv + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            be = __hip_atomic_load(cke, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          bool have = va0 != 0 && e10 != 0 && e00 != 0 && va1 != 0 && e11 != 0 && e01 != 0 && be != 0;
#pragma unroll
          for (int i = 0; i < C; i++) have = have && bv[i] != 0;
          if (__all(have)) {
            auto put = [&](bool act, int kc, unsigned long long va, unsigned long long e1, unsigned long long e0) {
              if (!act) return;
              const int ex = (int)(long long)(e1 - CH_EOFF);
              double xa = __longlong_as_double((long long)va);
              if ((kc & 7) == 0) {
                xa = ldexp(xa, (int)(long long)(e0 - CH_EOFF) - ex);
                lee[kc >> 3] = ex;
              }
              le[kc] = xa;
            };
            put(act0, kc0, va0, e10, e00);
            put(act1, kc1, va1, e11, e01);
            if (!fresh) {
#pragma unroll
              for (int i = 0; i < C; i++) v[i] = __longlong_as_double((long long)bv[i]);
              ep = (int)(be - CK_EOFF32);
            }
            break;
          }
          // Thousands of waves wait here while a table's first rows are walked, and every poll is a
          // request to the memory side that the spine's own hand-offs queue behind: sleep for about as
          // long as the blocks still missing take (a block: ~3 us), at most ~50 us, then look again.
          const unsigned done = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int missing = (done >= (unsigned)(b + 1)) ? 0 : min((int)((unsigned)(b + 1) - done), 16);
          if (missing == 0) __builtin_amdgcn_s_sleep(8);
          for (int i = 0; i < missing; i++) __builtin_amdgcn_s_sleep(100);
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
      }
      if (!ok) break;
      if (wdbg) wdbg[1] = wall_clock64();
      const double a = A.a[d];
      const int cl = c0w + lane * C;
#pragma unroll
      for (int i = 0; i < C; i++) {
        const int c = cl + i;
        if (fresh) v[i] = (c == 2) ? ldexp(1.0, -1 - PC_BIAS) : 0.0;
        coef[i] = (double)(2 + gs * U) - (double)c * a;
      }
      if constexpr (DOT != 0) {
        // ---- the tile as a sum: no logs but those of the cells that occur, nothing stored ----
        static_assert(DOT == 0 || C <= 2, "the staging area of a DOT worker holds 8 rows of at most 128 columns");
        double *stage = ck_stage + (size_t)wslot * (U * WS);
        int *se = &w_se[DOT ? wslot : 0][0];
        // first entry of every (trip, slice) item of the tile and of the item after the trip's last: lane
        // tt * (C + 1) + sl holds item_ptr[(gs + tt) * nsg + jw * C + sl]
        unsigned ip = 0;
        if (lane < nt * (C + 1)) ip = X.item_ptr[(size_t)(gs + lane / (C + 1)) * X.nsg + (unsigned)(jw * C + lane % (C + 1))];
        double s = 1.0, acc = 0.0;
        int tin = gs % TP;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (nothing of the tile has to be walked beyond the last trip that has an occurring cell)
        const unsigned ipn = (unsigned)__shfl_down((int)ip, C);
        const unsigned long long hm = __ballot(lane < nt * (C + 1) && lane % (C + 1) == 0 && ipn != ip);
        const int ge2 = hm ? gs + (63 - __builtin_clzll(hm)) / (C + 1) + 1 : gs;
        auto range = [&](int tt, unsigned &b0, unsigned &b1, unsigned &mid) {
          b0 = (unsigned)__builtin_amdgcn_readlane((int)ip, tt * (C + 1));
          b1 = (unsigned)__builtin_amdgcn_readlane((int)ip, tt * (C + 1) + C);
          mid = (C == 2) ? (unsigned)__builtin_amdgcn_readlane((int)ip, tt * (C + 1) + 1) : b1;
        };
        // The first 64 cells of every trip's list are asked for now, all at once (a list is fetched in ~1.5 us,
        // a trip walked in 0.2: one trip ahead is not early enough), and parked in LDS.
        unsigned short *epos = ck_epos + (size_t)wslot * (CK_MAXRB * 64);
        unsigned *ecnt = ck_ecnt + (size_t)wslot * (CK_MAXRB * 64);
        {
          unsigned short pp[CK_MAXRB];
          unsigned cc[CK_MAXRB];
#pragma unroll
          for (int tt = 0; tt < CK_MAXRB; tt++) {
            pp[tt] = 0;
            cc[tt] = 0;
            if (tt < ge2 - gs) {
              unsigned b0, b1, mid;
              range(tt, b0, b1, mid);
              if (b0 + lane < b1) {
                pp[tt] = X.ent_pos[b0 + lane];
                cc[tt] = X.ent_cnt[b0 + lane];
              }
            }
          }
#pragma unroll
          for (int tt = 0; tt < CK_MAXRB; tt++)
            if (tt < ge2 - gs) {
              epos[tt * 64 + lane] = pp[tt];
              ecnt[tt * 64 + lane] = cc[tt];
            }
        }
        for (int g = gs; g < ge2; g++) {
          if (g == g0w || tin == 0) {
            if (g != g0w) ck_renorm<C>(v, ep);
            const int el = lee[g - gs];
            int dl = wave_shr1(ep, ep) - ep;
            if (lane == 0) dl = el - ep;
            s = ldexp(1.0, min(max(dl, -1100), 220));
            se[lane] = ep;
          }
          unsigned b0, b1, mid;
          range(g - gs, b0, b1, mid);
          unsigned pos = epos[(g - gs) * 64 + lane], cnt = ecnt[(g - gs) * 64 + lane];
          const double *lrow = le + (g - gs) * U;
#pragma unroll
          for (int u = 0; u < U; u++) {
            const double t0 = wave_shr1(v[C - 1], lrow[u]) * s;
#pragma unroll
            for (int i = C - 1; i >= 1; i--) v[i] = fma(coef[i], v[i], v[i - 1]);
            v[0] = fma(coef[0], v[0], t0);
#pragma unroll
            for (int i = 0; i < C; i++) coef[i] += 1.0;
            if (b0 != b1) {  // (a trip none of whose cells occurs is only walked)
              if constexpr (C == 2) *reinterpret_cast<ck_double2 *>(stage + u * WS + lane * 2) = ck_double2{v[0], v[1]};
              else stage[u * WS + lane] = v[0];
            }
          }
          if (b0 != b1) {
            unsigned kk = b0 + lane;
            for (;;) {
              if (kk < b1) {
                const int cx = ((C == 2 && kk >= mid) ? 64 : 0) + (int)(pos & 63u);
                const double val = bfp_log(stage[(pos >> 6) * WS + cx], se[cx / C], lt);
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
          if (++tin == TP) tin = 0;
        }
        // fixed-shape tree over the wave: the same bits on every run, whoever computed the tile
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) X.dotp[(size_t)d * X.n_tiles + oi] = acc;
      } else {
      double s = 1.0;
      int tin = gs % TP;
      double *table = A.tables + (uint64_t)d * A.tstride;
      const unsigned voff = (unsigned)(lane * C) * 8u;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's own LDS writes above)
      for (int g = gs; g < ge; g++) {
        if (g == g0w || tin == 0) {
          if (g != g0w) ck_renorm<C>(v, ep);
          const int el = lee[g - gs];
          int dl = wave_shr1(ep, ep) - ep;
          if (lane == 0) dl = el - ep;
          s = ldexp(1.0, min(max(dl, -1100), 220));
        }
        const int r0 = 3 + g * U;
        const unsigned pitch = stb_row_pitch((unsigned)r0, M);
        const bool fast = (unsigned)(r0 + U - 1) <= N;
        double *rowp = table + stb_row_offset((unsigned)r0, M) + (size_t)(c0w - 2);
        const double *lrow = le + (g - gs) * U;
        double e[U];
#pragma unroll
        for (int u = 0; u < U; u++) e[u] = lrow[u];
        constexpr int RS = 8 / C;  // rows converted together: eight cells in flight
#pragma unroll
        for (int h = 0; h < U; h += RS) {
          double x[8], val[8];
#pragma unroll
          for (int u = 0; u < RS; u++) {
            const double t0 = wave_shr1(v[C - 1], e[h + u]) * s;
#pragma unroll
            for (int i = C - 1; i >= 1; i--) v[i] = fma(coef[i], v[i], v[i - 1]);
            v[0] = fma(coef[0], v[0], t0);
#pragma unroll
            for (int i = 0; i < C; i++) {
              coef[i] += 1.0;
              x[u * C + i] = v[i];
            }
          }
          ck_logs<8>(x, ep, lt, one_hi, val);
#pragma unroll
          for (int u = 0; u < RS; u++) {
            if constexpr ((CK_DIAG & 16) != 0) {
              // (diagnostic: the values are kept alive, the table is not written except for a tile's last row)
              if (g == ge - 1 && h + u == U - 1) store_sbase(rowp + (size_t)(h + u) * pitch, voff, val[u * C] + val[u * C + C - 1]);
              else asm volatile("" ::"v"(val[u * C]), "v"(val[u * C + C - 1]));
            } else if (fast || (unsigned)(r0 + h + u) <= N) {
              double *rp = rowp + (size_t)(h + u) * pitch;
              if constexpr (C == 1) {
                store_sbase(rp, voff, val[u]);
              } else if constexpr (C == 2) {
                store_sbase16(rp, voff, val[u * 2], val[u * 2 + 1]);
              } else {
                store_sbase16(rp, voff, val[u * 4], val[u * 4 + 1]);
                store_sbase16(rp + 2, voff, val[u * 4 + 2], val[u * 4 + 3]);
              }
            }
          }
        }
        if (++tin == TP) tin = 0;
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

// ---- column 1 as an edge stream: x_n = S^n_1 = prod_{k=1}^{n-1} (k - a), rows 2 .. R ---------------
//
// One workgroup per table.  Every thread multiplies up a contiguous chunk of rows as (mantissa,
// exponent) pairs, the chunk totals are scanned across the workgroup, and the rows are written in the
// format the spine's fetcher and the workers read: the significand relative to the exponent of the
// row's trip -- frozen per period, so that a period starts at 2^-PC_BIAS * [0.5,1) -- as an 8-byte
// granule per row, and that exponent + CH_EOFF per trip.
struct me_t {
  double m;
  int e;
};
__device__ __forceinline__ me_t me_mul(me_t x, me_t y) {
  double m = x.m * y.m;
  const int k = __builtin_amdgcn_frexp_exp(m);
  me_t r;
  r.m = __builtin_amdgcn_frexp_mant(m);
  r.e = x.e + y.e + k;
  return r;
}

__global__ __launch_bounds__(1024) void k_col1(const double *a_dev, ck_args X, int *period_e, int R) {
  // rows 1 .. R in 1024 contiguous chunks, one per thread; period_e: [D][periods + 2] scratch
  __shared__ double sm[16];
  __shared__ int se[16];
  const int d = blockIdx.x, t = threadIdx.x;
  const double a = a_dev[d];
  const int chunk = (R + 1023) / 1024;
  const int n_lo = 1 + t * chunk, n_hi = min(R, n_lo + chunk - 1);
  const int TP = X.TP;
  const int periods = (X.G + TP - 1) / TP;
  int *pe = period_e + (size_t)d * (periods + 2);  // [0]: the row before the table (trip -1), [1 + p]: period p
  // factor of row n: x_n = x_{n-1} * (n - 1 - a) for n >= 2, x_1 = 1
  me_t acc = {0.5, 1};  // 1.0
  for (int n = n_lo; n <= n_hi; n++) {
    if (n >= 2) {
      me_t f = {(double)(n - 1) - a, 0};
      acc = me_mul(acc, f);
    }
  }
  // exclusive scan of the chunk totals: inside the wave with shuffles, across the 16 waves through LDS
  // (barriers of a 1024-thread workgroup are what this kernel's time is made of: three, not twenty)
  const int lane = t & 63, wv = t >> 6;
  me_t inc = acc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    me_t y = {__shfl_up(inc.m, off), __shfl_up(inc.e, off)};
    if (lane >= off) inc = me_mul(inc, y);
  }
  if (lane == 63) {
    sm[wv] = inc.m;
    se[wv] = inc.e;
  }
  __syncthreads();
  me_t wp = {0.5, 1};
  for (int w = 0; w < wv; w++) {
    me_t y = {sm[w], se[w]};
    wp = me_mul(wp, y);
  }
  me_t up = {__shfl_up(inc.m, 1), __shfl_up(inc.e, 1)};
  me_t pre0 = (lane == 0) ? wp : me_mul(wp, up);
  // the exponent a period is written under is that of its first row, 3 + 8 TP p (of row 2 for the row
  // before the table): whoever owns that row says so
  // (rows are walked with running counters: an integer division per row would cost more than the row)
  const int PR = CK_U * TP;  // rows of a period
  me_t pre = pre0;
  int p_of = (n_lo >= 3) ? (n_lo - 3) / PR : -1, r_in = (n_lo >= 3) ? (n_lo - 3) - p_of * PR : 0;  // period of row n, row within it
  for (int n = n_lo; n <= n_hi; n++) {
    if (n >= 2) {
      me_t f = {(double)(n - 1) - a, 0};
      pre = me_mul(pre, f);
    }
    if (n == 2) pe[0] = pre.e + PC_BIAS;
    if (n >= 3) {
      if (r_in == 0) pe[1 + p_of] = pre.e + PC_BIAS;
      if (++r_in == PR) {
        r_in = 0;
        p_of++;
      }
    } else if (n == 2) {
      p_of = 0;
      r_in = 0;
    }
  }
  __threadfence_block();
  __syncthreads();
  unsigned long long *ev = X.edge_v + (size_t)d * (X.JW + 1) * X.EV;
  unsigned long long *ee = X.edge_e + (size_t)d * (X.JW + 1) * X.NP + 1;
  // the chunk once more, now written out: the significand relative to its trip's exponent as an 8-byte
  // granule per row, the exponent + CH_EOFF per trip (by the owner of the trip's first row)
  pre = pre0;
  p_of = (n_lo >= 3) ? (n_lo - 3) / PR : -1;
  r_in = (n_lo >= 3) ? (n_lo - 3) - p_of * PR : 0;
  int cur_p = -2, E = 0;
  for (int n = n_lo; n <= n_hi; n++) {
    if (n >= 2) {
      me_t f = {(double)(n - 1) - a, 0};
      pre = me_mul(pre, f);
      const int p = (n >= 3) ? p_of : -1;
      if (p != cur_p) {
        cur_p = p;
        E = pe[1 + p];
      }
      ev[n] = (unsigned long long)__double_as_longlong(ldexp(pre.m, pre.e - E));
      if (n == 2) ee[-1] = (unsigned long long)((long long)E + (long long)CH_EOFF);
      else if ((r_in & (CK_U - 1)) == 0) ee[p_of * TP + (r_in >> 3)] = (unsigned long long)((long long)E + (long long)CH_EOFF);
    }
    if (n >= 3) {
      if (++r_in == PR) {
        r_in = 0;
        p_of++;
      }
    } else if (n == 2) {
      p_of = 0;
      r_in = 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side

static inline int ck_first_trip_h(int c) { return (c <= 3) ? 0 : (c - 3) / CK_U; }

struct ck_geom {
  int C, P, B, JW, G, TP, RB, NBK, R;
  uint64_t EV, NP;
  unsigned n_tiles;
  size_t off_cu, off_prog, off_ee, off_ev, off_cke, off_ckv, zero_bytes, off_tm, off_te, bytes;
  bool ok;
};

// strip shape: C columns per lane, P spine waves per workgroup, blocks of RB trips
static ck_geom ck_geometry(unsigned N, unsigned M, int D, bool summing = false) {
  ck_geom g;
  memset(&g, 0, sizeof(g));
  g.ok = false;
  if (N < 3 || M < 2 || D < 1 || N >= (1u << 20)) return g;
  // (MI355X, tools/sweep_forms.sh: at 10^4 columns 4 columns a lane win from 2 tables on, 0.98 against 1.03 ms at
  // 8; at 4000 columns 2 win up to 16 tables, 0.36 against 0.38 ms, and lose at 32, 0.62 against 0.58)
  const uint64_t total_cells = (uint64_t)D * stb_table_cells(N, M);
  g.C = stb_env_int("STB_CK_C", total_cells < 150000000ull ? 2 : 4);
  if (g.C != 1 && g.C != 2 && g.C != 4) g.C = 2;
  // (a summing fill stores nothing: its tiles are cheap, the spine decides, and narrow strips walk faster;
  // its workers stage 8 rows of a strip in LDS, which holds them for up to 128 columns)
  if (summing) g.C = (stb_env_int("STB_CK_DOT_C", 2) == 1) ? 1 : 2;
  g.P = stb_env_int("STB_CK_P", 4);
  if (g.P < 1 || g.P > 4) g.P = 4;
  int Pc = stb_period_rows(N);
  const int Penv = stb_env_int("STB_FILL_P", 0);
  if (Penv > 0 && Penv < Pc) Pc = Penv;
  g.TP = Pc / CK_U;
  if (g.TP < 2) return g;  // (the exponent ring of 16 periods must span the CK_RE trips two waves may be apart)
  if (g.TP > CK_MAXRB) g.TP = CK_MAXRB;
  const int target = stb_env_int("STB_CK_BLOCK_ROWS", 96) / CK_U;
  int K = (target + g.TP / 2) / g.TP;
  if (K < 1) K = 1;
  while (K > 1 && K * g.TP > CK_MAXRB) K--;
  g.RB = K * g.TP;
  const unsigned cmax = (M < N - 1) ? M : N - 1;  // columns 2..cmax hold stored cells
  const int WS = 64 * g.C;
  g.JW = (int)((cmax - 1 + WS - 1) / WS);
  if (g.JW < 1) g.JW = 1;
  g.B = (g.JW + g.P - 1) / g.P;
  g.G = (int)((N - 2 + CK_U - 1) / CK_U);
  g.NBK = (g.G + g.RB - 1) / g.RB;
  if (g.JW >= 65536 || g.NBK >= 65536) return g;
  g.R = CK_U * g.G + 2;
  g.EV = (uint64_t)3 + (uint64_t)g.G * CK_U + 136;
  g.NP = (uint64_t)g.G + 26;
  long nt = 0;
  for (int jw = 0; jw < g.JW; jw++) nt += g.NBK - ck_first_trip_h(2 + jw * WS) / g.RB;
  g.n_tiles = (unsigned)nt;
  size_t o = 256;
  g.off_cu = o;
  o += 4096 * sizeof(unsigned);
  g.off_prog = o;
  o += stb_align_up((size_t)D * g.JW * sizeof(unsigned), 256);
  g.off_ee = o;
  o += stb_align_up((size_t)D * (g.JW + 1) * g.NP * 8, 256);
  g.off_ev = o;
  o += stb_align_up((size_t)D * (g.JW + 1) * g.EV * 8, 256);
  g.off_cke = o;
  o += stb_align_up((size_t)D * g.JW * g.NBK * 64 * sizeof(unsigned), 256);
  g.off_ckv = o;
  o += stb_align_up((size_t)D * g.JW * g.NBK * 64 * g.C * 8, 256);
  g.zero_bytes = o;
  g.off_tm = o;
  o += stb_align_up((size_t)D * (g.R + 1) * sizeof(double), 256);
  g.off_te = o;
  o += stb_align_up((size_t)D * (g.R + 1) * sizeof(int), 256);
  g.bytes = o;
  g.ok = true;
  return g;
}

bool stb_ck_eligible(unsigned N, unsigned M, int D) { return ck_geometry(N, M, D).ok; }

int stb_ck_tuning(unsigned N, unsigned M, int D, int *W_out, int *rows_out) {
  const ck_geom g = ck_geometry(N, M, D);
  if (W_out) *W_out = 64 * g.C;
  if (rows_out) *rows_out = g.RB * CK_U;
  return g.ok ? 0 : 1;
}

size_t stb_ck_workspace(unsigned N, unsigned M, int D) {
  size_t need = 0;
  // (the strip shape is a tunable and depends on the batch: size for every shape)
  static const int shapes[3] = {1, 2, 4};
  const ck_geom g0 = ck_geometry(N, M, D);
  if (!g0.ok) return 0;
  need = g0.bytes;
  // ... and for the shortest blocks a tunable can ask for (one period: the most checkpoints)
  const size_t nbk_max = ((size_t)g0.G + g0.TP - 1) / g0.TP + 1;
  for (int c : shapes) {
    // the same sizes with C forced: edge streams scale with 1/C, checkpoints do not
    const int WS = 64 * c;
    const unsigned cmax = (M < N - 1) ? M : N - 1;
    const size_t JW = (cmax - 1 + WS - 1) / WS + 1;
    const size_t b = 256 + 4096 * 5 + (size_t)D * (JW + 1) * (g0.NP + g0.EV) * 8 + (size_t)D * JW * nbk_max * 64 * (4 + 8 * c) +
                     (size_t)D * (g0.R + 1) * 12 + 4096 * 8;
    if (b > need) need = b;
  }
  return need + 256;
}

// the order in which the tiles of a table become ready, as jw | b << 16: strip jw finishes block b at
// about (b + 1) RB 8 r + (jw / P) L + (jw % P) lag
struct ck_order_entry {
  int dev;
  unsigned N, M;
  int C, P, RB, NBK, JW;
  int r_ns, L_ns, lag_ns;
  unsigned n;
  unsigned *d_order;
};
static std::mutex g_order_mu;
static std::vector<ck_order_entry> g_orders;

static int ck_order_list(const ck_geom &g, unsigned N, unsigned M, const unsigned **out) {
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  const int r_ns = stb_env_int("STB_CK_ORDER_R", 30), L_ns = stb_env_int("STB_CK_ORDER_L", 2700),
            lag_ns = stb_env_int("STB_CK_ORDER_LAG", 400);
  std::lock_guard<std::mutex> lock(g_order_mu);
  for (const ck_order_entry &e : g_orders)
    if (e.dev == dev && e.N == N && e.M == M && e.C == g.C && e.P == g.P && e.RB == g.RB && e.NBK == g.NBK && e.JW == g.JW &&
        e.r_ns == r_ns && e.L_ns == L_ns && e.lag_ns == lag_ns) {
      *out = e.d_order;
      return 0;
    }
  struct item {
    long key;
    unsigned code;
  };
  std::vector<item> v;
  v.reserve(g.n_tiles);
  const int WS = 64 * g.C;
  for (int jw = 0; jw < g.JW; jw++) {
    const int b0 = ck_first_trip_h(2 + jw * WS) / g.RB;
    for (int b = b0; b < g.NBK; b++) {
      item it;
      it.key = (long)(b + 1) * g.RB * CK_U * r_ns + (long)(jw / g.P) * L_ns + (long)(jw % g.P) * lag_ns;
      it.code = (unsigned)jw | ((unsigned)b << 16);
      v.push_back(it);
    }
  }
  std::stable_sort(v.begin(), v.end(), [](const item &x, const item &y) { return x.key < y.key; });
  if (v.size() != g.n_tiles) return stb_fail("stb_fill_S: tile count %zu != %u", v.size(), g.n_tiles);
  std::vector<unsigned> codes(v.size());
  for (size_t i = 0; i < v.size(); i++) codes[i] = v[i].code;
  ck_order_entry e;
  e.dev = dev;
  e.N = N;
  e.M = M;
  e.C = g.C;
  e.P = g.P;
  e.RB = g.RB;
  e.NBK = g.NBK;
  e.JW = g.JW;
  e.r_ns = r_ns;
  e.L_ns = L_ns;
  e.lag_ns = lag_ns;
  e.n = g.n_tiles;
  e.d_order = nullptr;
  HIPCHK(hipMalloc((void **)&e.d_order, codes.size() * sizeof(unsigned) + 16));
  HIPCHK(hipMemcpy(e.d_order, codes.data(), codes.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  if (g_orders.size() >= 16) {  // (shapes come and go in tests: keep the table small)
    (void)hipFree(g_orders.front().d_order);
    g_orders.erase(g_orders.begin());
  }
  g_orders.push_back(e);
  *out = e.d_order;
  return 0;
}

// tiles per table of the summing form (the partial sums a caller has to provide room for: D times as many)
unsigned stb_ck_dot_parts(unsigned N, unsigned M, int D) {
  const ck_geom g = ck_geometry(N, M, D, true);
  return g.ok ? g.n_tiles : 0;
}
// spine workgroups the summing form would launch for D tables (it pays while they all fit on the chip)
unsigned stb_ck_dot_spine(unsigned N, unsigned M, int D) {
  const ck_geom g = ck_geometry(N, M, D, true);
  return g.ok ? (unsigned)g.B * (unsigned)D : 0xffffffffu;
}

int stb_launch_ck(fill_args &A, int D, char *ws, size_t ws_left, const dot_request *dot, unsigned **hdr_out, hipStream_t st) {
  const unsigned N = A.N, M = A.M;
  const ck_geom g = ck_geometry(N, M, D, dot != nullptr);
  if (!g.ok) return stb_fail("stb_fill_S: the checkpointed form does not take N=%u M=%u D=%d", N, M, D);
  if (g.bytes > ws_left) return stb_fail("stb_fill_S: workspace too small for the checkpointed form");
  ck_args X;
  memset(&X, 0, sizeof(X));
  X.hdr = (unsigned *)ws;
  X.progress = (unsigned *)(ws + g.off_prog);
  {
    // compute units with a spine wave on them take no tile work while it walks, unless the spine needs
    // so many of them that too few would be left for the tiles
    int dev0 = 0, cus0 = 256;
    HIPCHK(hipGetDevice(&dev0));
    HIPCHK(hipDeviceGetAttribute(&cus0, hipDeviceAttributeMultiprocessorCount, dev0));
    const int excl = stb_env_int("STB_CK_EXCL", -1);
    const bool on = (excl < 0) ? ((unsigned)g.B * (unsigned)D * 10u <= (unsigned)cus0 * 4u) : (excl != 0);
    X.cu_busy = on ? (unsigned *)(ws + g.off_cu) : nullptr;
  }
  X.edge_e = (unsigned long long *)(ws + g.off_ee);
  X.edge_v = (unsigned long long *)(ws + g.off_ev);
  X.ck_e = (unsigned *)(ws + g.off_cke);
  X.ck_v = (unsigned long long *)(ws + g.off_ckv);
  X.EV = g.EV;
  X.NP = g.NP;
  X.D = D;
  X.B = g.B;
  X.JW = g.JW;
  X.NBK = g.NBK;
  X.TP = g.TP;
  X.RB = g.RB;
  X.G = g.G;
  X.n_tiles = g.n_tiles;
  X.n_spine = (unsigned)g.B * (unsigned)D;
  X.timeout = (unsigned long long)stb_env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;  // wall_clock64: 100 MHz
  X.poll_nap = stb_env_int("STB_CHAIN_POLL_NAP", 2);
  X.spare_work = stb_env_int("STB_CK_SPARE", 0);
  X.diag = stb_env_int("STB_CK_DIAG", 0);
  if (dot) {
    if (!dot->item_ptr || dot->col0 != 2)
      return stb_fail("stb_fill_S: the checkpointed form sums over cell lists built for strips that start at column 2");
    X.item_ptr = dot->item_ptr;
    X.ent_pos = dot->ent_pos;
    X.ent_cnt = dot->ent_cnt;
    X.nsg = dot->nsg;
    X.dotp = dot->dotp;
    const_cast<dot_request *>(dot)->parts_per_table = (int)g.n_tiles;
  }
  X.pub_nap = stb_env_int("STB_CK_PUB_NAP", 4);
  X.tp_magic = (unsigned)((0x100000000ull + (unsigned)X.TP - 1) / (unsigned)X.TP);
  if (ck_order_list(g, N, M, &X.order)) return 1;
  const char *tl_file = getenv("STB_CK_TIMELINE");
  const size_t dbg_words = (size_t)g.JW * (g.NBK + 2) + (size_t)g.n_tiles * 4;
  if (tl_file && *tl_file) {
    HIPCHK(hipMalloc((void **)&X.dbg, dbg_words * 8));
    HIPCHK(hipMemsetAsync(X.dbg, 0, dbg_words * 8, st));
  }
  HIPCHK(hipMemsetAsync(ws, 0, g.zero_bytes, st));
  *hdr_out = X.hdr;
  stb_launch_s1(A, D, st);
  hipLaunchKernelGGL(k_col1, dim3(D), dim3(1024), 0, st, A.a, X, (int *)(ws + g.off_tm), g.R);
  // every workgroup is generic: the first B*D tickets walk the spine, the others work on tiles
  int dev = 0, cus = 256;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // one workgroup per compute unit: a second one slows the spine (clock, memory latency of its hand-offs)
  // by more than its tiles gain -- 8 tables: 0.98 ms against 1.17
  const int per_cu = stb_env_int("STB_CK_WG_PER_CU", 1);
  unsigned grid = (unsigned)(cus * per_cu);
  const unsigned min_workers = (unsigned)stb_env_int("STB_CK_MIN_WORKERS", 64);
  if (grid < X.n_spine + min_workers) grid = X.n_spine + min_workers;
  if (stb_env_int("STB_CK_GRID", 0) > 0) grid = (unsigned)stb_env_int("STB_CK_GRID", 0);  // (diagnostic: spine alone)
  const int shape = g.C * 10 + g.P;
  // dynamic LDS: the tile order (storing form), or the staging rows and the cell lists of a tile (summing form)
  const size_t shm_fill = (g.n_tiles <= CK_ORDER_LDS) ? (size_t)g.n_tiles * sizeof(unsigned) : 0;
  if (dot) {
    const size_t shm = (size_t)CK_U * CK_NW * 64 * g.C * sizeof(double) + (size_t)CK_NW * CK_MAXRB * 64 * (sizeof(unsigned) + sizeof(unsigned short));
#define CKD(CC, PP) STB_LAUNCH_SHM((k_fill_ck<CC, PP, 1>), dim3(grid), dim3(64 * CK_NW), shm, st, A, X)
    switch (shape) {
      case 14: CKD(1, 4); break;
      case 21: CKD(2, 1); break;
      case 22: CKD(2, 2); break;
      case 24: CKD(2, 4); break;
      default: return stb_fail("stb_fill_S: no summing checkpointed kernel for C=%d P=%d", g.C, g.P);
    }
#undef CKD
  } else {
#define CKL(CC, PP) STB_LAUNCH_SHM((k_fill_ck<CC, PP, 0>), dim3(grid), dim3(64 * CK_NW), shm_fill, st, A, X)
  switch (shape) {
    case 11: CKL(1, 1); break;
    case 12: CKL(1, 2); break;
    case 14: CKL(1, 4); break;
    case 21: CKL(2, 1); break;
    case 22: CKL(2, 2); break;
    case 24: CKL(2, 4); break;
    case 41: CKL(4, 1); break;
    case 42: CKL(4, 2); break;
    case 44: CKL(4, 4); break;
    default: return stb_fail("stb_fill_S: no checkpointed kernel for C=%d P=%d", g.C, g.P);
  }
#undef CKL
  }
  HIPCHK(hipGetLastError());
  if (X.dbg) {
    // spine: one line per wave strip "S jw start b1 b2 ... end"; workers: "W jw b claimed loaded done hwid"
    HIPCHK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(dbg_words);
    std::vector<unsigned> ord(g.n_tiles);
    HIPCHK(hipMemcpy(h.data(), X.dbg, dbg_words * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ord.data(), X.order, g.n_tiles * 4, hipMemcpyDeviceToHost));
    FILE *f = fopen(tl_file, "w");
    if (f) {
      fprintf(f, "G %d %d %d %d %d %d %d %d\n", g.C, g.P, g.JW, g.NBK, g.RB, g.TP, g.G, D);
      for (int jw = 0; jw < g.JW; jw++) {
        fprintf(f, "S %d", jw);
        for (int b = 0; b < g.NBK + 2; b++) fprintf(f, " %llu", h[(size_t)jw * (g.NBK + 2) + b]);
        fprintf(f, "\n");
      }
      const unsigned long long *w = h.data() + (size_t)g.JW * (g.NBK + 2);
      for (unsigned k = 0; k < g.n_tiles; k++)
        fprintf(f, "W %u %u %llu %llu %llu %llu\n", ord[k] & 0xffffu, ord[k] >> 16, w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
      fclose(f);
    }
    (void)hipFree(X.dbg);
  }
  return 0;
}

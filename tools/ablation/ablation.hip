// ablation.hip -- the fill forms the chain and producer/consumer forms superseded, kept for
// A/B measurements and as independent cross-checks; compiled only by `make ABLATION=1`.
//
//   STB_FILL_FUSED        k_fill_bfp     recurrence and log in one wave, launched per row block
//   STB_FILL_SPLIT        k_rec + k_logconv   recurrence kernel, in-place log conversion on side streams
//   STB_FILL_SCALED_STEP  k_fill_rows<SCALED> (mantissa, exponent) cells renormalised every row, ocml log
//   STB_FILL_CHAINX       k_fill_chainx  the chain alone in its blocks, logs by converter blocks (D <= 2)
// All compute S_remake_part's table (reference lib/stable.c:321-388).

#include <type_traits>

#include "fill_chain.h"
#include "fill_rows.h"

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
  lt[lane] = A.lt[lane];
  lt[lane + 64] = A.lt[lane + 64];
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
};


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
}

// rows [ra, rb] of every table: raw significands -> logs, in place.  grid = (column chunks of 512,
// rb-ra+1 rows, D tables); a thread converts two adjacent columns (one 16-byte load and store).
__global__ __launch_bounds__(256) void k_logconv(fill_args A, split_args X, int ra, int rb) {
  __shared__ double2 lt[128];
  if (threadIdx.x < 128) lt[threadIdx.x] = A.lt[threadIdx.x];
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
  if (tid < 128) lt[tid] = A.lt[tid];
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
}

// ------------------------------------------------------------------------------------------------
// host side

#define STB_PPL_MAX 8  // renormalisation periods per launch of the split form

struct chainx_geom {
  int P, B, G, Q, NPer;
  uint64_t EV, NP, EWh;
  size_t prog_bytes, zero_bytes, bytes;  // progress words; what is zeroed per fill; everything
};
static chainx_geom chainx_geometry(unsigned N, unsigned M, int D) {
  chainx_geom g;
  g.P = stb_env_int("STB_CHAINX_P", 4);
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
  g.prog_bytes = stb_align_up((size_t)D * g.B * g.P * 32 * sizeof(unsigned), 256);  // one 128-byte line per word
  g.zero_bytes = 256 + g.prog_bytes + (size_t)D * g.B * (g.EV + g.NP) * sizeof(unsigned long long);
  g.bytes = stb_align_up(g.zero_bytes, 256) + 2048 + (size_t)D * g.NPer * g.EWh * sizeof(int);
  return g;
}

extern "C" size_t stb_ablation_workspace(unsigned N, unsigned M, int D) {
  const size_t W = stb_align_up((size_t)M + 2, 64);
  size_t need = (size_t)D * STB_EP_RING * STB_PPL_MAX * W * sizeof(int);  // exponent ring of the split form
  if (N >= 3 && M >= 2 && D >= 1 && D <= 2) {
    const size_t cx = chainx_geometry(N, M, D).bytes + 256;
    if (cx > need) need = cx;
  }
  return need;
}

// auxiliary streams and an event pool for the split variant (per host thread and device)
struct split_ctx {
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[1024];
  int made = 0;
};
static thread_local split_ctx g_split[16];

static int split_event(split_ctx &c, int i, hipEvent_t *out) {
  if (i >= 1024) return stb_fail("split fill: too many row groups");
  while (c.made <= i) {
    HIPCHK(hipEventCreateWithFlags(&c.ev[c.made], hipEventDisableTiming));
    c.made++;
  }
  *out = c.ev[i];
  return 0;
}

static int equal_periods(int R, int p) {  // equal-length renormalisation periods inside a launch
  if (p >= R) return R;
  const int per = (R + p - 1) / p;
  return (R + per - 1) / per;
}

static int launch_chainx(fill_args &A, int D, char *ws, size_t ws_left, unsigned **hdr_out, hipStream_t st) {
  const unsigned N = A.N, M = A.M;
  const chainx_geom cg = chainx_geometry(N, M, D);
  int Pc = stb_period_rows(N);
  const int Penv = stb_env_int("STB_FILL_P", 0);
  if (Penv > 0 && Penv < Pc) Pc = Penv;
  chain_args X;
  chainx_args Y;
  memset(&X, 0, sizeof(X));
  X.TP = Pc / CH_U;
  if (X.TP < 1) return stb_fail("stb_fill_S: renormalisation period %d shorter than a trip", Pc);
  X.tp_magic = (unsigned)((0x100000000ull + (unsigned)X.TP - 1) / (unsigned)X.TP);
  X.G = cg.G;
  X.D = D;
  X.B = cg.B;
  X.EV = cg.EV;
  X.NP = cg.NP;
  if (cg.bytes > ws_left) return stb_fail("stb_fill_S: workspace too small for the chain form");
  X.hdr = (unsigned *)ws;
  Y.progress = (unsigned *)(ws + 256);
  X.edge_e = (unsigned long long *)(ws + 256 + cg.prog_bytes);
  X.edge_v = X.edge_e + (size_t)D * cg.B * X.NP;
  char *tail = ws + stb_align_up(cg.zero_bytes, 256);
  Y.dump = (unsigned long long *)tail;
  Y.expo = (int *)(tail + 2048);
  Y.EWh = cg.EWh;
  Y.NPer = cg.NPer;
  Y.Q = cg.Q;
  X.timeout = (unsigned long long)stb_env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;  // 100 MHz ticks
  HIPCHK(hipMemsetAsync(ws, 0, stb_align_up(cg.zero_bytes, 16), st));
  *hdr_out = X.hdr;
  stb_launch_s1(A, D, st);
  const dim3 grid(((unsigned)cg.B + (unsigned)cg.Q) * (unsigned)D);
  if (cg.P == 1) STB_LAUNCH((k_fill_chainx<1>), grid, dim3(512), st, A, X, Y);
  else if (cg.P == 2) STB_LAUNCH((k_fill_chainx<2>), grid, dim3(512), st, A, X, Y);
  else STB_LAUNCH((k_fill_chainx<4>), grid, dim3(512), st, A, X, Y);
  HIPCHK(hipGetLastError());
  return 0;
}

static int launch_split(fill_args &A, int D, int C, int P, char *ws, hipStream_t st) {
  const int N = (int)A.N, M = (int)A.M, R = A.R;
  split_args X;
  X.P = P;
  X.PPL = (R + P - 1) / P;
  if (X.PPL > STB_PPL_MAX)
    return stb_fail("stb_fill_S: %d renormalisation periods per launch (max %d); lower STB_FILL_R", X.PPL, STB_PPL_MAX);
  X.epbuf = (int *)ws;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return stb_fail("device index %d out of range", dev);
  split_ctx &cx = g_split[dev];
  for (auto &q : cx.aux)
    if (!q) HIPCHK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
  int G = stb_env_int("STB_SPLIT_GROUP", 8);  // row-blocks converted per k_logconv launch
  if (G < 1) G = 1;
  if (G > STB_EP_RING / 2) G = STB_EP_RING / 2;
  const int nlaunch = (N - 1 + R - 1) / R;
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
    if (n1 > N) n1 = N;
    int ncols = (n1 < M ? n1 : M) - 1;
    if (ncols < 1) ncols = 1;
    const dim3 grid((ncols + A.Wv - 1) / A.Wv, D);
    switch (C) {
      case 1: STB_LAUNCH((k_rec<1>), grid, dim3(64), st, A, X, k); break;
      case 2: STB_LAUNCH((k_rec<2>), grid, dim3(64), st, A, X, k); break;
      default: STB_LAUNCH((k_rec<4>), grid, dim3(64), st, A, X, k); break;
    }
    if (k % G == G - 1 || k == nlaunch - 1) {
      hipEvent_t ea, eb;
      if (split_event(cx, g, &ea) || split_event(cx, ngroups + g, &eb)) return 1;
      hipStream_t q = cx.aux[g % 3];
      HIPCHK(hipEventRecord(ea, st));
      HIPCHK(hipStreamWaitEvent(q, ea, 0));
      const int ra = 2 + g * G * R;
      const int rb = n1;
      const int cm = (rb < M ? rb : M) - 1;  // columns 2..min(rb,M)
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
  return 0;
}

extern "C" int stb_ablation_fill(fill_args &A, int D, int variant, char *ws, size_t ws_left, unsigned **hdr_out,
                                 hipStream_t st) {
  const unsigned N = A.N, M = A.M;
  *hdr_out = nullptr;
  if (variant == STB_FILL_CHAINX) return launch_chainx(A, D, ws, ws_left, hdr_out, st);
  const bool split = variant == STB_FILL_SPLIT;
  const bool few = (uint64_t)D * M < 40000;
  const int C = stb_env_int("STB_FILL_C", split ? 2 : (few ? 1 : 2));
  if (C != 1 && C != 2 && C != 4) return stb_fail("STB_FILL_C must be 1, 2 or 4");
  A.R = stb_env_int("STB_FILL_R", split ? 96 : (few ? 48 : 64));
  if (A.R < 1) A.R = 1;
  A.H = (A.R + C - 1) / C * C;
  if (A.H > 64 * C - C) return stb_fail("STB_FILL_R=%d too large for C=%d", A.R, C);
  A.Wv = 64 * C - A.H;
  // rows per renormalisation period: these forms start a period at 2^-900 with 1700 bits of head-room
  int bits = 1;
  while ((1ull << bits) < (unsigned long long)N) bits++;
  int P = 1700 / (2 * bits + 1);
  const int Penv = stb_env_int("STB_FILL_P", 0);
  if (Penv > 0 && Penv < P) P = Penv;
  if (P < 1) P = 1;
  P = equal_periods(A.R, P);
  if (split) {
    if ((size_t)D * STB_EP_RING * STB_PPL_MAX * A.W * sizeof(int) > ws_left)
      return stb_fail("stb_fill_S: workspace too small for the split form");
    return launch_split(A, D, C, P, ws, st);
  }
  const int Nn = (int)N, Mm = (int)M, R = A.R;
  const int nlaunch = (Nn - 1 + R - 1) / R;
  for (int k = 0; k < nlaunch; k++) {
    int n1 = 2 + (k + 1) * R - 1;
    if (n1 > Nn) n1 = Nn;
    int ncols = (n1 < Mm ? n1 : Mm) - 1;
    if (ncols < 1) ncols = 1;
    const dim3 grid((ncols + A.Wv - 1) / A.Wv, D);
    if (variant == STB_FILL_SCALED_STEP) {
      switch (C) {
        case 1: STB_LAUNCH((k_fill_rows<1, STB_MODE_SCALED>), grid, dim3(64), st, A, k); break;
        case 2: STB_LAUNCH((k_fill_rows<2, STB_MODE_SCALED>), grid, dim3(64), st, A, k); break;
        default: STB_LAUNCH((k_fill_rows<4, STB_MODE_SCALED>), grid, dim3(64), st, A, k); break;
      }
    } else {  // STB_FILL_FUSED
      switch (C) {
        case 1: STB_LAUNCH((k_fill_bfp<1>), grid, dim3(64), st, A, k, P); break;
        case 2: STB_LAUNCH((k_fill_bfp<2>), grid, dim3(64), st, A, k, P); break;
        default: STB_LAUNCH((k_fill_bfp<4>), grid, dim3(64), st, A, k, P); break;
      }
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

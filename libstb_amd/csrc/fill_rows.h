// fill_rows.h -- k_fill_rows: the launch-per-row-block fill in the REFERENCE'S OWN operation order.
//
//   STB_MODE_LOGDOM  S table, logadd(log(n - m a - 1) + up, left) per cell with ocml log / exp and
//                    contraction off: lib/stable.c:95-103, :380-388 as written (the cross-check
//                    against the default linear-domain forms, and the fill for N >= 2^27)
//   STB_MODE_VRATIO  V table, lib/stable.c:475-480 (the form k_fillv_chain replaced; STB_FILLV_CHAIN=0)
//   STB_MODE_SCALED  S table, (mantissa, exponent) cells renormalised every row (ablation builds only)
//
// One wavefront per column strip, C adjacent columns per lane, left neighbour through one DPP wave
// shift per row; a launch advances R rows from the frontier (the previous launch's last row) and
// recomputes an R-column halo so that strips never talk to each other.
#ifndef STB_FILL_ROWS_H
#define STB_FILL_ROWS_H

#include "stb_common.h"

// A table value S (not its log) as mant * 2^expo, mant in [0.5,1), or exact zero (mant 0, expo EZ).
// The recurrence  S^n_m = (n-1-m a) S^{n-1}_m + S^{n-1}_{m-1}  is then one fma plus exponent
// bookkeeping -- no transcendental on the dependent chain, and every step rounds once at 2^-53
// relative, which is tighter than the reference's log-domain step (one rounding at ulp(log S)).

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

#endif

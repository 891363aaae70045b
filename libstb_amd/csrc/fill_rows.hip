// fill_rows.hip -- launch side of k_fill_rows (fill_rows.h) for the two forms kept in the default
// build: the S table in the reference's log-domain operation order (STB_FILL_LOGDOMAIN, and any
// table with N >= 2^27 rows), and the V table per row block (STB_FILLV_CHAIN=0).

#include "fill_rows.h"

template <int C>
static void launch_rows_c(const fill_args &A, int k, dim3 grid, int what, hipStream_t st) {
  if (what == STB_ROWS_VRATIO)
    STB_LAUNCH((k_fill_rows<C, STB_MODE_VRATIO>), grid, dim3(64), st, A, k);
  else
    STB_LAUNCH((k_fill_rows<C, STB_MODE_LOGDOM>), grid, dim3(64), st, A, k);
}

// rows 2..N in launches of A.R rows; strips of A.Wv owned columns (A.R, A.H, A.Wv set by the caller)
int stb_launch_rows(fill_args &A, int D, int C, int what, hipStream_t st) {
  const int N = (int)A.N, M = (int)A.M, R = A.R;
  const int nlaunch = (N - 1 + R - 1) / R;
  for (int k = 0; k < nlaunch; k++) {
    int n1 = 2 + (k + 1) * R - 1;
    if (n1 > N) n1 = N;
    int ncols = (n1 < M ? n1 : M) - 1;  // owned columns 2..min(n1,M)
    if (ncols < 1) ncols = 1;
    const dim3 grid((ncols + A.Wv - 1) / A.Wv, D);
    switch (C) {
      case 1: launch_rows_c<1>(A, k, grid, what, st); break;
      case 2: launch_rows_c<2>(A, k, grid, what, st); break;
      default: launch_rows_c<4>(A, k, grid, what, st); break;
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

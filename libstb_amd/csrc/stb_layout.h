/*
 * stb_layout.h -- where a log-Stirling table lives in memory (HBM and the host mirror).
 *
 * The reference keeps `double **S` with S[n-3][m-2] = log S^n_{m,a} for 3<=n<=usedN,
 * 2<=m<=min(n-1,usedM), one malloc per row (reference lib/stable.h:77, lib/stable.c:207-231).
 * Here a table is ONE slab: rows back to back, row n holding min(n-2, M-1) values followed by
 * alignment padding and a fixed slack (see below).  The host mirror is a
 * byte-for-byte copy of the device slab (one hipMemcpy), and the row-pointer vector the
 * reference's struct exposes simply points into it.
 *
 * Usable from C, C++ and HIP device code.
 */
#ifndef STB_LAYOUT_H
#define STB_LAYOUT_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define STB_HD __host__ __device__ static inline
#else
#define STB_HD static inline
#endif

/* Row geometry.  Row n of an S table stores m = 2..min(n-1,M): len(n) = min(n-2, M-1) values.
 * In the slab each row occupies pitch(n) = roundup(len(n), STB_ROW_ALIGN) + STB_ROW_SLACK elements:
 *  - every row base is 512-byte aligned, so a wave's 512 B / 1 KiB store never straddles lines;
 *  - at least STB_ROW_SLACK elements after the last stored value belong to the same row and are
 *    never read.  The fill kernels rely on this: a wavefront that straddles the diagonal (or column
 *    M) stores all of its 64*C columns unconditionally, the excess lands in the slack (C <= 4).
 * N = M = 10000: 52.9 M elements (423 MB) for 49 985 001 stored values. */
#define STB_ROW_ALIGN 64u
#define STB_ROW_SLACK 256u

STB_HD unsigned stb_row_len(unsigned n, unsigned M) {
  if (n < 3) return 0;
  return (n - 2 < M - 1) ? n - 2 : M - 1;
}

STB_HD unsigned stb_row_pitch(unsigned n, unsigned M) {
  if (n < 3) return 0;
  return ((stb_row_len(n, M) + STB_ROW_ALIGN - 1) & ~(STB_ROW_ALIGN - 1)) + STB_ROW_SLACK;
}

/* sum_{L=1..k} (roundup(L, ALIGN) + SLACK) */
STB_HD uint64_t stb_tri_padded(uint64_t k) {
  const uint64_t q = k / STB_ROW_ALIGN, r = k % STB_ROW_ALIGN;
  return STB_ROW_ALIGN * (STB_ROW_ALIGN * (q * (q + 1) / 2) + r * (q + 1)) + k * STB_ROW_SLACK;
}

/* element offset of row n (3<=n) inside the slab */
STB_HD uint64_t stb_row_offset(unsigned n, unsigned M) {
  if (n <= 3) return 0;
  if (n <= M + 1) return stb_tri_padded((uint64_t)n - 3);
  return stb_tri_padded((uint64_t)M - 1) + (uint64_t)(n - M - 2) * stb_row_pitch(M + 1, M);
}

/* slab size in elements for bounds (N,M) */
STB_HD uint64_t stb_table_elems(unsigned N, unsigned M) { return stb_row_offset(N + 1, M); }

/* algorithmic cell count: sum_{n=3..N} min(n-2,M-1)  (cells(N,M)) */
STB_HD uint64_t stb_table_cells(unsigned N, unsigned M) {
  uint64_t k;
  if (N < 3) return 0;
  if (N <= M + 1) {
    k = (uint64_t)N - 2;
    return k * (k + 1) / 2;
  }
  return ((uint64_t)M - 1) * M / 2 + (uint64_t)(N - M - 1) * ((uint64_t)M - 1);
}

/* V table (reference lib/stable.h:85): V[n-2][m-2] = V^n_{m,a}, 2<=n<=N, 2<=m<=min(n,M) */
STB_HD unsigned stb_vrow_len(unsigned n, unsigned M) {
  if (n < 2) return 0;
  return (n - 1 < M - 1) ? n - 1 : M - 1;
}
STB_HD unsigned stb_vrow_pitch(unsigned n, unsigned M) {
  if (n < 2) return 0;
  return ((stb_vrow_len(n, M) + STB_ROW_ALIGN - 1) & ~(STB_ROW_ALIGN - 1)) + STB_ROW_SLACK;
}
STB_HD uint64_t stb_vrow_offset(unsigned n, unsigned M) {
  if (n <= 2) return 0;
  if (n <= M) return stb_tri_padded((uint64_t)n - 2);
  return stb_tri_padded((uint64_t)M - 1) + (uint64_t)(n - M - 1) * stb_vrow_pitch(M, M);
}
STB_HD uint64_t stb_vtable_elems(unsigned N, unsigned M) { return stb_vrow_offset(N + 1, M); }
STB_HD uint64_t stb_vtable_cells(unsigned N, unsigned M) {
  uint64_t k;
  if (N < 2) return 0;
  if (N <= M) {
    k = (uint64_t)N - 1;
    return k * (k + 1) / 2;
  }
  return ((uint64_t)M - 1) * M / 2 + (uint64_t)(N - M) * ((uint64_t)M - 1);
}

#endif

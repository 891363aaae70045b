/*
 * oracle/stb_oracle.h -- TEST INFRASTRUCTURE ONLY.  NOT part of the product.
 *
 * Scalar CPU restatement of the arithmetic on libstb's Stirling-table / hyper-parameter
 * posterior path, written from the maths, citing the reference line each
 * function follows.  It is the checker the HIP path is compared against and the "port" CPU
 * baseline bench.py times.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; the product library (libstb_amd) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every function below bit-for-bit
 * against the real reference compiled into oracle/_ref/libstb_ref.so (same libm), and
 * tests/test_oracle_golden.py checks it against the committed fixtures in tests/golden/.
 *
 * Table layout used here (ours, not the reference's row-pointer vectors): one packed array,
 * row n (3<=n<=N) holds log S^n_{m,a} for m=2..min(n-1,M), rows back to back.
 */
#ifndef STB_ORACLE_H
#define STB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* number of stored cells in rows 3..N with column bound M (cells(N,M)) */
uint64_t orc_cells(unsigned N, unsigned M);
/* offset (in doubles) of row n in the packed array; length of row n */
uint64_t orc_row_offset(unsigned n, unsigned M);
unsigned orc_row_len(unsigned n, unsigned M);

/* lib/stable.c:95-103 */
double orc_logadd(double V, double lp);

/* lib/stable.c:321-388 (double S branch, startN==0): fills S1[0..N) and the packed table */
void orc_fill_S(double a, unsigned N, unsigned M, double *S1, double *table);
/* lib/stable.c:451-482 (double V branch, startN==0): V rows n=2..N, entries m=2..min(n,M),
 * packed back to back (row n has min(n-1,M-1) entries) */
void orc_fill_V(double a, unsigned N, unsigned M, double *vtable);
uint64_t orc_vcells(unsigned N, unsigned M);
uint64_t orc_vrow_offset(unsigned n, unsigned M);

/* lib/stable.c:941-974 restricted to an already-filled table of bounds (N,M); n>N or m>M
 * (no growth here) returns -HUGE_VAL.  S1 has N entries. */
double orc_S_S(const double *table, const double *S1, unsigned N, unsigned M, unsigned n,
               unsigned m);
/* lib/stable.c:875-898, :900-939 on a filled V table of bounds (N,M) (no growth) */
double orc_S_V(const double *vtable, unsigned N, unsigned M, unsigned n, unsigned m);
double orc_S_U(const double *vtable, double a, unsigned N, unsigned M, unsigned n, unsigned m);
double orc_S_UV(const double *vtable, double a, unsigned N, unsigned M, unsigned n, unsigned m);

/* lib/stable.c:1057-1084 */
double orc_S_asympt(double a, unsigned n, unsigned m);

/* lib/stable.c:564-630: growth policy of S_extend (integers only). in: current used/max bounds and
 * the (N,M) that S_S/S_V passes (already +1); out: new usedN/usedM */
void orc_extend_policy(unsigned usedN, unsigned usedM, unsigned maxN, unsigned maxM, int N, int M,
                       unsigned *newN, unsigned *newM);
/* lib/stable.c:118-129: argument clamps of S_make (including the :126-127 quirk) */
void orc_make_clamp(unsigned *initN, unsigned *initM, unsigned *maxN, unsigned *maxM);

/* lib/samplea.c:46-83 minus the table build: restaurant terms + sequential gather-sum over a
 * table already filled for discount x with bounds (N,M).  Flat CSR groups: restaurant i owns
 * pairs [koff[i], koff[i+1]) */
double orc_aterms_sum(double x, int I, const int *K, const uint32_t *T, const uint32_t *nflat,
                      const uint16_t *tflat, const double *bpar, const double *table,
                      const double *S1, unsigned N, unsigned M);
/* full aterms: fill (scratch must hold orc_cells(N,M)+N doubles) then orc_aterms_sum */
double orc_aterms(double x, int I, const int *K, const uint32_t *T, const uint32_t *nflat,
                  const uint16_t *tflat, const double *bpar, unsigned N, unsigned M,
                  double *scratch);
/* lib/samplea.c:186-208: maxn = max n + 1, maxt = max t + 1 (both start at 1) */
void orc_scan_bounds(int I, const int *K, const uint32_t *nflat, const uint16_t *tflat, int *maxn,
                     int *maxt);

/* lib/sampleb.c:33-41 */
double orc_bterms(double x, double Q, double shape, int I, const uint32_t *T, double apar);
/* lib/samplea.c:85-150 (aterms2, LGCACHE form) for a given partition m in samplea2's layout */
double orc_aterms2(double x, int I, const int *K, const uint32_t *T, const uint32_t *nflat, const uint16_t *tflat,
                   const double *bpar, const uint16_t *m);
/* lib/samplea.c:295-320: sample the partition from the table for discount a and one uniform per pair */
size_t orc_partition(double a, const double *table, const double *S1, unsigned N, unsigned M, int I, const int *K,
                     const uint32_t *nflat, const uint16_t *tflat, const double *u, uint16_t *m);

/* lib/sapprox.c:28-71 (LS_NOPOLYGAMMA build: lib/digamma.h:25) */
double orc_S_approx(int n, int m, float a);

/* wall-clock helper for bench.py's cpu_baseline: seconds for `reps` fills, best-of */
/* rows[0..k) (ascending, >= 3) of a table too large to keep: out[i * M + m - 2], the loop of orc_fill_S over two row
 * buffers with the columns shared among `threads` threads; 0 on success */
int orc_rows_stream(double a, unsigned N, unsigned M, const unsigned *rows, int k, double *out, int threads);
double orc_time_fill(double a, unsigned N, unsigned M, int reps, double *S1, double *table);
/* fills rows 3..N but only times/returns after `rows` rows (bounded sample); returns seconds and
 * writes the number of cells produced */
double orc_time_fill_rows(double a, unsigned N, unsigned M, unsigned rows, double *S1,
                          double *table, uint64_t *cells_done);
/* threaded batch: D tables, one per thread (threads<=D), returns seconds; tables may be NULL ->
 * allocated and freed inside */
double orc_time_fill_batch(const double *a, int D, unsigned N, unsigned M, int threads);

#ifdef __cplusplus
}
#endif
#endif

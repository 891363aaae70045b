/*
 * oracle/stb_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT part of the product.
 *
 * CPU restatement (plain scalar C) of the arithmetic on libstb's S-table / hyper-parameter
 * posterior path.  See stb_oracle.h for the contract and the parity status (PINNED against the
 * compiled reference in oracle/_ref and the fixtures in tests/golden).
 *
 * Every floating-point expression keeps the association order of the reference line it cites,
 * because the recurrence is bit-reproducible only under that order; the file is built with
 * -ffp-contract=off for the same reason.
 */
#define _GNU_SOURCE
#include "stb_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ layout */

unsigned orc_row_len(unsigned n, unsigned M) {
  /* row n stores m=2..min(n-1,M) */
  if (n < 3) return 0;
  return (n - 2 < M - 1) ? n - 2 : M - 1;
}

uint64_t orc_row_offset(unsigned n, unsigned M) {
  /* rows 3..M+1 form a triangle (lengths 1..M-1), later rows all have M-1 entries */
  uint64_t k;
  if (n <= 3) return 0;
  if (n <= M + 1) {
    k = n - 3; /* rows before n */
    return k * (k + 1) / 2;
  }
  k = (uint64_t)(M - 1) * M / 2;
  return k + (uint64_t)(n - M - 2) * (M - 1);
}

uint64_t orc_cells(unsigned N, unsigned M) { return orc_row_offset(N + 1, M); }

/* V rows n=2..N store m=2..min(n,M): length min(n-1,M-1) */
uint64_t orc_vrow_offset(unsigned n, unsigned M) {
  uint64_t k;
  if (n <= 2) return 0;
  if (n <= M) {
    k = n - 2;
    return k * (k + 1) / 2;
  }
  k = (uint64_t)(M - 1) * M / 2;
  return k + (uint64_t)(n - M - 1) * (M - 1);
}
uint64_t orc_vcells(unsigned N, unsigned M) { return orc_vrow_offset(N + 1, M); }

/* ------------------------------------------------------------------ fill */

/* lib/stable.c:95-103: larger + log(1.0+exp(smaller-larger)); log(1.0+.) on purpose, not log1p */
double orc_logadd(double V, double lp) {
  double hi = V, lo = lp;
  if (lp > V) {
    hi = lp;
    lo = V;
  }
  return hi + log(1.0 + exp(lo - hi));
}

void orc_fill_S(double a, unsigned N, unsigned M, double *S1, double *table) {
  unsigned n, m;
  const double *prev;
  double *cur;
  /* lib/stable.c:337-348: S1 is a running sum of log(n-1-a), integer n-1 formed first */
  S1[0] = 0;
  for (n = 2; n <= N; n++) S1[n - 1] = S1[n - 2] + log((double)((int)n - 1) - a);
  if (N < 3) return;
  /* lib/stable.c:374: log S^3_2 */
  table[0] = orc_logadd(S1[1], log(2 - 2 * a));
  prev = table;
  for (n = 4; n <= N; n++) {
    unsigned last = (n - 1 < M) ? n - 1 : M; /* largest stored m in this row */
    cur = table + orc_row_offset(n, M);
    /* lib/stable.c:381-382: m=2 column fed by S1 */
    cur[0] = orc_logadd(log(((double)(int)n - 2 * a) - 1.0) + prev[0], S1[n - 2]);
    /* lib/stable.c:383-386 */
    for (m = 3; m <= last; m++) {
      double up = (m < n - 1) ? prev[m - 2] : 0.0; /* S^{n-1}_{n-1} := log 1 */
      cur[m - 2] =
          orc_logadd(log(((double)(int)n - (double)(int)m * a) - 1.0) + up, prev[m - 3]);
    }
    prev = cur;
  }
}

/* A few rows of a table too large to keep (N = M = 45 000 is 8 GB): the loop of orc_fill_S -- lib/stable.c:380-388, the
 * same operations in the same order, hence the same bits -- over two row buffers, the columns shared out among `threads`
 * threads that meet once per row (a cell needs the previous row only).  out[i * M + m - 2] = log S^{rows[i]}_m for
 * m = 2 .. min(rows[i] - 1, M); rows ascending, each >= 3.  Pinned against orc_fill_S in tests/test_oracle_golden.py. */
typedef struct {
  double a;
  unsigned N, M;
  const unsigned *rows;
  int k, threads, tid;
  double *out, *buf[2];
  const double *S1;
  pthread_barrier_t *bar;
} orc_rs_job;
static void *orc_rs_worker(void *vp) {
  orc_rs_job *J = (orc_rs_job *)vp;
  const double a = J->a;
  unsigned n, m;
  int next = 0;
  while (next < J->k && J->rows[next] < 3) next++;
  if (J->tid == 0) {
    J->buf[1][0] = orc_logadd(J->S1[1], log(2 - 2 * a)); /* row 3 lives in buf[3 & 1] */
    if (next < J->k && J->rows[next] == 3) J->out[(size_t)next * J->M] = J->buf[1][0];
  }
  if (next < J->k && J->rows[next] == 3) next++;
  pthread_barrier_wait(J->bar);
  for (n = 4; n <= J->N; n++) {
    const double *prev = J->buf[(n - 1) & 1];
    double *cur = J->buf[n & 1];
    const unsigned last = (n - 1 < J->M) ? n - 1 : J->M;
    /* columns 2 .. last in contiguous shares */
    const unsigned cnt = last - 1, lo = 2 + (unsigned)((uint64_t)cnt * (unsigned)J->tid / (unsigned)J->threads),
                   hi = 2 + (unsigned)((uint64_t)cnt * ((unsigned)J->tid + 1) / (unsigned)J->threads);
    for (m = lo; m < hi; m++) {
      if (m == 2) {
        cur[0] = orc_logadd(log(((double)(int)n - 2 * a) - 1.0) + prev[0], J->S1[n - 2]);
      } else {
        double up = (m < n - 1) ? prev[m - 2] : 0.0;
        cur[m - 2] = orc_logadd(log(((double)(int)n - (double)(int)m * a) - 1.0) + up, prev[m - 3]);
      }
    }
    if (next < J->k && J->rows[next] == n) {
      for (m = lo; m < hi; m++) J->out[(size_t)next * J->M + m - 2] = cur[m - 2];
      next++;
    }
    pthread_barrier_wait(J->bar);
  }
  return NULL;
}
int orc_rows_stream(double a, unsigned N, unsigned M, const unsigned *rows, int k, double *out, int threads) {
  pthread_barrier_t bar;
  pthread_t th[64];
  orc_rs_job J[64];
  double *S1, *b0, *b1;
  unsigned n;
  int t;
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  if (N < 3 || M < 2) return 1;
  S1 = (double *)malloc(sizeof(double) * N);
  b0 = (double *)calloc(M, sizeof(double));
  b1 = (double *)calloc(M, sizeof(double));
  if (!S1 || !b0 || !b1) {
    free(S1); free(b0); free(b1);
    return 1;
  }
  S1[0] = 0;
  for (n = 2; n <= N; n++) S1[n - 1] = S1[n - 2] + log((double)((int)n - 1) - a);
  pthread_barrier_init(&bar, NULL, (unsigned)threads);
  for (t = 0; t < threads; t++) {
    J[t].a = a; J[t].N = N; J[t].M = M; J[t].rows = rows; J[t].k = k; J[t].threads = threads; J[t].tid = t;
    J[t].out = out; J[t].buf[0] = b0; J[t].buf[1] = b1; J[t].S1 = S1; J[t].bar = &bar;
  }
  for (t = 1; t < threads; t++) pthread_create(&th[t], NULL, orc_rs_worker, &J[t]);
  orc_rs_worker(&J[0]);
  for (t = 1; t < threads; t++) pthread_join(th[t], NULL);
  pthread_barrier_destroy(&bar);
  free(S1); free(b0); free(b1);
  return 0;
}

void orc_fill_V(double a, unsigned N, unsigned M, double *v) {
  unsigned n, m;
  const double *prev;
  double *cur;
  if (N < 2) return;
  /* lib/stable.c:468-469: V^2_2 */
  v[0] = 1.0 / (1.0 - a);
  prev = v;
  for (n = 3; n <= N; n++) {
    unsigned last = (n < M) ? n : M;
    cur = v + orc_vrow_offset(n, M);
    /* lib/stable.c:475 */
    cur[0] = (1.0 + ((double)((int)n - 1) - 2 * a) * prev[0]) / ((double)((int)n - 1) - a);
    /* lib/stable.c:476-480 */
    for (m = 3; m <= last; m++) {
      double num = 1.0 + ((m < n) ? (((double)((int)n - 1) - (double)(int)m * a) * prev[m - 2]) : 0);
      double den = 1.0 / prev[m - 3] + ((double)((int)n - 1) - (double)((int)m - 1) * a);
      cur[m - 2] = num / den;
    }
    prev = cur;
  }
}

/* ------------------------------------------------------------------ lookups */

double orc_S_S(const double *table, const double *S1, unsigned N, unsigned M, unsigned n,
               unsigned m) {
  /* order of tests as lib/stable.c:941-974 */
  if (n == m) return 0;
  if (m == 1) {
    if (n == 0 || n > N) return -HUGE_VAL; /* lib/stable.c:825-827; no lazy growth here */
    return S1[n - 1];
  }
  if (n < m || m == 0) return -HUGE_VAL;
  if (m > M || n > N) return -HUGE_VAL;
  return table[orc_row_offset(n, M) + (m - 2)];
}

double orc_S_V(const double *v, unsigned N, unsigned M, unsigned n, unsigned m) {
  /* lib/stable.c:900-939 without growth: the reference extends when m>=usedM-1 || n>=usedN-1;
   * here such requests are simply served if stored, else 0 */
  if (m < 2) return 0;
  if (n < m) return 0;
  if (m > M || n > N) return 0;
  return v[orc_vrow_offset(n, M) + (m - 2)];
}

double orc_S_U(const double *v, double a, unsigned N, unsigned M, unsigned n, unsigned m) {
  /* lib/stable.c:875-883 */
  if (m == 1) return n - a;
  return n - m * a + 1 / orc_S_V(v, N, M, n, m);
}

double orc_S_UV(const double *v, double a, unsigned N, unsigned M, unsigned n, unsigned m) {
  /* lib/stable.c:885-897 */
  double SV;
  if (m == 1) return -HUGE_VAL;
  if (m == n + 1) return 1;
  if (m == n) return (n + 1.0) / (n - 1.0);
  SV = orc_S_V(v, N, M, n, m);
  return (n - m * a) * SV + 1.0;
}

double orc_S_asympt(double a, unsigned n, unsigned m) {
  if (a == 0) {
    /* lib/stable.c:1058-1065 (glibc gamma() == lgamma()) */
    double ln = log(n);
    return lgamma(n) + (m - 1) * log(ln) - lgamma(m) - lgamma(1 + (m - 1) / ln);
  } else {
    /* lib/stable.c:1066-1082 */
    double prod = 0;
    double la1 = lgamma(1.0 - a);
    double aln = a * log((double)n);
    double np = pow(n, -a);
    prod += lgamma((double)n) - la1 - lgamma((double)m) - (m - 1.0) * log(a) - aln;
    if (np < 1e-5)
      prod -= (m - 1) * np * (1 + np * (0.5 + np / 3.0));
    else
      prod += (m - 1) * log(1.0 - np);
    return prod;
  }
}

/* ------------------------------------------------------------------ integer policies */

void orc_make_clamp(unsigned *initN, unsigned *initM, unsigned *maxN, unsigned *maxM) {
  /* lib/stable.c:118-129, including the :126-127 assignment of maxM */
  if (*maxM < 10) *maxM = 10;
  if (*maxN < *maxM) *maxN = *maxM;
  if (*initM < 10) *initM = 10;
  if (*initN < *initM) *initN = *initM;
  if (*initN > *maxN) *initN = *maxM;
  if (*initN > *maxN) *initN = *maxN;
}

void orc_extend_policy(unsigned usedN, unsigned usedM, unsigned maxN, unsigned maxM, int N, int M,
                       unsigned *newN, unsigned *newM) {
  /* lib/stable.c:568-630.  N, M are signed ints there and the comparisons against the unsigned
   * bounds promote to unsigned; the 1.1 factors go through double and truncate on assignment. */
  N++;
  M++;
  if ((unsigned)N < usedN && (unsigned)M < usedM) {
    *newN = usedN;
    *newM = usedM;
    return;
  }
  if ((unsigned)N < usedN) N = usedN;
  if ((unsigned)N > maxN) N = maxN;
  if ((unsigned)N > usedN) {
    if (N < usedN * 1.1) N = usedN * 1.1;
    if ((unsigned)N < usedN + 50) N = usedN + 50;
    if ((unsigned)N > maxN) N = maxN;
  }
  if ((unsigned)M < usedM) M = usedM;
  if (N < M) M = N;
  if ((unsigned)M > maxM) M = maxM;
  if ((unsigned)M > usedM) {
    if (M < usedM * 1.1) M = usedM * 1.1;
    if ((unsigned)M < usedM + 50) M = usedM + 50;
    if ((unsigned)M > maxM) M = maxM;
    if ((unsigned)M > usedN) M = usedN;
  }
  *newN = N;
  *newM = M;
}

/* ------------------------------------------------------------------ posteriors */

void orc_scan_bounds(int I, const int *K, const uint32_t *nflat, const uint16_t *tflat, int *maxn,
                     int *maxt) {
  int i, k, mn = 1, mt = 1;
  size_t off = 0;
  for (i = 0; i < I; i++) {
    for (k = 0; k < K[i]; k++) {
      if ((int)tflat[off + k] >= mt) mt = tflat[off + k] + 1;
      /* lib/samplea.c:205: unsigned n compared against int maxn (promoted to unsigned) */
      if (nflat[off + k] >= (uint32_t)mn) mn = nflat[off + k] + 1;
    }
    off += K[i];
  }
  *maxn = mn;
  *maxt = mt;
}

double orc_aterms_sum(double x, int I, const int *K, const uint32_t *T, const uint32_t *nflat,
                      const uint16_t *tflat, const double *bpar, const double *table,
                      const double *S1, unsigned N, unsigned M) {
  /* lib/samplea.c:65-81: one running double, restaurant term first, then its pairs in order */
  double val = 0;
  size_t off = 0;
  int i, k;
  for (i = 0; i < I; i++) {
    val += T[i] * log(x) + lgamma(T[i] + bpar[i] / x) - lgamma(bpar[i] / x);
    for (k = 0; k < K[i]; k++)
      if (nflat[off + k] > 1) val += orc_S_S(table, S1, N, M, nflat[off + k], tflat[off + k]);
    off += K[i];
  }
  return val;
}

double orc_aterms(double x, int I, const int *K, const uint32_t *T, const uint32_t *nflat,
                  const uint16_t *tflat, const double *bpar, unsigned N, unsigned M,
                  double *scratch) {
  double *S1 = scratch;
  double *table = scratch + N;
  orc_fill_S(x, N, M, S1, table);
  return orc_aterms_sum(x, I, K, T, nflat, tflat, bpar, table, S1, N, M);
}

/* lib/lgamma.c:36-52 gcache_value for par: log of par (par+1) ... (par+j-1); the cache only memoises */
static double orc_gcache(int j, double par, double lgpar) {
  if (j <= 0) return 0;
  if (j == 1) return log(par);
  if (j == 2) return log(par * (par + 1));
  if (j == 3) return log(par * (par + 1) * (par + 2));
  return lgamma(j + par) - lgpar;
}

double orc_aterms2(double x, int I, const int *K, const uint32_t *T, const uint32_t *nflat, const uint16_t *tflat,
                   const double *bpar, const uint16_t *m) {
  /* lib/samplea.c:85-150 with LGCACHE: restaurant terms, then per pair the sampled table sizes */
  int i, k;
  size_t g = 0;
  double val = 0;
  const double par = 1 - x, lgpar = lgamma(1 - x);
  const uint16_t *mm = m;
  for (i = 0; i < I; i++) {
    val += T[i] * log(x) + lgamma(T[i] + bpar[i] / x) - lgamma(bpar[i] / x);
    for (k = 0; k < K[i]; k++, g++) {
      int n = (int)nflat[g];
      const int t = tflat[g];
      if (n > 0) {
        if (t == n) {
          ;
        } else if (t == 1) {
          val += orc_gcache(n - 1, par, lgpar);
        } else if (t > 1 && t < n) { /* (the reference leaves t = 0 and t > n undefined) */
          int l;
          for (l = t - 2; l >= 0; l--) {
            if (mm[l] > 1) val += orc_gcache(mm[l] - 1, par, lgpar);
            n -= mm[l];
          }
          if (n > 0) val += orc_gcache(n - 1, par, lgpar);
          mm += t - 1;
        }
      }
    }
  }
  return val;
}

static double orc_logminus(double x, double y) {
  /* lib/samplea.c:232-238 */
  if (y >= x) return -HUGE_VAL;
  if (y - x < -80) return x - exp(y - x);
  return x + log(1 - exp(y - x));
}

/* lib/samplea.c:295-320: the table sizes of every pair with 1 < t < n, from the table for discount a
 * and one uniform per such pair (u[], in pair order); m receives sum (t-1) entries; returns that count */
size_t orc_partition(double a, const double *table, const double *S1, unsigned N, unsigned M, int I, const int *K,
                     const uint32_t *nflat, const uint16_t *tflat, const double *u, uint16_t *m) {
  int i, k;
  size_t g = 0, nu = 0;
  uint16_t *mp = m;
  for (i = 0; i < I; i++)
    for (k = 0; k < K[i]; k++, g++) {
      const int t = tflat[g];
      int Nn = (int)nflat[g], Mm, l;
      if (!(t > 1 && t < Nn)) continue;
      {
        const double ptot = orc_S_S(table, S1, N, M, (unsigned)Nn, (unsigned)t);
        double rem = ptot + log(u[nu++]);
        for (Mm = t - 1; Mm >= 1; Mm--) {
          double fact = 0.0;
          for (l = 1; l <= Nn - Mm; l++) {
            double term;
            if (l > 1) fact += log((l - a) * (Nn - l + 1) / (l - 1));
            term = fact + orc_S_S(table, S1, N, M, (unsigned)(Nn - l), (unsigned)Mm) - ptot;
            if (term >= rem) break;
            rem = orc_logminus(rem, term);
          }
          if (l > Nn - Mm) l = Nn - Mm;
          mp[Mm - 1] = (uint16_t)l;
          Nn -= l;
        }
        mp += t - 1;
      }
    }
  return (size_t)(mp - m);
}

double orc_bterms(double x, double Q, double shape, int I, const uint32_t *T, double apar) {
  /* lib/sampleb.c:33-41 */
  int i;
  double lg = lgamma(x / apar);
  double val = -Q * x + (shape - 1) * log(x);
  for (i = 0; i < I; i++) val += lgamma(T[i] + x / apar) - lg;
  return val;
}

double orc_S_approx(int n, int m, float a) {
  /* lib/sapprox.c:28-71 with LS_NOPOLYGAMMA; n-k*a is evaluated in float exactly as written */
  if (n == m) return 0.0;
  if (n < m) return -HUGE_VAL;
  if (m == 1) return lgamma(n - a) - lgamma(1 - a);
  if (m == 2) {
    double ga = lgamma(n - a) - lgamma(1 - a);
    double g2a = lgamma(n - 2 * a) - lgamma(1 - 2 * a);
    return g2a - log(a) + log(exp(ga - g2a) - 1.0);
  }
  if (m == 3) {
    double ga = lgamma(n - a) - lgamma(1 - a);
    double g2a = lgamma(n - 2 * a) - lgamma(1 - 2 * a);
    double g3a = lgamma(n - 3 * a) - lgamma(1 - 3 * a);
    return g3a - 2 * log(a) - log(2.0) + log(exp(ga - g3a) - 2 * exp(g2a - g3a) + 1.0);
  }
  if (m == 4) {
    double ga = lgamma(n - a) - lgamma(1 - a);
    double g2a = lgamma(n - 2 * a) - lgamma(1 - 2 * a);
    double g3a = lgamma(n - 3 * a) - lgamma(1 - 3 * a);
    double g4a = lgamma(n - 4 * a) - lgamma(1 - 4 * a);
    return g4a - 3 * log(a) - log(6.0) +
           log(exp(ga - g4a) - 3 * exp(g2a - g4a) + 3 * exp(g3a - g4a) - 1.0);
  }
  return -HUGE_VAL;
}

/* ------------------------------------------------------------------ timing helpers */

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

double orc_time_fill(double a, unsigned N, unsigned M, int reps, double *S1, double *table) {
  double best = 1e300;
  int r;
  for (r = 0; r < reps; r++) {
    double t0 = now_s(), dt;
    orc_fill_S(a, N, M, S1, table);
    dt = now_s() - t0;
    if (dt < best) best = dt;
  }
  return best;
}

double orc_time_fill_rows(double a, unsigned N, unsigned M, unsigned rows, double *S1,
                          double *table, uint64_t *cells_done) {
  /* bounded sample of the same workload: the first `rows` rows of the (N,M) table */
  unsigned upto = (rows + 2 < N) ? rows + 2 : N;
  double t0 = now_s();
  orc_fill_S(a, upto, M, S1, table);
  *cells_done = orc_cells(upto, M);
  return now_s() - t0;
}

typedef struct {
  const double *a;
  int D, stride, first;
  unsigned N, M;
} batch_arg_t;

static void *batch_worker(void *vp) {
  batch_arg_t *b = vp;
  uint64_t cells = orc_cells(b->N, b->M);
  double *buf = malloc(sizeof(double) * (cells + b->N));
  int d;
  if (!buf) return NULL;
  for (d = b->first; d < b->D; d += b->stride) orc_fill_S(b->a[d], b->N, b->M, buf, buf + b->N);
  free(buf);
  return NULL;
}

double orc_time_fill_batch(const double *a, int D, unsigned N, unsigned M, int threads) {
  pthread_t th[256];
  batch_arg_t arg[256];
  int i;
  double t0;
  if (threads > D) threads = D;
  if (threads > 256) threads = 256;
  if (threads < 1) threads = 1;
  t0 = now_s();
  for (i = 0; i < threads; i++) {
    arg[i].a = a;
    arg[i].D = D;
    arg[i].stride = threads;
    arg[i].first = i;
    arg[i].N = N;
    arg[i].M = M;
    pthread_create(&th[i], NULL, batch_worker, &arg[i]);
  }
  for (i = 0; i < threads; i++) pthread_join(th[i], NULL);
  return now_s() - t0;
}

/*
 * samplea2.c -- the S-free MCMC step for the Pitman-Yor discount a (include/psample.h, under the
 * reference's SAMPLEA_M switch; reference lib/samplea.c:226-340 and aterms2 :85-150).
 *
 * Two stages, as in the reference:
 *  1. For every pair (n,t) with 1 < t < n, sample how the n customers split over the t tables, one
 *     table at a time, from the caller's table S (which must have been built for the current
 *     discount): one uniform per pair, then a walk over S_S values (lib/samplea.c:289-320).  This is
 *     host control flow over host lookups, like ARMS: it consumes the caller's drand48 stream in the
 *     reference's order and reads S through S_S, i.e. through the lazily mirrored device table.
 *  2. Draw a with ARMS (or the slice sampler) from the posterior given those table sizes, aterms2.
 *     That posterior depends on the partition only through the NUMBER of tables of each size, so the
 *     sizes are binned once and every evaluation is a device call on the histogram
 *     (stb_hist_aterms2: restaurant terms + sum over sizes).  No host evaluation path exists.
 */
#define SAMPLEA_M
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/psample.h"
#include "../../include/stb_hip.h"
#include "sampler_trace.h"

typedef struct {
  stb_hist_t *dev;
  int verbose;
} a2_posterior;

static double aterms2(double x, void *vp) {
  a2_posterior *ap = vp;
  double val;
  if (x <= 0) {
    fprintf(stderr, "Illegal discount value in aterms2()\n"); /* lib/samplea.c:97-100 */
    exit(1);
  }
  if (stb_hist_aterms2(ap->dev, &x, 1, &val)) {
    fprintf(stderr, "aterms2(): device evaluation failed: %s\n", stb_last_error());
    exit(1);
  }
  stb_trace_add(x, val);
  return val;
}

/* log(exp(x) - exp(y)), lib/samplea.c:232-238 */
static double logminus(double x, double y) {
  if (y >= x) return -HUGE_VAL;
  if (y - x < -80) return x - exp(y - x);
  return x + log(1 - exp(y - x));
}

static int use_slice2(void) {
  const char *s = getenv("STB_SAMPLER");
#ifdef PSAMPLE_ARS
  return s && strcmp(s, "slice") == 0;
#else
  return !(s && strcmp(s, "ars") == 0);
#endif
}

/* the table sizes the most recent samplea2 call sampled, in the reference's layout (for every pair
 * with 1 < t < n, in (i,k) order, t-1 entries: m[M-1] = customers at the M-th table instantiated);
 * kept for tests and callers that want the partition */
static _Thread_local stcnt_int *last_m; /* (of the calling thread's last samplea2) */
static _Thread_local size_t last_m_count;
size_t stb_samplea2_partition(const stcnt_int **m) {
  if (m) *m = last_m;
  return last_m_count;
}

double samplea2(double mya, stable_t *S, int I, int *K, scnt_int *T, scnt_int **n, stcnt_int **t,
                void (*getval)(scnt_int *n, stcnt_int *t, unsigned i, unsigned k), double *bpar, rngp_t rng,
                int loops, int verbose) {
  double inita[3] = {A_MIN, 1, A_MAX};
  a2_posterior ap;
  uint32_t *cnt;
  stcnt_int *mp;
  size_t n_m = 0;
  unsigned maxn = 2;
  int i, k;

  /* lib/samplea.c:257-270 */
  inita[1] = mya;
  if (fabs(inita[1] - A_MAX) / A_MAX < 0.00001) inita[1] = A_MAX * 0.999 + A_MIN * 0.001;
  if (fabs(inita[1] - A_MIN) / A_MIN < 0.00001) inita[1] = A_MIN * 0.999 + A_MAX * 0.001;
#ifdef SQUEEZEA
  if (inita[1] - SQUEEZEA > A_MIN) inita[0] = inita[1] - SQUEEZEA;
  if (inita[1] + SQUEEZEA < A_MAX) inita[2] = inita[1] + SQUEEZEA;
#endif

  /* lib/samplea.c:283-288: space for the table sizes */
  for (i = 0; i < I; i++)
    for (k = 0; k < K[i]; k++) {
      scnt_int nn;
      stcnt_int tt;
      if (getval)
        getval(&nn, &tt, i, k);
      else {
        nn = n[i][k];
        tt = t[i][k];
      }
      if (tt > 1 && tt < nn) n_m += tt - 1;
      if (nn > maxn) maxn = nn;
    }
  free(last_m);
  last_m = malloc(sizeof(*last_m) * (n_m ? n_m : 1));
  cnt = calloc((size_t)maxn + 2, sizeof(*cnt));
  if (!last_m || !cnt) {
    fprintf(stderr, "Out of memory for samplea()\n"); /* lib/samplea.c:290-293 */
    exit(1);
  }
  last_m_count = n_m;

  /* lib/samplea.c:295-320 (table sizes), and what aterms2 makes of them (:108-145), binned by size */
  mp = last_m;
  for (i = 0; i < I; i++)
    for (k = 0; k < K[i]; k++) {
      scnt_int nn;
      stcnt_int tt;
      if (getval)
        getval(&nn, &tt, i, k);
      else {
        nn = n[i][k];
        tt = t[i][k];
      }
      if (nn == 0 || tt == nn || tt == 0 || tt > nn) continue; /* aterms2: nothing (t = n); undefined in the reference otherwise */
      if (tt == 1) {
        cnt[nn]++; /* one table with all n customers: gcache_value(n-1) */
        continue;
      }
      {
        int N = (int)nn, M, l;
        const double ptot = S_S(S, (unsigned)N, tt);
        double rem = ptot + log(rng_unit(rng));
        for (M = tt - 1; M >= 1; M--) {
          /* each round instantiates another count */
          double fact = 0.0;
          for (l = 1; l <= N - M; l++) {
            double term;
            if (l > 1) fact += log((l - mya) * (N - l + 1) / (l - 1));
            term = fact + S_S(S, (unsigned)(N - l), (unsigned)M) - ptot;
            if (term >= rem) break;
            rem = logminus(rem, term);
          }
          if (l > N - M) l = N - M;
          mp[M - 1] = (stcnt_int)l;
          N -= l;
        }
        /* aterms2 walks the sizes l = t-2 .. 0 and then the remainder */
        for (l = tt - 2; l >= 0; l--)
          if (mp[l] > 1) cnt[mp[l]]++;
        if (N > 1) cnt[N]++;
        mp += tt - 1;
      }
    }

  ap.verbose = verbose;
  ap.dev = stb_hist_create(cnt, maxn + 1, I, T, bpar);
  free(cnt);
  if (!ap.dev) {
    fprintf(stderr, "Out of memory for samplea() (%s)\n", stb_last_error());
    exit(1);
  }
  stb_trace_reset();
  if (!use_slice2()) {
    int code = arms_simple(3, inita, inita + 2, aterms2, &ap, 0, inita + 1, &mya); /* lib/samplea.c:325 (with the data) */
    stb_trace_code(code);
    if (mya < inita[0] || mya > inita[2]) {
      fprintf(stderr, "Arms_simple(apar) returned value out of bounds\n");
      exit(1);
    }
  } else {
    inita[1] = A_MAX; /* lib/samplea.c:332 */
    if (SliceSimple(&mya, aterms2, inita, rng, loops, &ap)) {
      fprintf(stderr, "SliceSimple error\n");
      exit(1);
    }
  }
  stb_hist_free(ap.dev);
  return mya;
}

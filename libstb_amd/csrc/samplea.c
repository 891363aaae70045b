/*
 * samplea.c -- one MCMC step for the Pitman-Yor discount a (include/psample.h).
 *
 * Host control flow as the reference's lib/samplea.c:155-225: bracket the move inside
 * [A_MIN, A_MAX] and +-SQUEEZEA of the current value, scan the counts for the table bounds, then
 * draw with ARMS (or the slice sampler) from the log-posterior `aterms` (lib/samplea.c:46-83).
 *
 * What differs: every evaluation of aterms -- rebuilding the N x M table of log S^n_{m,x} for the
 * trial discount x, gathering S_S(n,t) over all pairs and adding the restaurant terms -- runs on
 * the GPU.  The ragged n[i][k], t[i][k] arrays (or the getval callback) are flattened once per
 * call and kept resident in HBM (stb_groups_*), so an evaluation costs kernel launches and one
 * 8-byte read-back, not a host pass over the data.  There is no host evaluation path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/psample.h"
#include "../../include/stb_hip.h"
#include "sampler_trace.h"

typedef struct {
  stb_groups_t *dev;
  int maxn, maxt;
  int verbose;
} a_posterior;

static double aterms(double x, void *vp) {
  a_posterior *ap = vp;
  double val;
  if (x <= 0) {
    fprintf(stderr, "Illegal discount value in aterms()\n"); /* lib/samplea.c:50-53 */
    exit(1);
  }
  if (ap->verbose > 1) fprintf(stderr, "Extending S for M=%d a=%lf\n", ap->maxt, x); /* :54-56 */
  if (stb_groups_aterms(ap->dev, &x, 1, &val)) {
    fprintf(stderr, "aterms(): device evaluation failed: %s\n", stb_last_error());
    exit(1);
  }
  stb_trace_add(x, val);
  return val;
}

/* environment switch for the sampler family, so both can be exercised from one build:
 * STB_SAMPLER=slice selects SliceSimple, anything else the compile-time default */
static int use_slice(void) {
  const char *s = getenv("STB_SAMPLER");
#ifdef PSAMPLE_ARS
  return s && strcmp(s, "slice") == 0;
#else
  return !(s && strcmp(s, "ars") == 0);
#endif
}

double samplea(double mya, int I, int *K, scnt_int *T, scnt_int **n, stcnt_int **t,
               void (*getval)(scnt_int *n, stcnt_int *t, unsigned i, unsigned k), double *bpar,
               rngp_t rng, int loops, int verbose) {
  double inita[3] = {A_MIN, 1, A_MAX};
  a_posterior ap;
  scnt_int *nflat;
  stcnt_int *tflat;
  size_t G = 0, g = 0;
  unsigned N, M;
  int i, k;

  /* lib/samplea.c:161-177: start point nudged off the ends, move limited to +-SQUEEZEA */
  inita[1] = mya;
  if (fabs(inita[1] - A_MAX) / A_MAX < 0.00001) inita[1] = A_MAX * 0.999 + A_MIN * 0.001;
  if (fabs(inita[1] - A_MIN) / A_MIN < 0.00001) inita[1] = A_MIN * 0.999 + A_MAX * 0.001;
#ifdef SQUEEZEA
  if (inita[1] - SQUEEZEA > A_MIN) inita[0] = inita[1] - SQUEEZEA;
  if (inita[1] + SQUEEZEA < A_MAX) inita[2] = inita[1] + SQUEEZEA;
#endif

  /* flatten the pairs and find the bounds: maxn = max n + 1, maxt = max t + 1, both at least 1
   * (lib/samplea.c:184-208) */
  for (i = 0; i < I; i++) G += (size_t)(K[i] > 0 ? K[i] : 0);
  nflat = malloc(sizeof(*nflat) * (G ? G : 1));
  tflat = malloc(sizeof(*tflat) * (G ? G : 1));
  if (!nflat || !tflat) {
    fprintf(stderr, "Out of memory for S table\n");
    exit(1);
  }
  ap.maxt = 1;
  ap.maxn = 1;
  ap.verbose = verbose;
  for (i = 0; i < I; i++)
    for (k = 0; k < K[i]; k++, g++) {
      if (getval)
        getval(&nflat[g], &tflat[g], i, k);
      else {
        nflat[g] = n[i][k];
        tflat[g] = t[i][k];
      }
      if ((int)tflat[g] >= ap.maxt) ap.maxt = tflat[g] + 1;
      if (nflat[g] >= (scnt_int)ap.maxn) ap.maxn = nflat[g] + 1;
    }
  /* the table aterms builds is S_make(maxn,maxt,maxn,maxt) (lib/samplea.c:60) after S_make's
   * clamps (lib/stable.c:118-129): M = max(maxt,10), N = max(maxn,M) */
  M = ap.maxt < 10 ? 10u : (unsigned)ap.maxt;
  N = (unsigned)ap.maxn < M ? M : (unsigned)ap.maxn;
  ap.dev = stb_groups_create(I, K, T, nflat, tflat, bpar, N, M, 1);
  free(nflat);
  free(tflat);
  if (!ap.dev) {
    fprintf(stderr, "Out of memory for S table (%s)\n", stb_last_error()); /* lib/samplea.c:61-64 */
    exit(1);
  }

  stb_trace_reset();
  if (!use_slice()) {
    int code = arms_simple(3, inita, inita + 2, aterms, &ap, 0, inita + 1, &mya); /* :210 */
    stb_trace_code(code);
    if (mya < inita[0] || mya > inita[2]) {
      fprintf(stderr, "Arms_simple(apar) returned value out of bounds\n");
      exit(1);
    }
  } else {
    inita[1] = A_MAX; /* lib/samplea.c:217: the slice bracket is [lower, A_MAX] */
    if (SliceSimple(&mya, aterms, inita, rng, loops, &ap)) {
      fprintf(stderr, "SliceSimple error\n");
      exit(1);
    }
  }
  stb_groups_free(ap.dev);
  return mya;
}

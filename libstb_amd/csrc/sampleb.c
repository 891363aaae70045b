/*
 * sampleb.c -- one MCMC step for the Pitman-Yor concentration b (include/psample.h).
 *
 * Host control flow as the reference's lib/sampleb.c:79-159: auxiliary q_i ~ Beta(b, N_i) give
 * Q = 1/scale - sum log q_i; for a == 0 the conditional is a Gamma (drawn directly, or through its
 * Gaussian limit above 400); otherwise b is drawn with ARMS (or the slice sampler) from the
 * log-posterior `bterms` (lib/sampleb.c:33-41).  The Beta/Gamma/Gaussian draws stay on the host:
 * they consume glibc's global rand48 stream one after another.
 *
 * Every evaluation of bterms -- the sum over restaurants of lgamma(T_i + x/a) - lgamma(x/a) --
 * runs on the GPU over a device-resident copy of T[] (stb_bterms); there is no host evaluation.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/psample.h"
#include "../../include/stb_hip.h"
#include "sampler_trace.h"

#define NPRE 3 /* abscissae ARMS is known to ask for first (lib/arms.c:117-119) */

typedef struct {
  double shape, Q, apar;
  stb_bctx_t *dev; /* T[] resident in HBM; an evaluation is two launches and one wait */
  /* values evaluated ahead of time, served when ARMS asks for exactly these abscissae */
  int npre;
  double xpre[NPRE], ypre[NPRE];
} b_posterior;

static double bterms(double x, void *vp) {
  b_posterior *bp = vp;
  double val;
  int i;
  for (i = 0; i < bp->npre; i++)
    if (x == bp->xpre[i]) {
      stb_trace_add(x, bp->ypre[i]);
      return bp->ypre[i];
    }
  if (stb_bterms_eval(bp->dev, &x, 1, bp->Q, bp->shape, bp->apar, &val)) {
    fprintf(stderr, "bterms(): device evaluation failed: %s\n", stb_last_error());
    exit(1);
  }
  stb_trace_add(x, val);
  return val;
}

static _Thread_local stb_bctx_t *kept_b; /* one per calling thread, like samplea's kept pairs */

void stb_sampleb_cache_clear(void) {
  if (kept_b) stb_bterms_free(kept_b);
  kept_b = NULL;
}

static int use_slice(void) {
  const char *s = getenv("STB_SAMPLER");
#ifdef PSAMPLE_ARS
  return s && strcmp(s, "slice") == 0;
#else
  return !(s && strcmp(s, "ars") == 0);
#endif
}

/* lib/sampleb.c:51-68: a few fixed-point steps towards the mode, used only to start the slice
 * sampler.  The reference needs digammaInv() for it, which its default build compiles out
 * (lib/digamma.h:25); here the start point is simply the current value. */

double sampleb(double b_in, int I, double shape, double scale, scnt_int *N, scnt_int *T, double apar,
               rngp_t rng, int loops, int verbose) {
  double Q, q, myb;
  int i;
  if (scale <= 0) {
    fprintf(stderr, "Illegal scale in sampleb()\n"); /* lib/sampleb.c:86-89 */
    exit(1);
  }
  Q = 1.0 / scale;
  for (i = 0; i < I; i++) {
    if (N[i] <= 0) continue;
    q = rng_beta(rng, b_in, (int)N[i]); /* lib/sampleb.c:94 */
    if (q <= 0) {
      fprintf(stderr, "Illegal q in sampleb(b=%lf)\n", b_in);
      exit(1);
    }
    Q -= log(q);
  }
  if (apar == 0) {
    /* lib/sampleb.c:101-118: b | q ~ Gamma(shape + sum T, 1/Q) */
    double Tsum = shape;
    for (i = 0; i < I; i++) Tsum += T[i];
    if (Tsum > 400) {
      do {
        myb = Tsum + rng_gaussian(rng, 1) * sqrt(Tsum);
      } while (myb <= 0);
    } else
      myb = rng_gamma(rng, Tsum);
    myb /= Q;
    if (myb < B_MIN) myb = B_MIN;
    if (myb > B_MAX) myb = B_MAX;
    if (verbose > 1) fprintf(stderr, "Sample b ~ gamma(%lg,%lg) = %lf\n", Tsum, Q, myb);
    return myb;
  }
  {
    double initb[3] = {B_MIN, 1, B_MAX};
    b_posterior bp;
    bp.Q = Q;
    bp.apar = apar;
    bp.shape = shape;
    bp.npre = 0;
    /* the device context of the previous call on this thread, when it is large enough: T[] is copied anew (it changes
     * from call to call), the stream, the pinned result buffer and the device memory are kept */
    if (kept_b && stb_bterms_update(kept_b, T, I) == 0)
      bp.dev = kept_b;
    else {
      if (kept_b) stb_bterms_free(kept_b);
      kept_b = bp.dev = stb_bterms_create(T, I);
    }
    if (!bp.dev) {
      fprintf(stderr, "sampleb(): no device memory for T[] (%s)\n", stb_last_error());
      exit(1);
    }
    stb_trace_reset();
    if (!use_slice()) {
      int code;
      /* lib/sampleb.c:127-139 */
      initb[1] = b_in;
      if (fabs(initb[1] - B_MAX) / B_MAX < 0.00001) initb[1] = B_MAX * 0.999 + B_MIN * 0.001;
      if (fabs(initb[1] - B_MIN) / B_MIN < 0.00001) initb[1] = B_MIN * 0.999 + B_MAX * 0.001;
      {
        /* ARMS starts from three abscissae it fixes before any evaluation (lib/arms.c:117-119, the same
         * expression here, so the same bits): evaluate them in ONE device call */
        double x3[NPRE], y3[NPRE];
        for (i = 0; i < NPRE; i++) x3[i] = initb[0] + (i + 1.0) * (initb[2] - initb[0]) / (NPRE + 1.0);
        if (stb_bterms_eval(bp.dev, x3, NPRE, bp.Q, bp.shape, bp.apar, y3)) {
          fprintf(stderr, "bterms(): device evaluation failed: %s\n", stb_last_error());
          exit(1);
        }
        for (i = 0; i < NPRE; i++) {
          bp.xpre[i] = x3[i];
          bp.ypre[i] = y3[i];
        }
        bp.npre = NPRE;
      }
      code = arms_simple(3, initb, initb + 2, bterms, &bp, 0, initb + 1, &myb);
      stb_trace_code(code);
      if (myb < B_MIN || myb > B_MAX) {
        fprintf(stderr, "Arms_simple(bpar) returned value out of bounds\n");
        exit(1);
      }
    } else {
      /* lib/sampleb.c:141-153 */
      myb = b_in;
      if (verbose > 1) fprintf(stderr, "Max b (%lg,%lg) -> %lg\n", b_in, Q, myb);
      initb[1] = B_MAX;
      if (SliceSimple(&myb, bterms, initb, rng, loops, &bp)) {
        fprintf(stderr, "SliceSimple error\n");
        exit(1);
      }
    }
    /* (bp.dev stays: kept_b; stb_sampler_cache_clear drops it) */
    if (verbose > 1) fprintf(stderr, "Sample b ~ G(%lg) = %lf\n", Q, myb);
  }
  return myb;
}

/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Probe layer compiled TOGETHER WITH the real reference sources (which stay under
 * /root/reference; see oracle/Makefile) into oracle/_ref/libstb_ref.so.  It contains no
 * restatement of the reference: it pulls the two sampler translation units in by #include so
 * that their file-static log-posteriors aterms() (lib/samplea.c:46) and bterms()
 * (lib/sampleb.c:33) can be evaluated at chosen abscissae, and adds field accessors so Python
 * (ctypes) never has to know the reference's struct layout.
 *
 * Used by tests/golden/gen_golden.py (to make fixtures) and by tests that pin
 * oracle/stb_oracle.c and the product's host logic against the real thing.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "psample.h" /* -I$(REF)/lib : declares arms_simple via arms.h before it is hooked */

/* every arms_simple() call made by samplea()/sampleb() is routed through a recording wrapper so
 * the sequence of abscissae, the log-posterior values and ARMS' return code can be dumped */
static int ref_arms_hook(int ninit, double *xl, double *xr,
                         double (*myfunc)(double x, void *mydata), void *mydata, int dometrop,
                         double *xprev, double *xsamp);
/* samplea2() (compiled only with -DSAMPLEA_M, into _ref/libstb_ref_m.so) hands ARMS a NULL `mydata`
 * (lib/samplea.c:325-326), which aterms2 then dereferences: while samplea.c is being included the
 * hook substitutes the address of the caller's local `ald` for a NULL argument -- the reference's
 * source is untouched, and samplea()'s own call (which passes &ald) is unaffected. */
#define arms_simple(ni, xl, xr, f, d, m, xp, xs) \
  ref_arms_hook(ni, xl, xr, f, ((d) ? (void *)(d) : (void *)&ald), m, xp, xs)
#ifdef REF_SLICE_A
/* _ref/libstb_ref_slice.so: samplea's OTHER branch (lib/samplea.c:216-221), which the reference compiles when
 * PSAMPLE_ARS is not defined (lib/psample.h:37).  The switch is turned off for samplea.c alone -- sampleb's slice
 * branch (lib/sampleb.c:141-153) needs digammaInv, which lib/digamma.h:25 compiles out in the shipped configuration --
 * and SliceSimple (lib/sslice.c, compiled in as it is) is reached through a hook that records every evaluation. */
#undef PSAMPLE_ARS
int SliceSimple(double *xp, double (*post)(double, void *), double *bounds, rngp_t rng, int loops, void *pars);
static int ref_slice_hook(double *xp, double (*post)(double, void *), double *bounds, rngp_t rng, int loops, void *pars);
#define SliceSimple ref_slice_hook
#include "samplea.c"
#undef SliceSimple
#define PSAMPLE_ARS
#else
#include "samplea.c" /* -I$(REF)/lib : lib/samplea.c (ALData, aterms, samplea; aterms2, samplea2) */
#endif
#undef arms_simple
#define arms_simple ref_arms_hook
#include "sampleb.c" /* -I$(REF)/lib : lib/sampleb.c (BLData, bterms, sampleb) */
#undef arms_simple

#define REF_TRACE_CAP 1024
static struct {
  double (*f)(double, void *);
  void *d;
  int n, code;
  double xl, xr;
  double xs[REF_TRACE_CAP], ys[REF_TRACE_CAP];
} ref_trace;

static double ref_tramp(double x, void *unused) {
  double y = ref_trace.f(x, ref_trace.d);
  (void)unused;
  if (ref_trace.n < REF_TRACE_CAP) {
    ref_trace.xs[ref_trace.n] = x;
    ref_trace.ys[ref_trace.n] = y;
  }
  ref_trace.n++;
  return y;
}
static void *ref_last_mydata;
static int ref_arms_hook(int ninit, double *xl, double *xr,
                         double (*myfunc)(double x, void *mydata), void *mydata, int dometrop,
                         double *xprev, double *xsamp) {
  ref_last_mydata = mydata;
  ref_trace.f = myfunc;
  ref_trace.d = mydata;
  ref_trace.n = 0;
  ref_trace.xl = *xl;
  ref_trace.xr = *xr;
  ref_trace.code = arms_simple(ninit, xl, xr, ref_tramp, NULL, dometrop, xprev, xsamp);
  return ref_trace.code;
}
#ifdef REF_SLICE_A
static int ref_slice_hook(double *xp, double (*post)(double, void *), double *bounds, rngp_t rng, int loops, void *pars) {
  ref_last_mydata = pars;
  ref_trace.f = post;
  ref_trace.d = pars;
  ref_trace.n = 0;
  ref_trace.xl = bounds[0];
  ref_trace.xr = bounds[1];
  ref_trace.code = SliceSimple(xp, ref_tramp, bounds, rng, loops, NULL);
  return ref_trace.code;
}
#endif
int ref_trace_count(void) { return ref_trace.n; }
int ref_trace_code(void) { return ref_trace.code; }
double ref_trace_x(int i) { return ref_trace.xs[i]; }
double ref_trace_y(int i) { return ref_trace.ys[i]; }
double ref_trace_xl(void) { return ref_trace.xl; }
double ref_trace_xr(void) { return ref_trace.xr; }

/* ---- stable_t accessors (lib/stable.h:62-113) ---- */
/* the reference's accessors in a loop (bench.py's drop-in leg: S_remake + host look-ups, beside the product's) */
void ref_probe(stable_t *sp, int which, const unsigned *n, const unsigned *m, size_t G, double *out) {
  size_t g;
  if (which)
    for (g = 0; g < G; g++) out[g] = S_V(sp, n[g], m[g]);
  else
    for (g = 0; g < G; g++) out[g] = S_S(sp, n[g], m[g]);
}
unsigned ref_usedN(stable_t *sp) { return sp->usedN; }
unsigned ref_usedM(stable_t *sp) { return sp->usedM; }
unsigned ref_usedN1(stable_t *sp) { return sp->usedN1; }
unsigned ref_maxN(stable_t *sp) { return sp->maxN; }
unsigned ref_maxM(stable_t *sp) { return sp->maxM; }
unsigned ref_startM(stable_t *sp) { return sp->startM; }
unsigned ref_memalloced(stable_t *sp) { return sp->memalloced; }
double ref_lga(stable_t *sp) { return sp->lga; }
double ref_a(stable_t *sp) { return sp->a; }
size_t ref_sizeof_stable(void) { return sizeof(stable_t); }

/* copy row n (3<=n<=usedN) of the double S table: entries M=2..min(n-1,usedM); returns count */
unsigned ref_copy_S_row(stable_t *sp, unsigned n, double *out) {
  unsigned len, i;
  if (!sp->S || n < 3 || n > sp->usedN) return 0;
  len = (n - 2 < sp->usedM - 1) ? n - 2 : sp->usedM - 1;
  for (i = 0; i < len; i++) out[i] = sp->S[n - 3][i];
  return len;
}
/* same for the V table: row n (2<=n<=usedN), entries M=2..min(n,usedM) */
unsigned ref_copy_V_row(stable_t *sp, unsigned n, double *out) {
  unsigned len, i;
  if (!sp->V || n < 2 || n > sp->usedN) return 0;
  len = (n - 1 < sp->usedM - 1) ? n - 1 : sp->usedM - 1;
  for (i = 0; i < len; i++) out[i] = sp->V[n - 2][i];
  return len;
}
/* float-storage variants */
unsigned ref_copy_Sf_row(stable_t *sp, unsigned n, float *out) {
  unsigned len, i;
  if (!sp->Sf || n < 3 || n > sp->usedN) return 0;
  len = (n - 2 < sp->usedM - 1) ? n - 2 : sp->usedM - 1;
  for (i = 0; i < len; i++) out[i] = sp->Sf[n - 3][i];
  return len;
}
unsigned ref_copy_Vf_row(stable_t *sp, unsigned n, float *out) {
  unsigned len, i;
  if (!sp->Vf || n < 2 || n > sp->usedN) return 0;
  len = (n - 1 < sp->usedM - 1) ? n - 1 : sp->usedM - 1;
  for (i = 0; i < len; i++) out[i] = sp->Vf[n - 2][i];
  return len;
}
unsigned ref_copy_S1(stable_t *sp, double *out, unsigned cnt) {
  unsigned i;
  if (cnt > sp->usedN1) cnt = sp->usedN1;
  for (i = 0; i < cnt; i++) out[i] = sp->S1[i];
  return cnt;
}

/* ---- aterms probe: flat (CSR) group arrays in, same maxn/maxt scan as samplea (:186-208) ---- */
typedef struct {
  ALData ald;
  scnt_int **n;
  stcnt_int **t;
} ref_aprobe_t;

void *ref_aterms_open(int I, int *K, scnt_int *T, scnt_int *nflat, stcnt_int *tflat,
                      double *bpar) {
  ref_aprobe_t *p = calloc(1, sizeof(*p));
  int i, k;
  size_t off = 0;
  p->n = malloc(sizeof(*p->n) * (I > 0 ? I : 1));
  p->t = malloc(sizeof(*p->t) * (I > 0 ? I : 1));
  for (i = 0; i < I; i++) {
    p->n[i] = nflat + off;
    p->t[i] = tflat + off;
    off += K[i];
  }
  p->ald.T = T;
  p->ald.n = p->n;
  p->ald.t = p->t;
  p->ald.I = I;
  p->ald.K = K;
  p->ald.val = NULL;
  p->ald.bpar = bpar;
  p->ald.verbose = 0;
  p->ald.maxt = 1;
  p->ald.maxn = 1;
  p->ald.S = NULL;
  for (i = 0; i < I; i++)
    for (k = 0; k < K[i]; k++) {
      if (p->t[i][k] >= p->ald.maxt) p->ald.maxt = p->t[i][k] + 1;
      if (p->n[i][k] >= p->ald.maxn) p->ald.maxn = p->n[i][k] + 1;
    }
  return p;
}
double ref_aterms_eval(void *h, double x) { return aterms(x, &((ref_aprobe_t *)h)->ald); }
int ref_aterms_maxn(void *h) { return ((ref_aprobe_t *)h)->ald.maxn; }
int ref_aterms_maxt(void *h) { return ((ref_aprobe_t *)h)->ald.maxt; }
void ref_aterms_close(void *h) {
  ref_aprobe_t *p = h;
  if (p->ald.S) S_free(p->ald.S);
  free(p->n);
  free(p->t);
  free(p);
}

/* ---- bterms probe ---- */
double ref_bterms_eval(double x, double Q, double shape, int I, scnt_int *T, double apar) {
  BLData bld;
  bld.Q = Q;
  bld.I = I;
  bld.T = T;
  bld.apar = apar;
  bld.shape = shape;
  return bterms(x, &bld);
}

/* ---- samplea/sampleb on flat arrays (build the ragged pointer vectors here) ---- */
double ref_samplea_flat(double a, int I, int *K, scnt_int *T, scnt_int *nflat, stcnt_int *tflat,
                        double *bpar, int loops, int verbose) {
  scnt_int **n = malloc(sizeof(*n) * (I > 0 ? I : 1));
  stcnt_int **t = malloc(sizeof(*t) * (I > 0 ? I : 1));
  size_t off = 0;
  int i;
  double r;
  for (i = 0; i < I; i++) {
    n[i] = nflat + off;
    t[i] = tflat + off;
    off += K[i];
  }
  r = samplea(a, I, K, T, n, t, NULL, bpar, NULL, loops, verbose);
  free(n);
  free(t);
  return r;
}

/* ---- ARMS probe on a family of analytic log-densities, so the product's own ARMS can be
 *      compared bit-for-bit with lib/arms.c under the same rand() stream ---- */
typedef struct {
  int kind;
  double p0, p1, p2;
  int calls;
  double xs[256];
} ref_dens_t;

static double ref_density(double x, void *vd) {
  ref_dens_t *d = vd;
  if (d->calls < 256) d->xs[d->calls] = x;
  d->calls++;
  switch (d->kind) {
    case 0: /* gaussian */
      return -0.5 * (x - d->p0) * (x - d->p0) / (d->p1 * d->p1);
    case 1: /* gamma(shape p0, rate p1) */
      return (d->p0 - 1.0) * log(x) - d->p1 * x;
    case 2: /* beta-like on (0,1) */
      return (d->p0 - 1.0) * log(x) + (d->p1 - 1.0) * log(1.0 - x);
    case 3: /* steep: p2 scales a concave quartic */
      return -d->p2 * ((x - d->p0) * (x - d->p0) * (x - d->p0) * (x - d->p0)) - d->p1 * x;
    default: /* non log-concave (bimodal) -> exercises the 2000 path */
      return log(exp(-0.5 * (x - d->p0) * (x - d->p0)) + exp(-0.5 * (x - d->p1) * (x - d->p1)));
  }
}
/* returns arms_simple's code; *xsamp gets the draw; xs_out (cap entries) the abscissae */
int ref_arms_probe(int kind, double p0, double p1, double p2, double xl, double xr, int dometrop,
                   double xprev, double *xsamp, int *ncalls, double *xs_out, int cap) {
  ref_dens_t d;
  int err, i;
  memset(&d, 0, sizeof(d));
  d.kind = kind;
  d.p0 = p0;
  d.p1 = p1;
  d.p2 = p2;
  err = arms_simple(3, &xl, &xr, ref_density, &d, dometrop, &xprev, xsamp);
  *ncalls = d.calls;
  for (i = 0; i < cap && i < d.calls && i < 256; i++) xs_out[i] = d.xs[i];
  return err;
}

int ref_slice_probe(int kind, double p0, double p1, double p2, double lo, double hi, double *x,
                    int loops, int *ncalls) {
  ref_dens_t d;
  double bounds[2];
  int err;
  extern int SliceSimple(double *xp, double (*post)(double, void *), double *bounds, rngp_t rng,
                         int loops, void *pars);
  memset(&d, 0, sizeof(d));
  d.kind = kind;
  d.p0 = p0;
  d.p1 = p1;
  d.p2 = p2;
  bounds[0] = lo;
  bounds[1] = hi;
  err = SliceSimple(x, ref_density, bounds, NULL, loops, &d);
  *ncalls = d.calls;
  return err;
}

#ifdef SAMPLEA_M
/* ---- the S-free discount sampler (lib/samplea.c:85-150 aterms2, :244-340 samplea2) ---- */
/* samplea2 on flat arrays; the table partition it sampled (ALData.m, never freed by the reference) can
 * be read back afterwards */
static stcnt_int *ref_m_last;
static size_t ref_m_count;
double ref_samplea2_flat(double a, stable_t *sp, int I, int *K, scnt_int *T, scnt_int *nflat, stcnt_int *tflat,
                         double *bpar, int loops, int verbose) {
  scnt_int **n = malloc(sizeof(*n) * (I > 0 ? I : 1));
  stcnt_int **t = malloc(sizeof(*t) * (I > 0 ? I : 1));
  size_t off = 0;
  int i, k;
  double r;
  ref_m_count = 0;
  for (i = 0; i < I; i++) {
    n[i] = nflat + off;
    t[i] = tflat + off;
    for (k = 0; k < K[i]; k++)
      if (t[i][k] > 1 && t[i][k] < n[i][k]) ref_m_count += t[i][k] - 1;
    off += K[i];
  }
  r = samplea2(a, sp, I, K, T, n, t, NULL, bpar, NULL, loops, verbose);
  ref_m_last = ((ALData *)ref_last_mydata)->m;
  free(n);
  free(t);
  return r;
}
size_t ref_m_size(void) { return ref_m_count; }
unsigned ref_m_get(size_t i) { return ref_m_last[i]; }

/* aterms2(x) for a GIVEN partition m (as samplea2 lays it out) */
double ref_aterms2_eval(double x, int I, int *K, scnt_int *T, scnt_int *nflat, stcnt_int *tflat, double *bpar,
                        stcnt_int *m) {
  ALData ald;
  scnt_int **n = malloc(sizeof(*n) * (I > 0 ? I : 1));
  stcnt_int **t = malloc(sizeof(*t) * (I > 0 ? I : 1));
  size_t off = 0;
  int i;
  double r;
  for (i = 0; i < I; i++) {
    n[i] = nflat + off;
    t[i] = tflat + off;
    off += K[i];
  }
  memset(&ald, 0, sizeof(ald));
  ald.T = T;
  ald.n = n;
  ald.t = t;
  ald.I = I;
  ald.K = K;
  ald.val = NULL;
  ald.bpar = bpar;
  ald.m = m;
  r = aterms2(x, &ald);
  free(n);
  free(t);
  return r;
}
#endif

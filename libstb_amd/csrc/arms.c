/*
 * arms.c -- derivative-free adaptive rejection (Metropolis) sampling, Gilks/Best/Tan 1995.
 *
 * Host control loop behind include/arms.h; behaviour-identical to the reference's lib/arms.c
 * (arms_simple :98-125, arms :129-264, initial :268-375, invert :398-462, test :466-563,
 * update :567-663, cumulate :667-697, meet :701-806, area :810-831, expshift/logshift :835-853,
 * u_random :913-918): same abscissae, same rand() consumption, same return codes, and every
 * floating-point expression keeps the reference's association order so that the draw is equal
 * to the last bit for a given srand() seed (the build uses -ffp-contract=off).
 *
 * The construction is different: the envelope is an x-ordered ARRAY of knots that alternates
 *     bound, density, crossing, density, crossing, ..., density, bound
 * and grows by inserting a (density, crossing) pair in place; "left/right neighbour" is index -1/+1,
 * where the reference walks a pointer-linked list threaded through an append-only pool.
 *
 * Each log-density evaluation is one device round trip when the callback is aterms/bterms, which
 * is why this loop stays on the host: it is ~10-20 serial, data-dependent evaluations.
 */
#include "../../include/arms.h"

#include <math.h>
#include <stdlib.h>

#define XEPS 0.00001 /* lib/arms.c:49: minimum relative distance of a new knot from its neighbours */
#define YEPS 0.1     /* lib/arms.c:50: |dy| below which a piece is integrated as a straight line */
#define EYEPS 0.001  /* lib/arms.c:51 */
#define RAND_TOP 2147483647 /* lib/arms.c:48 A_RAND_MAX */

typedef struct {
  double x, y;  /* position; y is the envelope height (log scale) */
  double ey;    /* exp(y - ymax + YCEIL) */
  double cum;   /* integral of the exponentiated envelope up to x */
  int on;       /* 1: y is an evaluated log-density, 0: a crossing of two chords or a bound */
} knot;

typedef struct {
  knot *k;
  int n, cap;       /* knots in use / allowed */
  double ymax;
  double convex;
  int *neval;
  double (*f)(double, void *);
  void *fdata;
  int metro;        /* Metropolis step enabled */
  double xprev, yprev;
} hull;

/* a candidate point inside the piece (lo, lo+1) */
typedef struct {
  double x, y, ey;
  int on;
  int lo;
} trial;

static double evaluate(hull *h, double x) {
  /* lib/arms.c:857-875 */
  double y = h->f(x, h->fdata);
  (*h->neval)++;
  return y;
}

double expshift(double y, double y0) {
  /* lib/arms.c:835-845 */
  if (y - y0 > -2.0 * YCEIL) return exp(y - y0 + YCEIL);
  return 0.0;
}

static double logshift(double y, double y0) { return (log(y) + y0 - YCEIL); /* lib/arms.c:849-853 */ }

static double uniform01(void) {
  /* lib/arms.c:913-918 */
  return ((double)rand() + 0.5) / ((double)RAND_TOP + 1.0);
}

/* lib/arms.c:701-806: place crossing knot i where the chords through its neighbours meet */
static int cross(hull *h, int i) {
  knot *k = h->k;
  const int last = h->n - 1;
  double gl = 0, gr = 0, grl = 0, dl = 0, dr = 0;
  const int il = (i >= 3);            /* a chord exists on the left:  knots i-3, i-1 */
  const int ir = (i + 3 <= last);     /* a chord exists on the right: knots i+1, i+3 */
  const int irl = (i >= 1 && i + 1 <= last);
  if (k[i].on) exit(30);
  if (il) gl = (k[i - 1].y - k[i - 3].y) / (k[i - 1].x - k[i - 3].x);
  if (ir) gr = (k[i + 1].y - k[i + 3].y) / (k[i + 1].x - k[i + 3].x);
  if (irl) grl = (k[i + 1].y - k[i - 1].y) / (k[i + 1].x - k[i - 1].x);

  if (irl && il && (gl < grl)) {
    if (!h->metro) return 1; /* not log-concave and no Metropolis step to repair it */
    gl = gl + (1.0 + h->convex) * (grl - gl);
  }
  if (irl && ir && (gr > grl)) {
    if (!h->metro) return 1;
    gr = gr + (1.0 + h->convex) * (grl - gr);
  }
  if (il && irl) {
    dr = (gl - grl) * (k[i + 1].x - k[i - 1].x);
    if (dr < YEPS) dr = YEPS;
  }
  if (ir && irl) {
    dl = (grl - gr) * (k[i + 1].x - k[i - 1].x);
    if (dl < YEPS) dl = YEPS;
  }
  if (il && ir && irl) {
    k[i].x = (dl * k[i + 1].x + dr * k[i - 1].x) / (dl + dr);
    k[i].y = (dl * k[i + 1].y + dr * k[i - 1].y + dl * dr) / (dl + dr);
  } else if (il && irl) {
    k[i].x = k[i + 1].x;
    k[i].y = k[i + 1].y + dr;
  } else if (ir && irl) {
    k[i].x = k[i - 1].x;
    k[i].y = k[i - 1].y + dl;
  } else if (il) {
    k[i].y = k[i - 1].y + gl * (k[i].x - k[i - 1].x); /* right bound */
  } else if (ir) {
    k[i].y = k[i + 1].y - gr * (k[i + 1].x - k[i].x); /* left bound */
  } else {
    exit(31);
  }
  if ((i >= 1 && k[i].x < k[i - 1].x) || (i + 1 <= last && k[i].x > k[i + 1].x)) exit(32);
  return 0;
}

/* lib/arms.c:810-831 */
static double piece_area(const knot *k, int i) {
  if (i == 0) exit(1);
  if (k[i - 1].x == k[i].x) return 0.;
  if (fabs(k[i].y - k[i - 1].y) < YEPS) return 0.5 * (k[i].ey + k[i - 1].ey) * (k[i].x - k[i - 1].x);
  return ((k[i].ey - k[i - 1].ey) / (k[i].y - k[i - 1].y)) * (k[i].x - k[i - 1].x);
}

/* lib/arms.c:667-697 */
static void integrate(hull *h) {
  knot *k = h->k;
  int i;
  h->ymax = k[0].y;
  for (i = 1; i < h->n; i++)
    if (k[i].y > h->ymax) h->ymax = k[i].y;
  for (i = 0; i < h->n; i++) k[i].ey = expshift(k[i].y, h->ymax);
  k[0].cum = 0.;
  for (i = 1; i < h->n; i++) k[i].cum = k[i - 1].cum + piece_area(k, i);
}

/* lib/arms.c:398-462: the x at cumulative probability prob under the envelope */
static void invert_cdf(hull *h, double prob, trial *p) {
  const knot *k = h->k;
  int q = h->n - 1;
  double u, xl = 0, xr = 0, yl, yr, eyl, eyr, prop;
  u = prob * k[q].cum;
  while (k[q - 1].cum > u) q--;
  p->lo = q - 1;
  p->on = 0;
  prop = (u - k[q - 1].cum) / (k[q].cum - k[q - 1].cum);
  if (k[q - 1].x == k[q].x) {
    p->x = k[q].x;
    p->y = k[q].y;
    p->ey = k[q].ey;
  } else {
    xl = k[q - 1].x;
    xr = k[q].x;
    yl = k[q - 1].y;
    yr = k[q].y;
    eyl = k[q - 1].ey;
    eyr = k[q].ey;
    if (fabs(yr - yl) < YEPS) {
      /* straight-line piece, as integrated by piece_area */
      if (fabs(eyr - eyl) > EYEPS * fabs(eyr + eyl)) {
        p->x = xl + ((xr - xl) / (eyr - eyl)) * (-eyl + sqrt((1. - prop) * eyl * eyl + prop * eyr * eyr));
      } else {
        p->x = xl + (xr - xl) * prop;
      }
      p->ey = ((p->x - xl) / (xr - xl)) * (eyr - eyl) + eyl;
      p->y = logshift(p->ey, h->ymax);
    } else {
      /* exponential piece */
      p->x = xl + ((xr - xl) / (yr - yl)) * (-yl + logshift(((1. - prop) * eyl + prop * eyr), h->ymax));
      p->y = ((p->x - xl) / (xr - xl)) * (yr - yl) + yl;
      p->ey = expshift(p->y, h->ymax);
    }
  }
  /* lib/arms.c:459: the guard compares against xl/xr even when the piece had zero length (both
   * still 0 then), exactly as the reference does */
  if ((p->x < xl) || (p->x > xr)) exit(1);
}

/* lib/arms.c:567-663: insert an evaluated point (and the crossing that separates it from the
 * density knot it sits next to) into the envelope */
static int absorb(hull *h, trial *p) {
  knot *k = h->k;
  int lo = p->lo, q, m, i, l2, r2;
  if (!p->on || (h->n > h->cap - 2)) return 0; /* nothing evaluated, or envelope full: ignore */
  /* open a two-knot gap after lo */
  for (i = h->n - 1; i > lo; i--) k[i + 2] = k[i];
  h->n += 2;
  if (k[lo].on && !k[lo + 3].on) {
    m = lo + 1; /* density on the left: the new crossing goes between it and the new point */
    q = lo + 2;
  } else if (!k[lo].on && k[lo + 3].on) {
    q = lo + 1;
    m = lo + 2;
  } else {
    exit(10);
  }
  k[q].x = p->x;
  k[q].y = p->y;
  k[q].on = 1;
  k[m].on = 0;
  /* keep the new point at least XEPS (relative) away from the next density knot / bound */
  l2 = (q - 2 >= 0) ? q - 2 : q - 1;
  r2 = (q + 2 <= h->n - 1) ? q + 2 : q + 1;
  if (k[q].x < (1. - XEPS) * k[l2].x + XEPS * k[r2].x) {
    k[q].x = (1. - XEPS) * k[l2].x + XEPS * k[r2].x;
    k[q].y = evaluate(h, k[q].x);
  } else if (k[q].x > XEPS * k[l2].x + (1. - XEPS) * k[r2].x) {
    k[q].x = XEPS * k[l2].x + (1. - XEPS) * k[r2].x;
    k[q].y = evaluate(h, k[q].x);
  }
  /* re-place the (up to four) crossings whose chords changed */
  if (cross(h, q - 1)) return 1;
  if (cross(h, q + 1)) return 1;
  if (q - 2 >= 0)
    if (cross(h, q - 3)) return 1;
  if (q + 2 <= h->n - 1)
    if (cross(h, q + 3)) return 1;
  integrate(h);
  return 0;
}

/* lib/arms.c:466-563: squeeze test, rejection test, optional Metropolis step.
 * 1 accept, 0 reject, -1 envelope violated without Metropolis */
static int judge(hull *h, trial *p) {
  const knot *k = h->k;
  double u, y, ysqueez, ynew, yold, znew, zold, w;
  int ql, qr;
  const int lo = p->lo, hi = p->lo + 1;

  u = uniform01() * p->ey;
  y = logshift(u, h->ymax);

  if (!h->metro && (lo >= 1) && (hi <= h->n - 2)) {
    ql = k[lo].on ? lo : lo - 1;
    qr = k[hi].on ? hi : hi + 1;
    ysqueez = (k[qr].y * (p->x - k[ql].x) + k[ql].y * (k[qr].x - p->x)) / (k[qr].x - k[ql].x);
    if (y <= ysqueez) return 1;
  }

  ynew = evaluate(h, p->x);

  if (!h->metro || (h->metro && (y >= ynew))) {
    p->y = ynew;
    p->ey = expshift(p->y, h->ymax);
    p->on = 1;
    if (absorb(h, p)) return -1;
    if (y >= ynew) return 0;
    return 1;
  }

  /* Metropolis: compare against the previous iterate under the current envelope */
  k = h->k;
  yold = h->yprev;
  ql = 0;
  while (k[ql + 1].x < h->xprev) ql++;
  qr = ql + 1;
  w = (h->xprev - k[ql].x) / (k[qr].x - k[ql].x);
  zold = k[ql].y + w * (k[qr].y - k[ql].y);
  znew = p->y;
  if (yold < zold) zold = yold;
  if (ynew < znew) znew = ynew;
  w = ynew - znew - yold + zold;
  if (w > 0.0) w = 0.0;
  if (w > -YCEIL)
    w = exp(w);
  else
    w = 0.0;
  u = uniform01();
  if (u > w) {
    /* stay: hand back the previous iterate */
    p->x = h->xprev;
    p->y = h->yprev;
    p->ey = expshift(p->y, h->ymax);
    p->on = 1;
    p->lo = ql;
  } else {
    h->xprev = p->x;
    h->yprev = ynew;
  }
  return 1;
}

/* lib/arms.c:268-375 */
static int build_hull(hull *h, const double *xinit, int ninit, double xl, double xr, int npoint) {
  int i, j, kx, mpoint;
  if (ninit < 3) return 1001;
  mpoint = 2 * ninit + 1;
  if (npoint < mpoint) return 1002;
  if ((xinit[0] <= xl) || (xinit[ninit - 1] >= xr)) return 1003;
  for (i = 1; i < ninit; i++)
    if (xinit[i] <= xinit[i - 1]) return 1004;
  if (h->convex < 0.0) return 1008;
  *h->neval = 0;
  h->cap = npoint;
  h->k = (knot *)malloc((size_t)(npoint + 2) * sizeof(knot));
  if (!h->k) return 1006;
  h->n = mpoint;
  h->k[0].x = xl;
  h->k[0].on = 0;
  for (j = 1, kx = 0; j < mpoint - 1; j++) {
    if (j % 2) {
      h->k[j].x = xinit[kx++];
      h->k[j].y = evaluate(h, h->k[j].x);
      h->k[j].on = 1;
    } else {
      h->k[j].on = 0;
    }
  }
  h->k[mpoint - 1].x = xr;
  h->k[mpoint - 1].on = 0;
  for (j = 0; j < mpoint; j += 2)
    if (cross(h, j)) return 2000;
  integrate(h);
  return 0;
}

int arms(double *xinit, int ninit, double *xl, double *xr, double (*myfunc)(double x, void *mydata),
         void *mydata, double *convex, int npoint, int dometrop, double *xprev, double *xsamp,
         int nsamp, double *qcent, double *xcent, int ncent, int *neval) {
  hull h;
  trial p;
  int msamp = 0, i, err, rejections = 0;

  for (i = 0; i < ncent; i++)
    if ((qcent[i] < 0.0) || (qcent[i] > 100.0)) return 1005;

  h.k = NULL;
  h.f = myfunc;
  h.fdata = mydata;
  h.convex = *convex;
  h.neval = neval;
  h.metro = dometrop;
  h.xprev = h.yprev = 0;

  err = build_hull(&h, xinit, ninit, *xl, *xr, npoint);
  if (err) {
    free(h.k);
    return err;
  }
  if (h.metro) {
    if ((*xprev < *xl) || (*xprev > *xr)) {
      /* lib/arms.c:207-217: previous iterate outside the bounds */
      if (*xprev < *xl) *xsamp = *xl;
      if (*xprev > *xr) *xsamp = *xr;
      free(h.k);
      return 1007;
    }
    h.xprev = *xprev;
    h.yprev = evaluate(&h, *xprev);
  }

  do {
    invert_cdf(&h, uniform01(), &p);
    i = judge(&h, &p);
    if (i == 1) {
      xsamp[msamp++] = p.x;
    } else if (i != 0) {
      free(h.k);
      return 2000;
    }
    /* lib/arms.c:241-248: give up after 100 passes through the loop */
    rejections++;
    if (rejections > 100) {
      free(h.k);
      return 2001;
    }
  } while (msamp < nsamp);

  for (i = 0; i < ncent; i++) {
    invert_cdf(&h, qcent[i] / 100.0, &p);
    xcent[i] = p.x;
  }
  free(h.k);
  return 0;
}

int arms_simple(int ninit, double *xl, double *xr, double (*myfunc)(double x, void *mydata),
                void *mydata, int dometrop, double *xprev, double *xsamp) {
  /* lib/arms.c:98-125 */
  double convex = 1.0, qcent, xcent;
  int npoint = 100, nsamp = 1, ncent = 0, neval, i, err;
  double *xinit = (double *)malloc(sizeof(double) * (size_t)(ninit > 0 ? ninit : 1));
  if (!xinit) return 1006;
  for (i = 0; i < ninit; i++) xinit[i] = *xl + (i + 1.0) * (*xr - *xl) / (ninit + 1.0);
  err = arms(xinit, ninit, xl, xr, myfunc, mydata, &convex, npoint, dometrop, xprev, xsamp, nsamp,
             &qcent, &xcent, ncent, &neval);
  free(xinit);
  return err;
}

/*
 * sslice.c -- shrinking-bracket slice sampler on a unimodal log-density (include/psample.h).
 *
 * Host control loop, behaviour-identical to the reference's lib/sslice.c:33-80: the same draws
 * from drand48() in the same order, at most TOOMANY-1 proposals per sweep, the bracket shrunk
 * towards the point the sweep started from, 1 returned on any failure.  Each call of `post`
 * is a device evaluation when the caller passes aterms/bterms.
 */
#include <math.h>
#include <stdio.h>

#include "../../include/psample.h"

#define TOOMANY 200 /* lib/sslice.c:24 */

int SliceSimple(double *xp, double (*post)(double, void *), double *bounds, rngp_t rng, int loops,
                void *pars) {
  double here = *xp;
  (void)rng;
  if (here < bounds[0] || here > bounds[1]) {
    fprintf(stderr, "SliceSimple: input value %lf outside bounds [%lg,%lg]\n", here, bounds[0],
            bounds[1]);
    return 1;
  }
  for (; loops > 0; loops--) {
    double lo = bounds[0], hi = bounds[1];
    /* slice level: log-density at the current point plus log of a uniform (lib/sslice.c:49-54) */
    double level = post(here, pars);
    int tries, hit = 0;
    level += log(rng_unit(rng));
    for (tries = 1; tries < TOOMANY; tries++) {
      here = lo + rng_unit(rng) * (hi - lo);
      if (post(here, pars) > level) {
        *xp = here;
        hit = 1;
        break;
      }
      /* rejected: pull the bracket in on the side of the last accepted point (lib/sslice.c:66-69) */
      if (here < *xp)
        lo = here;
      else
        hi = here;
    }
    if (!hit) {
      fprintf(stderr, "SliceSimple: giving up after %d tries, range=[%lg,%lg]\n", TOOMANY, lo, hi);
      return 1;
    }
  }
  return 0;
}

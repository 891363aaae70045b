/*
 * srng.h -- the RNG shim the samplers are written against; drop-in for the reference's
 * lib/srng.h:19-34.  The generator is glibc's process-global rand48 family, so (as in the
 * reference, lib/srng.h:4-6) it must not be used from several threads at once, and the `rng`
 * handle is ignored.
 */
#ifndef STB_AMD_SRNG_H
#define STB_AMD_SRNG_H
#ifndef __RNG_H
#define __RNG_H
#endif
#include <stdlib.h>
#include <time.h>
#ifdef __cplusplus
extern "C" {
#endif
double gsl_rng_gaussian_ziggurat(const double sigma); /* lib/srng.h:19 */
double gsl_rng_beta(const double a, const double b);  /* lib/srng.h:20 */
double gsl_rng_gamma(const double a);                 /* lib/srng.h:21 */
typedef void *rngp_t;                                 /* lib/srng.h:23 */
/* lib/srng.h:28-34 */
#define rng_seed(rng, seed) srand48(seed);
#define rng_time(rng, seed)  \
  {                          \
    *(seed) = time(NULL);    \
    srand48(*(seed));        \
  }
#define rng_unit(rng) drand48()
#define rng_beta(rng, a, b) gsl_rng_beta(a, b)
#define rng_gamma(rng, a) gsl_rng_gamma(a)
#define rng_gaussian(rng, a) gsl_rng_gaussian_ziggurat(a)
#define rng_free(rng)
#ifdef __cplusplus
}
#endif
#endif

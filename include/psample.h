/*
 * psample.h -- hyper-parameter samplers for the Pitman-Yor discount a and concentration b;
 * drop-in for the reference's lib/psample.h (:37 PSAMPLE_ARS, :58-59 B bounds, :64/:68 count types,
 * :79-84 sampleb, :89-94 A bounds, :104-110 samplea).
 *
 * Control flow (bracketing, ARMS / slice, RNG draws) runs on the host exactly as in the reference;
 * every evaluation of the log-posterior -- aterms (lib/samplea.c:46-83: table rebuild for the trial
 * discount + S_S gather-sum + restaurant terms) and bterms (lib/sampleb.c:33-41) -- runs on the
 * GPU through include/stb_hip.h.
 *
 * samplea2 (lib/psample.h:112-117, the S-free variant behind the reference's SAMPLEA_M switch,
 * lib/psample.h:22-30) is declared under the same switch and always present in the library.
 */
#ifndef STB_AMD_PSAMPLE_H
#define STB_AMD_PSAMPLE_H
#ifndef __PSAMPLE_H
#define __PSAMPLE_H
#endif

#include "stable.h"
#include "srng.h"

#ifdef __cplusplus
extern "C" {
#endif

/* lib/psample.h:37: adaptive rejection sampling is the default; undefine for the slice sampler */
#define PSAMPLE_ARS
#include "arms.h"

/* lib/psample.h:45-51, lib/sslice.c:33-80.  Always exported (the reference builds sslice.c into
 * the library unconditionally, lib/Makefile:10) although its prototype there is hidden under
 * PSAMPLE_ARS.  Returns non-zero on error. */
int SliceSimple(double *xp, double (*post)(double, void *), double *bounds, rngp_t rng, int loops,
                void *pars);

/* lib/psample.h:58-59 */
#define B_MIN 0.01
#define B_MAX 2000

typedef uint32_t scnt_int;  /* lib/psample.h:64: customer counts */
typedef uint16_t stcnt_int; /* lib/psample.h:68: table counts (t <= 65535) */

/* lib/psample.h:79-84, lib/sampleb.c:79-159: one MCMC step for b given Gamma(shape,scale) prior,
 * per-restaurant totals N[i], T[i] and discount apar */
double sampleb(double b_in, int I, double shape, double scale, scnt_int *N, scnt_int *T,
               double apar, rngp_t rng, int loops, int verbose);

/* lib/psample.h:89-94 */
#define A_MIN 0.01
#define A_MAX 0.98
#define SQUEEZEA 0.2

/* lib/psample.h:104-110, lib/samplea.c:155-225: one MCMC step for a.  n[i][k], t[i][k] for
 * k < K[i]; T[i] = sum_k t[i][k]; bpar[i] the concentration of restaurant i; counts may instead
 * come from getval(&n,&t,i,k).  Builds (and frees) its own table. */
double samplea(double apar, int I, int *K, scnt_int *T, scnt_int **n, stcnt_int **t,
               void (*getval)(scnt_int *n, stcnt_int *t, unsigned i, unsigned k), double *bpar,
               rngp_t rng, int loops, int verbose);

/* lib/psample.h:22-30, :112-117; lib/samplea.c:244-340.  The S-free discount step: with the caller's
 * table S (built for the CURRENT discount apar) sample, for every (n,t) with 1 < t < n, how the n
 * customers split over the t tables, then draw a from the posterior given those table sizes, which
 * needs no Stirling table (aterms2, lib/samplea.c:85-150).  Defined in the library whatever the
 * switch; declared under it as in the reference.  Differences from the reference, all in cases it
 * leaves undefined: ARMS gets the posterior's data (the reference passes NULL, lib/samplea.c:325-326,
 * and crashes); getval() is honoured in the sampling loops too; pairs with t = 0 or t > n contribute
 * nothing; the partition buffer is freed. */
#ifdef SAMPLEA_M
double samplea2(double apar, stable_t *S, int I, int *K, scnt_int *T, scnt_int **n, stcnt_int **t,
                void (*getval)(scnt_int *n, stcnt_int *t, unsigned i, unsigned k), double *bpar,
                rngp_t rng, int loops, int verbose);
#endif

#ifdef __cplusplus
}
#endif
#endif

/*
 * pyp_resample.c -- this repo's own end-to-end driver for the path libstb_amd accelerates
 * (the reference's counterpart is the hand-run test/demo.c).
 *
 *   1. synthesise seating data from a hierarchical Chinese-restaurant process: J restaurants share a
 *      base distribution over DISHES dishes; restaurant j seats NCUST customers under PYP(a0, b0);
 *   2. run the table-indicator Gibbs sampler on the table counts t[j][i] given the customer counts
 *      n[j][i], reading V^n_m = S^n_m / S^n_{m-1} from the library's ratio table (S_V);
 *   3. every few sweeps resample the concentration b (sampleb) and the discount a (samplea) and
 *      rebuild the table for the new discount (S_remake);
 *   4. optionally (-g D) also evaluate the discount's log-posterior on a D-point grid in one batched
 *      device call (stb_groups_aterms) and report its mode next to the sampled value;
 *   5. with -G k as well: the same grid sharded over k group sets, set s on GPU s % (number of GPUs), driven by
 *      THIS one host thread (stb_set_device + stb_groups_aterms_async on every set, then stb_groups_wait on
 *      every set: INTEGRATION.md section 5, pattern (a)) -- the discount axis of SURVEY 8e without MPI.
 *
 * All table builds and every log-posterior evaluation run on the GPU through libstb_amd.so; this file
 * only uses the public headers.  Usage: pyp_resample [-J 3] [-n 2000] [-a 0.5] [-b 10] [-c 60]
 *                                                    [-g 64] [-G 2] [-s seed]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "psample.h"
#include "stable.h"
#include "stb_hip.h"
#include "yaps.h"

#define DISHES 50

int main(int argc, char **argv) {
  int J = 3, ncust = 2000, cycles = 60, grid = 0, nsets = 0, c, j, i, it;
  double a0 = 0.5, b0 = 10.0;
  long seed = 12345;
  while ((c = getopt(argc, argv, "J:n:a:b:c:g:G:s:")) >= 0) {
    if (c == 'J') J = atoi(optarg);
    else if (c == 'n') ncust = atoi(optarg);
    else if (c == 'a') a0 = atof(optarg);
    else if (c == 'b') b0 = atof(optarg);
    else if (c == 'c') cycles = atoi(optarg);
    else if (c == 'g') grid = atoi(optarg);
    else if (c == 'G') nsets = atoi(optarg);
    else if (c == 's') seed = atol(optarg);
    else return 2;
  }
  srand48(seed);
  srand((unsigned)seed);

  /* ---- 1. data: seat customers, remember per (restaurant, dish) customers n and tables t ---- */
  scnt_int **n = malloc(sizeof(*n) * J), *N = calloc(J, sizeof(*N)), *T = calloc(J, sizeof(*T));
  stcnt_int **t = malloc(sizeof(*t) * J);
  int *K = malloc(sizeof(int) * J);
  int *dish_of = malloc(sizeof(int) * (size_t)J * ncust); /* customer -> dish, for the Gibbs sweep */
  unsigned maxn = 1;
  for (j = 0; j < J; j++) {
    /* tables of this restaurant: size and dish */
    int *tsize = calloc(ncust, sizeof(int)), *tdish = calloc(ncust, sizeof(int)), ntab = 0, cst;
    n[j] = calloc(DISHES, sizeof(scnt_int));
    t[j] = calloc(DISHES, sizeof(stcnt_int));
    K[j] = DISHES;
    for (cst = 0; cst < ncust; cst++) {
      double pnew = (b0 + a0 * ntab) / (b0 + cst);
      int tab;
      if (ntab == 0 || rng_unit(0) < pnew) {
        tab = ntab++;
        tdish[tab] = (int)(rng_unit(0) * DISHES) % DISHES; /* uniform base distribution */
        t[j][tdish[tab]]++;
      } else {
        /* existing table k with probability proportional to (size_k - a0) */
        double u = rng_unit(0) * (cst - a0 * ntab);
        for (tab = 0; tab < ntab - 1; tab++) {
          u -= tsize[tab] - a0;
          if (u < 0) break;
        }
      }
      tsize[tab]++;
      n[j][tdish[tab]]++;
      dish_of[(size_t)j * ncust + cst] = tdish[tab];
    }
    N[j] = ncust;
    T[j] = ntab;
    for (i = 0; i < DISHES; i++)
      if (n[j][i] >= maxn) maxn = n[j][i] + 1;
    free(tsize);
    free(tdish);
  }
  printf("data: %d restaurants x %d customers, true a=%.3f b=%.2f, tables:", J, ncust, a0, b0);
  for (j = 0; j < J; j++) printf(" %u", T[j]);
  printf("\n");

  /* ---- 2./3. Gibbs on table counts with periodic hyper-parameter resampling ---- */
  double a = 0.3, b = 5.0, asum = 0, bsum = 0;
  int kept = 0;
  unsigned maxt = maxn < 400 ? maxn : 400;
  stable_t *S = S_make(maxn, maxt, maxn, maxn, a, S_STABLE | S_UVTABLE);
  if (!S) yaps_quit("S_make failed: %s\n", stb_last_error());
  double *bvec = malloc(sizeof(double) * J);
  for (it = 0; it < cycles; it++) {
    for (j = 0; j < J; j++) {
      int cst;
      for (cst = 0; cst < ncust; cst++) {
        i = dish_of[(size_t)j * ncust + cst];
        unsigned nn = n[j][i];
        if (nn == 1) continue; /* a single customer always opens the table */
        /* remove this customer's indicator: it heads a table with probability (t-1)/(n-1) */
        if (t[j][i] > 1 && (nn - 1) * rng_unit(0) < (double)(t[j][i] - 1)) {
          t[j][i]--;
          T[j]--;
        }
        /* odds of opening a table: H (b + a T) t/(n-t+1) V^n_{t+1}, V from the ratio table */
        double odds = (1.0 / DISHES) * (b + T[j] * a) * t[j][i] / (nn - t[j][i] + 1.0) * S_V(S, nn, t[j][i] + 1);
        if (rng_unit(0) < odds / (odds + 1.0)) {
          t[j][i]++;
          T[j]++;
        }
      }
    }
    if (it % 3 == 2) {
      b = sampleb(b, J, 1.1, 20.0, N, T, a, 0, 1, 0);
      for (j = 0; j < J; j++) bvec[j] = b;
      a = samplea(a, J, K, T, n, t, NULL, bvec, 0, 1, 0);
      if (S_remake(S, a)) yaps_quit("S_remake failed\n");
      if (it >= cycles / 2) {
        asum += a;
        bsum += b;
        kept++;
      }
    }
  }
  printf("posterior means after %d sweeps: a=%.3f b=%.2f; tables:", cycles, kept ? asum / kept : a, kept ? bsum / kept : b);
  for (j = 0; j < J; j++) printf(" %u", T[j]);
  printf("\n");

  /* ---- 4. batched grid evaluation of the discount posterior ---- */
  if (grid > 1 && grid <= 64) {
    size_t G = (size_t)J * DISHES, g = 0;
    scnt_int *nf = malloc(sizeof(*nf) * G);
    stcnt_int *tf = malloc(sizeof(*tf) * G);
    unsigned mt = 1;
    double x[64], lp[64];
    int d, best = 0;
    for (j = 0; j < J; j++)
      for (i = 0; i < DISHES; i++, g++) {
        nf[g] = n[j][i];
        tf[g] = t[j][i];
        if (tf[g] >= mt) mt = tf[g] + 1;
      }
    unsigned M = mt < 10 ? 10 : mt, Nb = maxn < M ? M : maxn;
    stb_groups_t *gs = stb_groups_create(J, K, T, nf, tf, bvec, Nb, M, grid);
    if (!gs) yaps_quit("stb_groups_create: %s\n", stb_last_error());
    for (d = 0; d < grid; d++) x[d] = 0.02 + 0.95 * (d + 0.5) / grid;
    if (stb_groups_aterms(gs, x, grid, lp)) yaps_quit("grid evaluation: %s\n", stb_last_error());
    for (d = 1; d < grid; d++)
      if (lp[d] > lp[best]) best = d;
    printf("grid of %d discounts in one batched call: posterior mode at a=%.3f (log-posterior %.3f)\n", grid,
           x[best], lp[best]);
    stb_groups_free(gs);
    /* ---- 5. the same grid, a contiguous block of it per group set, the sets spread over the GPUs ---- */
    if (nsets >= 1 && nsets <= 8 && nsets <= grid) {
      const int ndev = stb_device_count(), home = stb_get_device();
      stb_groups_t *set[8];
      double lp2[64];
      int lo[9], s, worst = 0;
      for (s = 0; s <= nsets; s++) lo[s] = (int)((long)grid * s / nsets);
      for (s = 0; s < nsets; s++) {
        if (stb_set_device(s % (ndev > 0 ? ndev : 1))) yaps_quit("stb_set_device: %s\n", stb_last_error());
        set[s] = stb_groups_create(J, K, T, nf, tf, bvec, Nb, M, lo[s + 1] - lo[s]); /* (lives on that GPU from now on) */
        if (!set[s]) yaps_quit("stb_groups_create: %s\n", stb_last_error());
      }
      stb_set_device(home);
      /* one call: contiguous blocks of the grid queued on every set, then all waited for -- the GPUs (or the sets of one
         GPU) work side by side (what it does inside: stb_groups_aterms_async on each, then stb_groups_wait on each) */
      if (stb_groups_aterms_multi(set, nsets, x, grid, lp2)) yaps_quit("grid evaluation: %s\n", stb_last_error());
      for (d = 0; d < grid; d++) {
        const double err = fabs(lp2[d] - lp[d]) / (fabs(lp[d]) > 1 ? fabs(lp[d]) : 1);
        if (err > 1e-12) worst++;
      }
      best = 0;
      for (d = 1; d < grid; d++)
        if (lp2[d] > lp2[best]) best = d;
      printf("the same grid over %d group sets on %d GPU(s), one host thread: posterior mode at a=%.3f (log-posterior %.3f), %d of %d values differ from the single call by more than 1e-12\n",
             nsets, ndev < nsets ? ndev : nsets, x[best], lp2[best], worst, grid);
      for (s = 0; s < nsets; s++) stb_groups_free(set[s]);
    }
    free(nf);
    free(tf);
  }
  S_free(S);
  return 0;
}

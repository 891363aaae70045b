/*
 * sapprox.c -- closed-form log S^n_{m,a} and its a-derivative for m <= 4 (include/sapprox.h).
 *
 * Follows the reference's lib/sapprox.c:28-71 and :76-114 as built there (LS_NOPOLYGAMMA defined
 * in lib/digamma.h:25, so the small-a polygamma branch does not exist) and Radford Neal's digamma
 * series of lib/digamma.c:34-48.  `a` is a float and every `n - k*a` is formed in single
 * precision before widening, as the reference's C expressions do; the functions therefore agree
 * with the recurrence only for dyadic a with m*a < 1.  O(1) host work; not on
 * the device path, no in-tree callers -- kept for link compatibility and as a known-answer check.
 */
#include <math.h>

#include "../../include/sapprox.h"

double digammaRN(double x) {
  /* recurrence up to x > 5, then the asymptotic series in 1/x^2 (lib/digamma.c:38-47) */
  double shift = 0, f, t;
  while (x <= 5) {
    shift -= 1 / x;
    x += 1;
  }
  f = 1 / (x * x);
  t = f * (-1 / 12.0 +
           f * (1 / 120.0 +
                f * (-1 / 252.0 +
                     f * (1 / 240.0 +
                          f * (-1 / 132.0 + f * (691 / 32760.0 + f * (-1 / 12.0 + f * 3617 / 8160.0)))))));
  return shift + log(x) - 0.5 / x + t;
}

/* log Gamma(n-k a)/Gamma(1-k a), argument arithmetic in float like the reference */
static double lgr(int n, int k, float a) { return lgamma(n - k * a) - lgamma(1 - k * a); }

double S_approx(int n, int m, float a) {
  if (n == m) return 0.0;
  if (n < m) return -HUGE_VAL;
  if (m == 1) return lgamma(n - a) - lgamma(1 - a);
  if (m == 2) {
    double ga = lgr(n, 1, a), g2a = lgamma(n - 2 * a) - lgamma(1 - 2 * a);
    return g2a - log(a) + log(exp(ga - g2a) - 1.0); /* lib/sapprox.c:52-55 */
  }
  if (m == 3) {
    double ga = lgr(n, 1, a), g2a = lgamma(n - 2 * a) - lgamma(1 - 2 * a);
    double g3a = lgamma(n - 3 * a) - lgamma(1 - 3 * a);
    return g3a - 2 * log(a) - log(2.0) + log(exp(ga - g3a) - 2 * exp(g2a - g3a) + 1.0); /* :56-61 */
  }
  if (m == 4) {
    double ga = lgr(n, 1, a), g2a = lgamma(n - 2 * a) - lgamma(1 - 2 * a);
    double g3a = lgamma(n - 3 * a) - lgamma(1 - 3 * a);
    double g4a = lgamma(n - 4 * a) - lgamma(1 - 4 * a);
    return g4a - 3 * log(a) - log(6.0) +
           log(exp(ga - g4a) - 3 * exp(g2a - g4a) + 3 * exp(g3a - g4a) - 1.0); /* :62-69 */
  }
  return -HUGE_VAL;
}

double S_approx_da(int n, int m, float a) {
  double snm;
  if (n == m) return 0.0;
  if (n < m) return -HUGE_VAL;
  if (m == 1) return -(digammaRN(n - a) - digammaRN(1 - a)); /* lib/sapprox.c:83-85 */
  snm = S_approx(n, m, a);
  if (m == 2) {
    /* lib/sapprox.c:87-92 */
    double ga = lgr(n, 1, a), g2a = lgamma(n - 2 * a) - lgamma(1 - 2 * a);
    double dga = -(digammaRN(n - a) - digammaRN(1 - a));
    double dg2a = -2.0 * (digammaRN(n - 2 * a) - digammaRN(1 - 2 * a));
    return (exp(ga - snm) * dga - exp(g2a - snm) * dg2a - 1) / a;
  }
  if (m == 3) {
    /* lib/sapprox.c:93-100 */
    double ga = lgr(n, 1, a), g2a = lgamma(n - 2 * a) - lgamma(1 - 2 * a);
    double g3a = lgamma(n - 3 * a) - lgamma(1 - 3 * a);
    double dga = -(digammaRN(n - a) - digammaRN(1 - a));
    double dg2a = -2 * (digammaRN(n - 2 * a) - digammaRN(1 - 2 * a));
    double dg3a = -3 * (digammaRN(n - 3 * a) - digammaRN(1 - 3 * a));
    return -2 / a + (exp(ga - snm) * dga - 2 * exp(g2a - snm) * dg2a + exp(g3a - snm) * dg3a) / 2 / a / a;
  }
  if (m == 4) {
    /* lib/sapprox.c:101-112 */
    double ga = lgr(n, 1, a), g2a = lgamma(n - 2 * a) - lgamma(1 - 2 * a);
    double g3a = lgamma(n - 3 * a) - lgamma(1 - 3 * a);
    double g4a = lgamma(n - 4 * a) - lgamma(1 - 4 * a);
    double dga = -(digammaRN(n - a) - digammaRN(1 - a));
    double dg2a = -2 * (digammaRN(n - 2 * a) - digammaRN(1 - 2 * a));
    double dg3a = -3 * (digammaRN(n - 3 * a) - digammaRN(1 - 3 * a));
    double dg4a = -4 * (digammaRN(n - 4 * a) - digammaRN(1 - 4 * a));
    return -3 / a +
           (exp(ga - snm) * dga - 3 * exp(g2a - snm) * dg2a + 3 * exp(g3a - snm) * dg3a -
            exp(g4a - snm) * dg4a) / 3 / a / a / a;
  }
  return -HUGE_VAL;
}

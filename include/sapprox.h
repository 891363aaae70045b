/*
 * sapprox.h -- closed-form log S^n_{m,a} for m <= 4; drop-in for the reference's
 * lib/sapprox.h:24,29 (lib/sapprox.c:28-71, :76-114, LS_NOPOLYGAMMA build).  `a` is a float and
 * n - k*a is formed in single precision, as in the reference.  Returns -HUGE_VAL for m > 4.
 * Host-only: O(1) work, no in-tree callers; kept for link compatibility and as a known-answer check.
 */
#ifndef STB_AMD_SAPPROX_H
#define STB_AMD_SAPPROX_H
#ifdef __cplusplus
extern "C" {
#endif
double S_approx(int n, int m, float a);    /* lib/sapprox.h:24 */
double S_approx_da(int n, int m, float a); /* lib/sapprox.h:29 */
double digammaRN(double x);                /* lib/digamma.h:40, used by S_approx_da */
#ifdef __cplusplus
}
#endif
#endif

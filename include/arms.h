/*
 * arms.h -- adaptive rejection (Metropolis) sampling; drop-in for the reference's lib/arms.h:3-15.
 *
 * Same entry points, argument order, return codes (0, 1001..1008, 2000, 2001) and the same use of
 * libc rand() as lib/arms.c, so a given srand() seed yields the same draw and the same sequence of
 * log-density evaluations.  The implementation (libstb_amd/csrc/arms.c) is new: the envelope is a
 * sorted array of knots rather than a pointer-linked list.
 */
#ifndef STB_AMD_ARMS_H
#define STB_AMD_ARMS_H
#ifdef __cplusplus
extern "C" {
#endif
/* lib/arms.h:3-5, lib/arms.c:98-125: ninit evenly spaced starting abscissae inside (*xl,*xr),
 * at most 100 envelope points, one draw into *xsamp */
int arms_simple(int ninit, double *xl, double *xr, double (*myfunc)(double x, void *mydata),
                void *mydata, int dometrop, double *xprev, double *xsamp);
/* lib/arms.h:7-11, lib/arms.c:129-264 */
int arms(double *xinit, int ninit, double *xl, double *xr,
         double (*myfunc)(double x, void *mydata), void *mydata, double *convex, int npoint,
         int dometrop, double *xprev, double *xsamp, int nsamp, double *qcent, double *xcent,
         int ncent, int *neval);
/* lib/arms.h:13, lib/arms.c:835-845: exp(y - y0 + YCEIL), 0 below -2*YCEIL */
double expshift(double y, double y0);
#define YCEIL 50. /* lib/arms.h:15 */
#ifdef __cplusplus
}
#endif
#endif

/*
 * randist.c -- the three variate generators the samplers draw from (include/srng.h):
 * standard normal by a 128-level ziggurat, Gamma(a,1) by Marsaglia & Tsang's squeeze, Beta as a
 * ratio of Gammas.  Host code: the stream is glibc's process-global rand48 and is inherently
 * sequential (reference lib/srng.h:4-6), and sampleb (lib/sampleb.c:94,108,111) consumes it.
 *
 * Stream-identical to the reference's lib/gslrandist.c (uniform_int :53-72, gaussian :194-233,
 * gamma :235-272, beta :274-282): same draws from lrand48()/drand48() in the same order and the
 * same floating-point expressions, so a given srand48() seed yields the same variates.
 *
 * The ziggurat tables are not transcribed: they are rebuilt at first use from their definition.
 * With f(x) = exp(-x^2/2), R the right-most level and v = R f(R) + f(R)/R the common strip area
 * (the tail replaced by an exponential wedge), the levels satisfy x_127 = R,
 * f(x_i) = f(x_{i+1}) + v / x_{i+1}, and R is THE value for which the recursion closes at
 * f(x_0) = 1.  The published tables hold f(x_i), 2^-24 x_{i+1} and floor(2^24 x_i / x_{i+1})
 * printed to 12 significant digits, so computing them in long double and rounding the same way
 * reproduces them exactly (tests/test_host_logic.py checks all 384 numbers against the reference's
 * file when it is present, and the variate streams against golden draws everywhere).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/srng.h"

#define ZIG_R 3.44428647676 /* the 12-digit value the sampler itself uses (lib/gslrandist.c:75) */
/* the closing root of the level recursion, to long double precision */
#define ZIG_R_EXACT 3.44428647676128387941542569357292536L

static double zig_y[128];        /* f(x_i) */
static double zig_w[128];        /* 2^-24 * x_{i+1}  (i<127), 2^-24 * v/f(R) for the base strip */
static unsigned long zig_k[128]; /* floor(2^24 * x_i/x_{i+1}), base strip: floor(2^24 R f(R)/v) */
static int zig_ready = 0;

static double round12(long double x) {
  char buf[64];
  snprintf(buf, sizeof buf, "%.12Lg", x);
  return strtod(buf, NULL);
}

static void zig_build(void) {
  long double x[129], R = ZIG_R_EXACT, fR = expl(-R * R / 2), v = R * fR + fR / R;
  int i;
  x[127] = R;
  for (i = 126; i >= 1; i--) {
    long double fy = expl(-x[i + 1] * x[i + 1] / 2) + v / x[i + 1];
    x[i] = sqrtl(-2 * logl(fy));
  }
  x[0] = 0;
  for (i = 0; i < 128; i++) {
    zig_y[i] = round12(expl(-x[i] * x[i] / 2));
    if (i < 127) {
      zig_w[i] = round12(x[i + 1] / 16777216.0L);
      zig_k[i] = (unsigned long)floorl(x[i] / x[i + 1] * 16777216.0L);
    } else {
      zig_w[i] = round12(v / fR / 16777216.0L);
      zig_k[i] = (unsigned long)floorl(R * fR / v * 16777216.0L);
    }
  }
  zig_ready = 1;
}

/* debug/test accessor: which = 0 y, 1 w, 2 k */
double stb_zig_table(int which, int i) {
  if (!zig_ready) zig_build();
  if (i < 0 || i > 127) return 0;
  return which == 0 ? zig_y[i] : which == 1 ? zig_w[i] : (double)zig_k[i];
}

static double unit_pos(void) {
  /* lib/gslrandist.c:53-58 */
  double u = rng_unit(0);
  while (u == 0) u = rng_unit(0);
  return u;
}

static unsigned long below(unsigned long n) {
  /* lib/gslrandist.c:60-78: lrand48() has 31 bits while `range` is 2^30, so about half of the
   * candidates are rejected -- kept, because it decides how much of the stream is consumed */
  const unsigned long range = 1UL << 30;
  unsigned long scale, k;
  if (n > range || n == 0) return 0;
  scale = range / n;
  do {
    k = (unsigned long)lrand48() / scale;
  } while (k >= n);
  return k;
}

double gsl_rng_gaussian_ziggurat(const double sigma) {
  unsigned long i, j;
  int sign;
  double x, y;
  if (!zig_ready) zig_build();
  for (;;) {
    i = below(256);      /* strip, with the sign in bit 7 */
    j = below(16777216); /* 24-bit position inside it */
    sign = (i & 0x80) ? +1 : -1;
    i &= 0x7f;
    x = j * zig_w[i];
    if (j < zig_k[i]) break; /* inside the rectangle: no float test needed */
    if (i < 127) {
      double y0 = zig_y[i], y1 = zig_y[i + 1];
      double U1 = rng_unit(0);
      y = y1 + (y0 - y1) * U1;
    } else {
      /* exponential wedge over the tail */
      double U1 = 1.0 - rng_unit(0);
      double U2 = rng_unit(0);
      x = ZIG_R - log(U1) / ZIG_R;
      y = exp(-ZIG_R * (x - 0.5 * ZIG_R)) * U2;
    }
    if (y < exp(-0.5 * x * x)) break;
  }
  return sign * sigma * x;
}

double gsl_rng_gamma(const double a) {
  if (a < 1) {
    /* Gamma(a) = Gamma(a+1) U^{1/a}; the uniform is drawn first (lib/gslrandist.c:240-244) */
    double u = unit_pos();
    return gsl_rng_gamma(1.0 + a) * pow(u, 1.0 / a);
  }
  {
    double x, v, u;
    const double d = a - 1.0 / 3.0;
    const double c = (1.0 / 3.0) / sqrt(d);
    for (;;) {
      do {
        x = gsl_rng_gaussian_ziggurat(1.0);
        v = 1.0 + c * x;
      } while (v <= 0);
      v = v * v * v;
      u = unit_pos();
      if (u < 1 - 0.0331 * x * x * x * x) break;
      if (log(u) < 0.5 * x * x + d * (1 - v + log(v))) break;
    }
    return d * v;
  }
}

double gsl_rng_beta(const double a, const double b) {
  double x1 = gsl_rng_gamma(a);
  double x2 = gsl_rng_gamma(b);
  return x1 / (x1 + x2);
}

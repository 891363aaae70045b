/* sampler_trace.c -- record of the log-posterior evaluations of the most recent samplea/sampleb
 * call (abscissa, value) and ARMS' return code; see include/stb_hip.h. */
#include "sampler_trace.h"

#define CAP 1024
/* (per calling thread: samplea / sampleb of two threads keep separate records) */
static _Thread_local double xs[CAP], ys[CAP];
static _Thread_local int count = 0, code = 0;

void stb_trace_reset(void) {
  count = 0;
  code = 0;
}
void stb_trace_add(double x, double y) {
  if (count < CAP) {
    xs[count] = x;
    ys[count] = y;
  }
  count++;
}
void stb_trace_code(int c) { code = c; }

int stb_sampler_trace_count(void) { return count; }
int stb_sampler_trace_code(void) { return code; }
int stb_sampler_trace_get(int i, double *x, double *y) {
  if (i < 0 || i >= count || i >= CAP) return 1;
  *x = xs[i];
  *y = ys[i];
  return 0;
}

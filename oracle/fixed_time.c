/* oracle/fixed_time.c -- TEST INFRASTRUCTURE ONLY.  The reference's test/demo.c re-seeds its Gibbs
 * sampler from time(NULL) (test/demo.c:343-344); linking this into the two demo builds pins that
 * seed so the reference-library run and the libstb_amd run can be compared line by line. */
#include <time.h>
time_t time(time_t *t) {
  if (t) *t = (time_t)424242;
  return (time_t)424242;
}

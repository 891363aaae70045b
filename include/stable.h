/*
 * stable.h -- libstb_amd's drop-in for the reference's lib/stable.h (libstb 1.8).
 *
 * Same names, argument meaning, return conventions and struct field order as the reference
 * (wbuntine/libstb lib/stable.h:38-44 flags, :62-113 stable_t, :128-190 functions), so a caller
 * such as samplea.c / demo.c / hca compiles and links against libstb_amd.so unchanged.  What
 * differs is behind the interface: S_make / S_remake / growth fill the table on an MI355X
 * (kernels in libstb_amd/csrc/fill_*.hip) and S_S / S_V / S_U / S_UV read a host mirror that is
 * copied from the device on demand, 128 rows at a time.  There is no CPU fill: without a HIP device S_make returns NULL
 * after reporting through yaps_message().
 *
 * Each declaration cites the reference line it replaces.
 */
#ifndef STB_AMD_STABLE_H
#define STB_AMD_STABLE_H
#ifndef __STABLE_H
#define __STABLE_H /* the reference's include guard, so mixed include orders stay single */
#endif

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flag bits, fixed for the life of a table -- reference lib/stable.h:38-44 */
#define S_STABLE 1       /* keep the log S^n_{m,a} table */
#define S_UVTABLE 2      /* keep the V (ratio) table, which also serves U */
#define S_FLOAT 4        /* store table values as float (arithmetic stays double) */
#define S_VERBOSE 8      /* one report line to stderr after each (re)build */
#define S_QUITONBOUND 16 /* overrunning maxN/maxM is fatal instead of log(0) / 0 */
#define S_THREADS 32     /* serialise growth behind a mutex (needs S_USE_THREADS) */
#define S_ASYMPT 64      /* beyond maxN answer with the asymptotic formula */

/* reference lib/stable.h:49-55 */
#ifdef H_THREADS
#define S_USE_THREADS
#endif
#ifdef S_USE_THREADS
#include <pthread.h>
#endif

/*
 * reference lib/stable.h:62-113.  Field order and types are the reference's; `impl` is new and
 * sits after `tag`, before the optional mutex, so its offset does not depend on S_USE_THREADS
 * (callers and the library may be compiled with different settings, as in the reference's own
 * lib/ vs test/ Makefiles).  Treat every field as private.
 */
typedef struct stable_s {
  unsigned maxM, maxN;   /* inclusive bounds, never change */
  unsigned usedM, usedN; /* inclusive bounds currently filled; grow on demand */
  unsigned startM;       /* usedM at creation (reference: extent of its first malloc block) */
  double **S;            /* S[n-3][m-2] = log S^n_{m,a}; row pointers into the host mirror */
  float **Sf;            /* same, float storage (S_FLOAT) */
  double *SfrontN;       /* S_FLOAT: last row in double,   m = 2..usedM   */
  double *SfrontM;       /* S_FLOAT: last column in double, n > usedM     */
  double **V;            /* V[n-2][m-2] = S^n_m / S^n_{m-1} */
  float **Vf;
  double *VfrontN;
  double *VfrontM;
  unsigned usedN1;       /* entries allocated in S1 (>= usedN, grows lazily up to maxN) */
  double *S1;            /* S1[n-1] = log S^n_{1,a}; 0 marks "not yet computed" beyond usedN */
  double lga;            /* lgamma(1-a) */
  double a;              /* discount the tables were built for */
  uint32_t flags;
  uint32_t memalloced;   /* bytes held (host mirror + device), saturating */
  char *tag;             /* name used in messages; owned (strdup'd by S_tag) */
  void *impl;            /* device slabs, mirror bookkeeping -- libstb_amd private */
#ifdef S_USE_THREADS
  pthread_mutex_t mutex;
#endif
} stable_t;

/* reference lib/stable.h:128-130, lib/stable.c:110-312.  Arguments are clamped exactly as the
 * reference does (maxM>=10, maxN>=maxM, initM>=10, initN>=initM, initN<=maxN).  NULL on error. */
stable_t *S_make(unsigned initN, unsigned initM, unsigned maxN, unsigned maxM, double a,
                 uint32_t flags);
/* reference lib/stable.h:131, lib/stable.c:105-108 */
void S_tag(stable_t *S, char *tag);
/* reference lib/stable.h:137, lib/stable.c:549-554: refill for a new discount; non-zero on error */
int S_remake(stable_t *sp, double a);
/* reference lib/stable.h:139, lib/stable.c:980-1023; NULL-safe */
void S_free(stable_t *sp);

/* reference lib/stable.h:153, lib/stable.c:941-974: log S^n_{m,a}; grows the table inside the max
 * bounds; -HUGE_VAL when out of bounds or no S table */
double S_S(stable_t *sp, unsigned n, unsigned m);
/* reference lib/stable.h:161, lib/stable.c:822-873: log Gamma(n-a)/Gamma(1-a), cached */
double S_S1(stable_t *sp, unsigned n);
/* reference lib/stable.h:168, lib/stable.c:1057-1084 */
double S_asympt(stable_t *sp, unsigned n, unsigned m);
/* reference lib/stable.h:175, lib/stable.c:875-883: U^n_m = S^{n+1}_m / S^n_m */
double S_U(stable_t *sp, unsigned n, unsigned m);
/* reference lib/stable.h:179, lib/stable.c:885-897: U*V with one table read */
double S_UV(stable_t *sp, unsigned n, unsigned m);
/* reference lib/stable.h:186, lib/stable.c:900-939: V^n_m = S^n_m / S^n_{m-1}; 0 when illegal */
double S_V(stable_t *sp, unsigned n, unsigned m);
/* reference lib/stable.h:191, lib/stable.c:1025-1055 */
void S_report(stable_t *sp, FILE *fp);

/* reference lib/stable.h:193-197 */
#ifdef isfinite
#define ISFINITE(x) isfinite(x)
#else
#define ISFINITE(x) finite(x)
#endif

/* ---- additions (not in the reference) ----
 * The host mirror of the tables is filled lazily, block of 128 rows by block, the first time one of
 * the accessors above touches a row (see libstb_amd/csrc/stable_host.c).  A caller that reads the
 * row-pointer fields sp->S / sp->V directly (the reference's in-tree callers never do) must ask for
 * the whole mirror first -- after S_make, after every S_remake and after growth -- or run with
 * STB_MIRROR=eager in the environment, which makes every build do it.  Returns non-zero on a device
 * error. */
int stb_table_sync(stable_t *sp);
/* (round 6: a miss also sends the blocks BEHIND the touched one on their way, asynchronously, at least 16 MB of them --
 * STB_MIRROR_AHEAD_MB -- so that a caller who goes through the table waits for the first block of a region only;
 * STB_MIRROR=lazy is the old block-by-block behaviour) */
/* out[g] = S_S (which = 0) or S_V (which = 1) of (n[g], m[g]): the accessors above in a loop */
void stb_table_probe(stable_t *sp, int which, const unsigned *n, const unsigned *m, size_t G, double *out);
/* how many 128-row blocks of the S and V mirrors have been copied from the device so far */
void stb_table_mirrored(stable_t *sp, unsigned *s_blocks, unsigned *v_blocks);
/* bytes this table holds on the device (slabs + fill workspace) and on the host (mirror + vectors); their sum, capped
 * at 2^32 - 1, is what the memalloced field and S_report show.  An S_FLOAT table's slabs are float slabs only. */
void stb_table_bytes(stable_t *sp, unsigned long long *device_bytes, unsigned long long *host_bytes);

#ifdef __cplusplus
}
#endif
#endif

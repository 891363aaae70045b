/*
 * stable_host.c -- host side of the Stirling-table object behind include/stable.h.
 *
 * Mirrors the reference's lib/stable.c interface (S_make :110, S_remake :549, S_extend :564,
 * S_S1 :822, S_U :875, S_UV :885, S_V :900, S_S :941, S_free :980, S_report :1025,
 * S_asympt :1057) but not its construction: tables are filled on the GPU (stb_fill_S / stb_fill_V
 * in stb_kernels.hip) into one slab per table and copied into a pinned host mirror; the row
 * pointers S[n-3], V[n-2] of the public struct point into that mirror, so the accessors are the
 * same two loads as in the reference.  Growth recomputes the whole table for the new bounds (the
 * result equals the reference's incremental extension) under the reference's integer
 * growth policy.  There is no CPU fill anywhere in this file.
 *
 * Deliberate deviations from reference behaviour (each is a reference defect):
 *  - S_S1 beyond usedN returns the value for the n that was asked for (lib/stable.c:845-871
 *    overwrites n with the grown size first) and never returns holding the mutex (:842-844).
 *  - usedN1 is kept >= usedN after growth (lib/stable.c:809 passes the stale value, after which
 *    :856-857 would zero valid S1 entries).
 *  - S_make with neither table flag frees the struct it allocated (:131-132 leaks it).
 *  - m < 2 in S_V returns 0 instead of indexing before the row.
 *  - when one growth step is capped short of the requested column (new usedM <= old usedN,
 *    lib/stable.c:626-628) the reference reads past the end of the row; here growth repeats until
 *    the request is covered, so usedM can end up larger than the reference's in that case only.
 *  - S_FLOAT: the SfrontN/SfrontM/VfrontN/VfrontM double frontiers stay NULL; they exist in the
 *    reference only so that incremental extension loses no precision (lib/stable.c:389-449), and
 *    growth here recomputes from scratch in double.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define H_THREADS /* as the reference's lib/Makefile:7: the library always carries the mutex */
#include "../../include/stable.h"
#include "../../include/stb_hip.h"
#include "../../include/yaps.h"
#include "stb_layout.h"

/* one generation of host mirror; old generations are parked here when readers may still hold
 * row pointers into them (S_THREADS), mirroring the reference's allocate-copy-swap realloc
 * (lib/stable.c:56-81) */
typedef struct mirror_gen {
  void *slab;       /* pinned or malloc'd */
  void *rows;       /* row-pointer vector */
  int pinned;
  struct mirror_gen *next;
} mirror_gen;

typedef struct stb_impl {
  /* device */
  int dev; /* the GPU the table lives on (stb_get_device() at S_make time) */
  double *d_S, *d_V, *d_S1;
  float *d_Sf, *d_Vf; /* S_FLOAT: narrowed copies of the double slabs, same element offsets */
  void *d_ws;
  size_t ws_bytes;
  uint64_t d_S_elems, d_V_elems, d_S1_elems, d_Sf_elems, d_Vf_elems;
  /* host mirrors (current generation); elements are double, or float under S_FLOAT */
  void *h_S, *h_V;
  int h_S_pinned, h_V_pinned;
  uint64_t h_S_elems, h_V_elems;
  mirror_gen *retired;
  uint64_t bytes_host, bytes_dev;
} stb_impl;

static void account(stable_t *sp) {
  stb_impl *im = sp->impl;
  uint64_t tot = im->bytes_host + im->bytes_dev;
  sp->memalloced = tot > 0xffffffffull ? 0xffffffffu : (uint32_t)tot;
}

static void lock(stable_t *sp) {
  if (sp->flags & S_THREADS) pthread_mutex_lock(&sp->mutex);
}
static void unlock(stable_t *sp) {
  if (sp->flags & S_THREADS) pthread_mutex_unlock(&sp->mutex);
}

static void *host_slab(size_t bytes, int *pinned) {
  void *p = stb_host_malloc(bytes);
  *pinned = p != NULL;
  if (!p) p = malloc(bytes ? bytes : 1);
  return p;
}
static void host_slab_free(void *p, int pinned) {
  if (!p) return;
  if (pinned)
    stb_host_free(p);
  else
    free(p);
}

static void retire(stable_t *sp, void *slab, int pinned, void *rows) {
  stb_impl *im = sp->impl;
  if (!slab && !rows) return;
  if (sp->flags & S_THREADS) {
    mirror_gen *g = malloc(sizeof(*g));
    if (g) {
      g->slab = slab;
      g->rows = rows;
      g->pinned = pinned;
      g->next = im->retired;
      im->retired = g;
      return;
    }
  }
  host_slab_free(slab, pinned);
  free(rows);
}

/* host-side storage prepared for new bounds but not yet visible through the public struct */
typedef struct pending {
  void *hS, *hV;
  int hS_pinned, hV_pinned, hS_new, hV_new;
  uint64_t hS_elems, hV_elems;
  void **rowsS, **rowsV; /* double** or float** */
} pending;

/* bytes per stored table value: the reference's S_FLOAT keeps tables as float (lib/stable.h:31-33) */
static size_t esz(const stable_t *sp) { return (sp->flags & S_FLOAT) ? sizeof(float) : sizeof(double); }

static int devf_grow(stb_impl *im, float **slot, uint64_t *have, uint64_t want) {
  if (want <= *have) return 0;
  stb_device_free(*slot);
  im->bytes_dev -= *have * sizeof(float);
  *have = 0;
  *slot = stb_device_malloc(sizeof(float) * want);
  if (!*slot) return 1;
  *have = want;
  im->bytes_dev += want * sizeof(float);
  return 0;
}

static int dev_grow(stb_impl *im, double **slot, uint64_t *have, uint64_t want) {
  if (want <= *have) return 0;
  stb_device_free(*slot);
  im->bytes_dev -= *have * sizeof(double);
  *have = 0;
  *slot = stb_device_malloc(sizeof(double) * want);
  if (!*slot) return 1;
  *have = want;
  im->bytes_dev += want * sizeof(double);
  return 0;
}

static void pending_drop(pending *p) {
  if (p->hS_new) host_slab_free(p->hS, p->hS_pinned);
  if (p->hV_new) host_slab_free(p->hV, p->hV_pinned);
  free(p->rowsS);
  free(p->rowsV);
  memset(p, 0, sizeof(*p));
}

/* size device slabs and scratch for bounds (N,M) and prepare (but do not publish) host mirrors and
 * row-pointer vectors.  A mirror slab is reused when it is already large enough AND nobody can be
 * reading it (first build, or S_remake at unchanged bounds); growth always gets a fresh slab so
 * that concurrent readers keep seeing complete data (lib/stable.c:56-81, :539-545). */
static int provision(stable_t *sp, unsigned N, unsigned M, int growing, pending *p) {
  stb_impl *im = sp->impl;
  size_t ws = stb_fill_workspace_bytes(N, M, 1);
  unsigned n;
  memset(p, 0, sizeof(*p));
  if (ws > im->ws_bytes) {
    stb_device_free(im->d_ws);
    im->bytes_dev -= im->ws_bytes;
    im->ws_bytes = 0;
    im->d_ws = stb_device_malloc(ws);
    if (!im->d_ws) return 1;
    im->ws_bytes = ws;
    im->bytes_dev += ws;
  }
  if (dev_grow(im, &im->d_S1, &im->d_S1_elems, N)) return 1;
  if (sp->flags & S_STABLE) {
    uint64_t el = stb_table_elems(N, M);
    if (el < 2) el = 2;
    if (dev_grow(im, &im->d_S, &im->d_S_elems, el)) return 1;
    if ((sp->flags & S_FLOAT) && devf_grow(im, &im->d_Sf, &im->d_Sf_elems, el)) return 1;
    if (growing || el > im->h_S_elems) {
      p->hS = host_slab(esz(sp) * el, &p->hS_pinned);
      if (!p->hS) return 1;
      p->hS_new = 1;
      p->hS_elems = el;
    } else {
      p->hS = im->h_S;
      p->hS_pinned = im->h_S_pinned;
      p->hS_elems = im->h_S_elems;
    }
    p->rowsS = malloc(sizeof(void *) * (N > 3 ? N : 3));
    if (!p->rowsS) return 1;
    for (n = 3; n <= N; n++) p->rowsS[n - 3] = (char *)p->hS + esz(sp) * stb_row_offset(n, M);
  }
  if (sp->flags & S_UVTABLE) {
    uint64_t el = stb_vtable_elems(N, M);
    if (el < 2) el = 2;
    if (!(sp->flags & S_STABLE) && dev_grow(im, &im->d_S, &im->d_S_elems, stb_table_elems(N, 2) + 2))
      return 1; /* scratch for the S1-only fill */
    if (dev_grow(im, &im->d_V, &im->d_V_elems, el)) return 1;
    if ((sp->flags & S_FLOAT) && devf_grow(im, &im->d_Vf, &im->d_Vf_elems, el)) return 1;
    if (growing || el > im->h_V_elems) {
      p->hV = host_slab(esz(sp) * el, &p->hV_pinned);
      if (!p->hV) return 1;
      p->hV_new = 1;
      p->hV_elems = el;
    } else {
      p->hV = im->h_V;
      p->hV_pinned = im->h_V_pinned;
      p->hV_elems = im->h_V_elems;
    }
    p->rowsV = malloc(sizeof(void *) * (N > 2 ? N : 2));
    if (!p->rowsV) return 1;
    for (n = 2; n <= N; n++) p->rowsV[n - 2] = (char *)p->hV + esz(sp) * stb_vrow_offset(n, M);
  }
  return 0;
}

/* device fill for discount a at bounds (N,M), copied into the given mirrors; S1[0..N) refreshed */
static int build(stable_t *sp, double a, unsigned N, unsigned M, void *hS, void *hV) {
  stb_impl *im = sp->impl;
  if (sp->flags & S_STABLE) {
    if (stb_fill_S(&a, 1, N, M, im->d_S, im->d_S_elems, im->d_S1, N, im->d_ws, im->ws_bytes,
                   stb_default_variant(), NULL))
      return 1;
    if (sp->flags & S_FLOAT) {
      /* all arithmetic was done in double (as lib/stable.c:389-449 does through its frontier
       * vectors); only the stored values are narrowed, on the device, so the copy is half as big */
      if (stb_table_to_float(im->d_S, im->d_Sf, stb_table_elems(N, M), NULL)) return 1;
      if (stb_memcpy_d2h(hS, im->d_Sf, sizeof(float) * stb_table_elems(N, M), NULL)) return 1;
    } else if (stb_memcpy_d2h(hS, im->d_S, sizeof(double) * stb_table_elems(N, M), NULL))
      return 1;
    if (stb_memcpy_d2h(sp->S1, im->d_S1, sizeof(double) * N, NULL)) return 1;
    if (stb_fill_status()) return 1; /* (the copies above waited for the fill) */
  }
  if (sp->flags & S_UVTABLE) {
    if (stb_fill_V(&a, 1, N, M, im->d_V, im->d_V_elems, im->d_ws, im->ws_bytes, NULL)) return 1;
    if (sp->flags & S_FLOAT) {
      if (stb_table_to_float(im->d_V, im->d_Vf, stb_vtable_elems(N, M), NULL)) return 1;
      if (stb_memcpy_d2h(hV, im->d_Vf, sizeof(float) * stb_vtable_elems(N, M), NULL)) return 1;
    } else if (stb_memcpy_d2h(hV, im->d_V, sizeof(double) * stb_vtable_elems(N, M), NULL))
      return 1;
    if (stb_fill_status()) return 1; /* before any later fill reuses the workspace and its header */
  }
  if (!(sp->flags & S_STABLE)) {
    /* U/V-only tables still keep S1 (lib/stable.c:155, :337-348): take it from a width-2 S fill
     * on the device (N-2 cells) so that no table arithmetic runs on the host */
    if (stb_fill_S(&a, 1, N, 2, im->d_S, im->d_S_elems, im->d_S1, N, im->d_ws, im->ws_bytes,
                   stb_default_variant(), NULL))
      return 1;
    if (stb_memcpy_d2h(sp->S1, im->d_S1, sizeof(double) * N, NULL)) return 1;
    if (stb_fill_status()) return 1;
  }
  if (stb_stream_sync(NULL)) return 1;
  return 0;
}

/* make prepared storage visible: pointers first, bounds last (lib/stable.c:539-545) */
static void publish(stable_t *sp, pending *p, unsigned N, unsigned M) {
  stb_impl *im = sp->impl;
  if (sp->flags & S_STABLE) {
    void *oldrows;
    if (sp->flags & S_FLOAT) {
      oldrows = sp->Sf;
      sp->Sf = (float **)p->rowsS;
    } else {
      oldrows = sp->S;
      sp->S = (double **)p->rowsS;
    }
    retire(sp, NULL, 0, oldrows);
    if (p->hS_new) {
      retire(sp, im->h_S, im->h_S_pinned, NULL);
      im->bytes_host += (p->hS_elems - im->h_S_elems) * esz(sp);
      im->h_S = p->hS;
      im->h_S_pinned = p->hS_pinned;
      im->h_S_elems = p->hS_elems;
    }
  }
  if (sp->flags & S_UVTABLE) {
    void *oldrows;
    if (sp->flags & S_FLOAT) {
      oldrows = sp->Vf;
      sp->Vf = (float **)p->rowsV;
    } else {
      oldrows = sp->V;
      sp->V = (double **)p->rowsV;
    }
    retire(sp, NULL, 0, oldrows);
    if (p->hV_new) {
      retire(sp, im->h_V, im->h_V_pinned, NULL);
      im->bytes_host += (p->hV_elems - im->h_V_elems) * esz(sp);
      im->h_V = p->hV;
      im->h_V_pinned = p->hV_pinned;
      im->h_V_elems = p->hV_elems;
    }
  }
  sp->usedN = N;
  sp->usedM = M;
  account(sp);
}

stable_t *S_make(unsigned initN, unsigned initM, unsigned maxN, unsigned maxM, double a,
                 uint32_t flags) {
  stable_t *sp;
  stb_impl *im;
  /* lib/stable.c:118-129, including the :126-127 assignment of maxM */
  if (maxM < 10) maxM = 10;
  if (maxN < maxM) maxN = maxM;
  if (initM < 10) initM = 10;
  if (initN < initM) initN = initM;
  if (initN > maxN) initN = maxM;
  if (initN > maxN) initN = maxN;
  if (initN < initM) initM = initN; /* the quirk above can leave initN<initM: unusable, clamp */

  if ((flags & S_STABLE) == 0 && (flags & S_UVTABLE) == 0) return NULL; /* lib/stable.c:131-132 */
  if (!(a >= 0.0 && a < 1.0)) {
    yaps_message("S_make: discount %lf outside [0,1)\n", a);
    return NULL;
  }
  if (stb_device_count() < 1) {
    yaps_message("S_make: no HIP device available; libstb_amd has no CPU table fill\n");
    return NULL;
  }
  sp = calloc(1, sizeof(*sp));
  im = calloc(1, sizeof(*im));
  if (!sp || !im) {
    free(sp);
    free(im);
    return NULL;
  }
  sp->impl = im;
  im->dev = stb_get_device();
  if (im->dev < 0 || im->dev >= stb_device_count()) {
    yaps_message("S_make: device %d does not exist (stb_set_device / STB_DEVICE)\n", im->dev);
    free(sp);
    free(im);
    return NULL;
  }
  sp->flags = flags;
  sp->maxN = maxN;
  sp->maxM = maxM;
  sp->usedN = initN;
  sp->usedM = initM;
  sp->usedN1 = initN;
  sp->startM = initM;
  if (flags & S_THREADS) pthread_mutex_init(&sp->mutex, NULL);
  sp->S1 = malloc(sizeof(double) * initN);
  if (!sp->S1) {
    S_free(sp);
    return NULL;
  }
  im->bytes_host = sizeof(*sp) + sizeof(double) * initN;
  sp->a = a;
  sp->lga = lgamma(1.0 - a); /* lib/stable.c:329 */
  {
    pending p;
    const int prev_dev = stb_device_enter(im->dev);
    if (provision(sp, initN, initM, 0, &p) || build(sp, a, initN, initM, p.hS, p.hV)) {
      yaps_message("S_make: %s\n", stb_last_error());
      pending_drop(&p);
      stb_device_leave(prev_dev);
      S_free(sp);
      return NULL;
    }
    publish(sp, &p, initN, initM);
    stb_device_leave(prev_dev);
  }
  if (flags & S_VERBOSE) S_report(sp, stderr);
  return sp;
}

void S_tag(stable_t *sp, char *tag) {
  /* lib/stable.c:105-108 */
  sp->tag = malloc(strlen(tag) + 1);
  if (sp->tag) strcpy(sp->tag, tag);
}

int S_remake(stable_t *sp, double a) {
  unsigned n;
  if (!sp || !sp->impl) return 1;
  if (!(a >= 0.0 && a < 1.0)) return 1;
  {
    stb_impl *im = sp->impl;
    const int prev_dev = stb_device_enter(im->dev);
    const int rc = build(sp, a, sp->usedN, sp->usedM, im->h_S, im->h_V);
    stb_device_leave(prev_dev);
    if (rc) {
      /* the table keeps its old discount (the mirrors may be partly overwritten: remake again) */
      yaps_message("S_remake: %s\n", stb_last_error());
      return 1;
    }
  }
  sp->a = a;
  sp->lga = lgamma(1.0 - a); /* lib/stable.c:328-329 */
  /* the discount changed, so lazily cached S1 entries past usedN are void (lib/stable.c:350-353) */
  for (n = sp->usedN; n < sp->usedN1; n++) sp->S1[n] = 0;
  if (sp->flags & S_VERBOSE) S_report(sp, stderr);
  return 0;
}

/* lib/stable.c:564-630: the growth policy, integers only (kept separate so it can be unit-tested
 * without a device; exported for that purpose) */
void stb_extend_policy(unsigned usedN, unsigned usedM, unsigned maxN, unsigned maxM, int N, int M,
                       unsigned *newN, unsigned *newM) {
  unsigned n, m;
  N++;
  M++;
  if ((unsigned)N < usedN && (unsigned)M < usedM) {
    *newN = usedN;
    *newM = usedM;
    return;
  }
  n = (unsigned)N;
  if (n < usedN) n = usedN;
  if (n > maxN) n = maxN;
  if (n > usedN) {
    /* at least 10 % and at least 50 more rows; the 1.1 product truncates like the reference's
     * int = double assignment */
    if ((double)n < usedN * 1.1) n = (unsigned)(usedN * 1.1);
    if (n < usedN + 50) n = usedN + 50;
    if (n > maxN) n = maxN;
  }
  m = (unsigned)M;
  if (m < usedM) m = usedM;
  if (n < m) m = n;
  if (m > maxM) m = maxM;
  if (m > usedM) {
    if ((double)m < usedM * 1.1) m = (unsigned)(usedM * 1.1);
    if (m < usedM + 50) m = usedM + 50;
    if (m > maxM) m = maxM;
    if (m > usedN) m = usedN;
  }
  *newN = n;
  *newM = m;
}

/* grow to cover (N,M) as requested by an accessor; non-zero on allocation/device failure */
static int extend(stable_t *sp, int N, int M) {
  unsigned newN, newM;
  int rc = 0;
  lock(sp);
  stb_extend_policy(sp->usedN, sp->usedM, sp->maxN, sp->maxM, N, M, &newN, &newM);
  if (newN != sp->usedN || newM != sp->usedM) {
    stb_impl *im = sp->impl;
    if (newN > sp->usedN1) {
      /* allocate-copy-swap so a concurrent S_S1 reader never sees freed memory */
      double *s1 = malloc(sizeof(double) * newN);
      if (!s1) {
        unlock(sp);
        return 1;
      }
      memcpy(s1, sp->S1, sizeof(double) * sp->usedN1);
      memset(s1 + sp->usedN1, 0, sizeof(double) * (newN - sp->usedN1));
      retire(sp, NULL, 0, sp->S1);
      sp->S1 = s1;
      im->bytes_host += sizeof(double) * (newN - sp->usedN1);
      sp->usedN1 = newN;
    }
    {
      pending p;
      const int prev_dev = stb_device_enter(im->dev);
      rc = provision(sp, newN, newM, 1, &p) || build(sp, sp->a, newN, newM, p.hS, p.hV);
      if (!rc)
        publish(sp, &p, newN, newM);
      else
        pending_drop(&p);
      stb_device_leave(prev_dev);
    }
  }
  unlock(sp);
  return rc;
}

double S_S1(stable_t *sp, unsigned n) {
  double v;
  if (n == 0) return -HUGE_VAL;
  if (!sp->S1) return -HUGE_VAL;
  if (n <= sp->usedN) return sp->S1[n - 1];
  if (n > sp->maxN) {
    /* lib/stable.c:842-844: log(0) unless the asymptote flag is set; with it, Gamma(n-a)/Gamma(1-a)
     * is available in closed form, so answer exactly */
    if (!(sp->flags & S_ASYMPT)) return -HUGE_VAL;
    return lgamma(n - sp->a) - sp->lga;
  }
  lock(sp);
  if (n > sp->usedN1) {
    /* lib/stable.c:845-857: grow the cache by at least 10 % / 50 entries, capped at maxN */
    unsigned g = n;
    double *s1;
    if ((double)g < sp->usedN1 * 1.1) g = (unsigned)(sp->usedN1 * 1.1);
    if (g < sp->usedN1 + 50) g = sp->usedN1 + 50;
    if (g > sp->maxN) g = sp->maxN;
    s1 = malloc(sizeof(double) * g);
    if (!s1) {
      unlock(sp);
      return -HUGE_VAL;
    }
    memcpy(s1, sp->S1, sizeof(double) * sp->usedN1);
    memset(s1 + sp->usedN1, 0, sizeof(double) * (g - sp->usedN1));
    retire(sp, NULL, 0, sp->S1);
    sp->S1 = s1;
    ((stb_impl *)sp->impl)->bytes_host += sizeof(double) * (g - sp->usedN1);
    sp->usedN1 = g;
    account(sp);
  }
  if (sp->S1[n - 1] == 0) {
    /* lib/stable.c:859-864 */
    if (sp->S1[n - 2] == 0)
      sp->S1[n - 1] = lgamma(n - sp->a) - sp->lga;
    else
      sp->S1[n - 1] = sp->S1[n - 2] + log(n - 1 - sp->a);
  }
  v = sp->S1[n - 1];
  unlock(sp);
  return v;
}

double S_U(stable_t *sp, unsigned n, unsigned m) {
  /* lib/stable.c:875-883 */
  if (m == 1) return n - sp->a;
  if (m <= 1) yaps_quit("Bad constraints in S_U(%s,%u,%u)\n", sp->tag, n, m);
  return n - m * sp->a + 1 / S_V(sp, n, m);
}

double S_UV(stable_t *sp, unsigned n, unsigned m) {
  /* lib/stable.c:885-897 */
  double SV;
  if (m == 1) return -HUGE_VAL;
  if (m == n + 1) return 1; /* S^n_n == 1 */
  if (m == n) return (n + 1.0) / (n - 1.0);
  SV = S_V(sp, n, m);
  return (n - m * sp->a) * SV + 1.0;
}

double S_V(stable_t *sp, unsigned n, unsigned m) {
  if ((sp->flags & S_UVTABLE) == 0) return 0;
  if (m >= sp->usedM - 1 || n >= sp->usedN - 1) {
    /* lib/stable.c:903-925 */
    if (n > sp->maxN || m > sp->maxM) {
      if (n > sp->maxN && (sp->flags & S_ASYMPT)) {
        if (sp->a > 0) return (1.0 - pow(n, -sp->a)) / sp->a / (m - 1);
        {
          double ln = log(n);
          return ln / (m - 1) * exp(lgamma(1 + (m - 2) / ln) - lgamma(1 + (m - 1) / ln));
        }
      }
      if (sp->flags & S_QUITONBOUND) {
        if (sp->tag)
          yaps_quit("S_V(%u,%u,%lf) tagged '%s' hit bounds (%u,%u)\n", n, m, sp->a, sp->tag,
                    sp->maxN, sp->maxM);
        else
          yaps_quit("S_V(%u,%u,%lf) hit bounds\n", n, m, sp->a);
      }
      return 0;
    }
    {
      /* one step as the reference does (lib/stable.c:924); more only while the request is still
       * outside the table (see S_S) */
      int tries;
      if (extend(sp, n + 1, m + 1)) yaps_quit("S_extend() out of memory\n");
      for (tries = 0; tries < 3 && (m > sp->usedM || n > sp->usedN); tries++)
        if (extend(sp, n + 1, m + 1)) yaps_quit("S_extend() out of memory\n");
    }
  }
  if (m < 2) return 0;
  if (n < m) return 0;
  if (n > sp->usedN || m > sp->usedM) return 0; /* growth was capped by the max bounds */
  if (sp->flags & S_FLOAT) return sp->Vf[n - 2][m - 2];
  return sp->V[n - 2][m - 2];
}

double S_S(stable_t *sp, unsigned N, unsigned T) {
  /* test order of lib/stable.c:941-974 */
  if ((sp->flags & S_STABLE) == 0) return -HUGE_VAL;
  if (N == T) return 0;
  if (T == 1) return S_S1(sp, N);
  if (N < T || T == 0) return -HUGE_VAL;
  if (T > sp->usedM || N > sp->usedN) {
    if (N > sp->maxN || T > sp->maxM) {
      if (N > sp->maxN && (sp->flags & S_ASYMPT)) return S_asympt(sp, N, T);
      if (sp->flags & S_QUITONBOUND) {
        if (sp->tag)
          yaps_quit("S_S(%u,%u,%lf) tagged '%s' hit bounds\n", N, T, sp->a, sp->tag);
        else
          yaps_quit("S_S(%u,%u,%lf) hit bounds\n", N, T, sp->a);
      }
      return -HUGE_VAL;
    }
    {
      /* one growth step can stop short of T: the policy caps the new usedM at the OLD usedN
       * (lib/stable.c:626-628), after which the reference indexes past the end of the row.  Here
       * growth is repeated until the request is covered (the second step always suffices). */
      int tries;
      for (tries = 0; tries < 4 && (T > sp->usedM || N > sp->usedN); tries++)
        if (extend(sp, N + 1, T + 1)) yaps_quit("S_extend() out of memory\n");
      if (T > sp->usedM || N > sp->usedN) return -HUGE_VAL;
    }
  }
  if (sp->flags & S_FLOAT) return sp->Sf[N - 3][T - 2];
  return sp->S[N - 3][T - 2];
}

void S_free(stable_t *sp) {
  stb_impl *im;
  if (!sp) return;
  im = sp->impl;
  free(sp->tag);
  free(sp->S1);
  free(sp->S);
  free(sp->V);
  free(sp->Sf);
  free(sp->Vf);
  if (im) {
    const int prev_dev = stb_device_enter(im->dev);
    mirror_gen *g = im->retired;
    while (g) {
      mirror_gen *nx = g->next;
      host_slab_free(g->slab, g->pinned);
      free(g->rows);
      free(g);
      g = nx;
    }
    host_slab_free(im->h_S, im->h_S_pinned);
    host_slab_free(im->h_V, im->h_V_pinned);
    stb_device_free(im->d_S);
    stb_device_free(im->d_V);
    stb_device_free(im->d_Sf);
    stb_device_free(im->d_Vf);
    stb_device_free(im->d_S1);
    stb_device_free(im->d_ws);
    stb_device_leave(prev_dev);
    free(im);
  }
  if (sp->flags & S_THREADS) pthread_mutex_destroy(&sp->mutex);
  free(sp);
}

void S_report(stable_t *sp, FILE *fp) {
  /* text format of lib/stable.c:1025-1055, byte for byte (including the doubled newline of the
   * FILE variant) */
  const char *s = (sp->flags & S_STABLE) ? "+S" : "";
  const char *uv = (sp->flags & S_UVTABLE) ? "+U/V" : "";
  const char *ty = (sp->flags & S_FLOAT) ? "float" : "double";
  if (fp) {
    if (sp->tag)
      fprintf(fp, "S-table '%s': ", sp->tag);
    else
      fprintf(fp, "S-table: ");
    fprintf(fp, "a=%lf, N=%u/%u, M=%u/%u, %s%s %s", sp->a, sp->usedN, sp->maxN, sp->usedM,
            sp->maxM, s, uv, ty);
    fprintf(fp, " mem=%uk\n", sp->memalloced / 1024);
    fprintf(fp, "\n");
  } else {
    if (sp->tag)
      yaps_message("S-table '%s': ", sp->tag);
    else
      yaps_message("S-table: ");
    yaps_message("a=%lf, N=%u/%u, M=%u/%u, %s%s %s", sp->a, sp->usedN, sp->maxN, sp->usedM,
                 sp->maxM, s, uv, ty);
    yaps_message(" mem=%uk", sp->memalloced / 1024);
    yaps_message("\n");
  }
}

double S_asympt(stable_t *sp, unsigned n, unsigned m) {
  if (sp->a == 0) {
    /* lib/stable.c:1058-1065: Hwang's expansion for Stirling numbers of the first kind */
    double ln = log(n);
    return lgamma(n) + (m - 1) * log(ln) - lgamma(m) - lgamma(1 + (m - 1) / ln);
  } else {
    /* lib/stable.c:1066-1082: Gamma(n) / (Gamma(1-a) Gamma(m) a^{m-1} n^a) (1-n^{-a})^{m-1} */
    double acc = 0;
    double la1 = lgamma(1.0 - sp->a);
    double aln = sp->a * log((double)n);
    double np = pow(n, -sp->a);
    acc += lgamma((double)n) - la1 - lgamma((double)m) - (m - 1.0) * log(sp->a) - aln;
    if (np < 1e-5)
      acc -= (m - 1) * np * (1 + np * (0.5 + np / 3.0));
    else
      acc += (m - 1) * log(1.0 - np);
    return acc;
  }
}

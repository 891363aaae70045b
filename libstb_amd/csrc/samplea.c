/*
 * samplea.c -- one MCMC step for the Pitman-Yor discount a (include/psample.h).
 *
 * Host control flow as the reference's lib/samplea.c:155-225: bracket the move inside
 * [A_MIN, A_MAX] and +-SQUEEZEA of the current value, scan the counts for the table bounds, then
 * draw with ARMS (or the slice sampler) from the log-posterior `aterms` (lib/samplea.c:46-83).
 *
 * What differs: every evaluation of aterms -- rebuilding the N x M table of log S^n_{m,x} for the
 * trial discount x, gathering S_S(n,t) over all pairs and adding the restaurant terms -- runs on
 * the GPU.  The ragged n[i][k], t[i][k] arrays (or the getval callback) are flattened once per
 * call and kept resident in HBM (stb_groups_*), so an evaluation costs kernel launches and one
 * 8-byte read-back, not a host pass over the data.  There is no host evaluation path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/psample.h"
#include "../../include/stb_hip.h"
#include "sampler_trace.h"

#define NPRE 3 /* abscissae ARMS is known to ask for first (lib/arms.c:117-119) */

typedef struct {
  stb_groups_t *dev;
  int maxn, maxt;
  int verbose;
  int keep; /* the device set outlives this call (see `kept`) */
  int reused; /* ... and came from the previous call: its evaluations sum inside the table walk */
  /* values evaluated ahead of time, served when ARMS asks for exactly these abscissae */
  int npre;
  double xpre[NPRE], ypre[NPRE];
} a_posterior;

static double aterms(double x, void *vp) {
  a_posterior *ap = vp;
  double val;
  int i;
  if (x <= 0) {
    fprintf(stderr, "Illegal discount value in aterms()\n"); /* lib/samplea.c:50-53 */
    exit(1);
  }
  if (ap->verbose > 1) fprintf(stderr, "Extending S for M=%d a=%lf\n", ap->maxt, x); /* :54-56 */
  for (i = 0; i < ap->npre; i++)
    if (x == ap->xpre[i]) {
      stb_trace_add(x, ap->ypre[i]);
      return ap->ypre[i];
    }
  if (stb_groups_aterms(ap->dev, &x, 1, &val)) {
    fprintf(stderr, "aterms(): device evaluation failed: %s\n", stb_last_error());
    exit(1);
  }
  stb_trace_add(x, val);
  return val;
}

/* The device set of the last call, per calling thread.  A Gibbs sampler changes its counts between two calls
 * (test/demo.c:405-445), so what is kept is the set as a CONTAINER -- device buffers, pinned staging, stream, count
 * slab: a call with the same number of restaurants and pairs hands its pairs over with stb_groups_pairs_begin / _put /
 * _commit and allocates nothing.  Only with STB_SAMPLEA_CACHE=1 in the environment are the CONTENTS reused too, when a
 * 128-bit fingerprint of K, n, t and the bounds says the pairs are those of the last call (a sampler that resamples a
 * several times over unchanged counts); T and bpar are refreshed on every call. */
typedef struct {
  uint64_t a, b;
} fp128;
static _Thread_local struct { /* (one per calling thread: samplers of different threads do not share a set) */
  stb_groups_t *dev;
  fp128 fp;
  int have_fp; /* the set's pairs are those `fp` was taken from */
  int I;
  size_t G;
  unsigned N, M;
} kept;

void stb_sampleb_cache_clear(void); /* sampleb.c */

void stb_sampler_cache_clear(void) {
  if (kept.dev) stb_groups_free(kept.dev);
  memset(&kept, 0, sizeof(kept));
  stb_sampleb_cache_clear();
}

static uint64_t mix64(uint64_t h, uint64_t v) {
  h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  return h ^ (h >> 29);
}
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
/* the same job with the AES round as the mixing step (four 128-bit lanes, a round per 16 bytes of input, three more
 * rounds to fold them): 0.17 ms for the 6 MB of 10^6 pairs where the multiplicative hash below takes 0.48 -- a third
 * of a whole samplea call.  Only a fingerprint that says "the same pairs as last time" (STB_SAMPLEA_CACHE=1): 128 bits
 * that depend on every input bit; they need not be the same on every machine.  Not a cryptographic hash (the rounds are
 * un-keyed): an adversary could construct a collision, chance will not -- see INTEGRATION.md for what a collision does. */
__attribute__((target("aes,sse4.1"))) static fp128 hash_bytes_aes(uint64_t h, const void *p, size_t bytes) {
  const unsigned char *b = p;
  __m128i s0 = _mm_set_epi64x((long long)h, (long long)0x6a09e667f3bcc908ull),
          s1 = _mm_set_epi64x((long long)~h, (long long)0xbb67ae8584caa73bull),
          s2 = _mm_set_epi64x((long long)(h * 3u), (long long)0x3c6ef372fe94f82bull),
          s3 = _mm_set_epi64x((long long)(h ^ 0x5555u), (long long)0xa54ff53a5f1d36f1ull);
  unsigned char tail[64];
  size_t i = 0;
  for (; i + 64 <= bytes; i += 64) {
    s0 = _mm_aesenc_si128(s0, _mm_loadu_si128((const __m128i *)(b + i)));
    s1 = _mm_aesenc_si128(s1, _mm_loadu_si128((const __m128i *)(b + i + 16)));
    s2 = _mm_aesenc_si128(s2, _mm_loadu_si128((const __m128i *)(b + i + 32)));
    s3 = _mm_aesenc_si128(s3, _mm_loadu_si128((const __m128i *)(b + i + 48)));
  }
  memset(tail, 0, sizeof(tail));
  if (bytes > i) memcpy(tail, b + i, bytes - i);
  s0 = _mm_aesenc_si128(s0, _mm_loadu_si128((const __m128i *)(tail)));
  s1 = _mm_aesenc_si128(s1, _mm_loadu_si128((const __m128i *)(tail + 16)));
  s2 = _mm_aesenc_si128(s2, _mm_loadu_si128((const __m128i *)(tail + 32)));
  s3 = _mm_aesenc_si128(s3, _mm_loadu_si128((const __m128i *)(tail + 48)));
  s0 = _mm_aesenc_si128(s0, s1);
  s2 = _mm_aesenc_si128(s2, s3);
  s0 = _mm_aesenc_si128(s0, s2);
  s0 = _mm_aesenc_si128(s0, _mm_set_epi64x((long long)bytes, 0));
  s0 = _mm_aesenc_si128(s0, s0);
  {
    fp128 r;
    r.a = (uint64_t)_mm_extract_epi64(s0, 0);
    r.b = (uint64_t)_mm_extract_epi64(s0, 1);
    return r;
  }
}
#define STB_HAVE_AES_HASH 1
#endif

static uint64_t hash_bytes_mul(uint64_t h, const void *p, size_t bytes) {
  const unsigned char *b = p;
  uint64_t lanes[4] = {h, h ^ 0x6a09e667f3bcc908ull, h ^ 0xbb67ae8584caa73bull, h ^ 0x3c6ef372fe94f82bull};
  size_t i = 0;
  for (; i + 32 <= bytes; i += 32) {
    uint64_t w[4];
    memcpy(w, b + i, 32);
    lanes[0] = (lanes[0] ^ w[0]) * 0x9fb21c651e98df25ull;
    lanes[1] = (lanes[1] ^ w[1]) * 0xc2b2ae3d27d4eb4full;
    lanes[2] = (lanes[2] ^ w[2]) * 0x165667b19e3779f9ull;
    lanes[3] = (lanes[3] ^ w[3]) * 0x27d4eb2f165667c5ull;
    lanes[0] ^= lanes[0] >> 31;
    lanes[1] ^= lanes[1] >> 29;
    lanes[2] ^= lanes[2] >> 33;
    lanes[3] ^= lanes[3] >> 30;
  }
  for (; i < bytes; i++) lanes[i & 3] = (lanes[i & 3] ^ b[i]) * 0x100000001b3ull;
  return mix64(mix64(mix64(mix64(h, lanes[0]), lanes[1]), lanes[2]), lanes[3] ^ bytes);
}
static fp128 hash_bytes(fp128 h, const void *p, size_t bytes) {
  fp128 r;
#ifdef STB_HAVE_AES_HASH
  static int have_aes = -1; /* (a benign race: every thread computes the same value) */
  if (have_aes < 0) have_aes = __builtin_cpu_supports("aes") && __builtin_cpu_supports("sse4.1") ? 1 : 0;
  if (have_aes) {
    r = hash_bytes_aes(h.a ^ (h.b << 1 | h.b >> 63), p, bytes);
    r.a ^= h.b; /* (chained: every piece's fingerprint depends on all of the earlier ones') */
    return r;
  }
#endif
  r.a = hash_bytes_mul(h.a, p, bytes); /* (two passes with unrelated seeds: 2 x 64 bits) */
  r.b = hash_bytes_mul(h.b ^ 0xa54ff53a5f1d36f1ull, p, bytes);
  return r;
}

/* largest entry of an array (0 for an empty one); loops the compiler turns into vector code */
__attribute__((optimize("O3"))) static unsigned max_u32(const scnt_int *p, size_t n) {
  unsigned m = 0;
  size_t i;
  for (i = 0; i < n; i++) m = p[i] > m ? p[i] : m;
  return m;
}
__attribute__((optimize("O3"))) static unsigned max_u16(const stcnt_int *p, size_t n) {
  unsigned m = 0;
  size_t i;
  for (i = 0; i < n; i++) m = p[i] > m ? p[i] : m;
  return m;
}

/* environment switch for the sampler family, so both can be exercised from one build:
 * STB_SAMPLER=slice selects SliceSimple, anything else the compile-time default */
static int use_slice(void) {
  const char *s = getenv("STB_SAMPLER");
#ifdef PSAMPLE_ARS
  return s && strcmp(s, "slice") == 0;
#else
  return !(s && strcmp(s, "ars") == 0);
#endif
}

double samplea(double mya, int I, int *K, scnt_int *T, scnt_int **n, stcnt_int **t,
               void (*getval)(scnt_int *n, stcnt_int *t, unsigned i, unsigned k), double *bpar,
               rngp_t rng, int loops, int verbose) {
  double inita[3] = {A_MIN, 1, A_MAX};
  a_posterior ap;
  scnt_int *nflat = NULL;
  stcnt_int *tflat = NULL;
  size_t G = 0, g = 0;
  unsigned N, M, mn = 0, mt = 0;
  fp128 fp = {0x5eedull, 0x9e3779b97f4a7c15ull};
  double xspec[NPRE], yspec[NPRE];
  int i, k, cache, spec = 0, hit = 0;
  const int slice = use_slice();

  /* lib/samplea.c:161-177: start point nudged off the ends, move limited to +-SQUEEZEA */
  inita[1] = mya;
  if (fabs(inita[1] - A_MAX) / A_MAX < 0.00001) inita[1] = A_MAX * 0.999 + A_MIN * 0.001;
  if (fabs(inita[1] - A_MIN) / A_MIN < 0.00001) inita[1] = A_MIN * 0.999 + A_MAX * 0.001;
#ifdef SQUEEZEA
  if (inita[1] - SQUEEZEA > A_MIN) inita[0] = inita[1] - SQUEEZEA;
  if (inita[1] + SQUEEZEA < A_MAX) inita[2] = inita[1] + SQUEEZEA;
#endif
  /* ARMS starts from three abscissae it fixes before any evaluation (lib/arms.c:117-119, the same expression here, so
   * the same bits): they are evaluated in ONE batched device call */
  for (i = 0; i < NPRE; i++) xspec[i] = inita[0] + (i + 1.0) * (inita[2] - inita[0]) / (NPRE + 1.0);

  for (i = 0; i < I; i++) G += (size_t)(K[i] > 0 ? K[i] : 0);
  ap.verbose = verbose;
  if (getval) { /* (a callback has to be asked pair by pair: flattened first, then handed over as arrays) */
    nflat = malloc(sizeof(*nflat) * (G ? G : 1));
    tflat = malloc(sizeof(*tflat) * (G ? G : 1));
    if (!nflat || !tflat) {
      fprintf(stderr, "Out of memory for S table\n");
      exit(1);
    }
    for (i = 0; i < I; i++)
      for (k = 0; k < K[i]; k++, g++) getval(&nflat[g], &tflat[g], i, k);
  }
  {
    const char *ce = getenv("STB_SAMPLEA_CACHE");
    cache = ce && strcmp(ce, "1") == 0;
  }
  /* the thread's device set, as a container: one of this shape is used again, another shape makes a new one (empty:
   * pairs and bounds follow) */
  if (kept.dev && (kept.I != I || kept.G != G)) stb_sampler_cache_clear();
  if (!kept.dev) {
    kept.dev = stb_groups_create(I, K, NULL, NULL, NULL, NULL, 0, 0, NPRE);
    if (!kept.dev) {
      fprintf(stderr, "Out of memory for S table (%s)\n", stb_last_error()); /* lib/samplea.c:61-64 */
      exit(1);
    }
    kept.I = I;
    kept.G = G;
    kept.have_fp = 0;
  }
  ap.dev = kept.dev;
  if (cache) {
    /* The contents may still be right (a sampler that resamples a over unchanged counts): the three starting
     * evaluations are queued on that guess BEFORE the host reads the caller's pairs, so the device walks the tables
     * while the host takes the fingerprint of 6 MB; should it differ, the values are waited for and thrown away. */
    size_t off = 0;
    if (kept.have_fp && !slice)
      spec = !stb_groups_update_restaurants(kept.dev, T, bpar) && !stb_groups_aterms_async(kept.dev, xspec, NPRE, yspec, NULL);
    fp = hash_bytes(fp, K, sizeof(int) * (size_t)(I > 0 ? I : 0));
    for (i = 0; i < I; i++) {
      const size_t Ki = (size_t)(K[i] > 0 ? K[i] : 0);
      const scnt_int *ni = getval ? nflat + off : n[i];
      const stcnt_int *ti = getval ? tflat + off : t[i];
      const unsigned a = max_u32(ni, Ki), b = max_u16(ti, Ki);
      mn = a > mn ? a : mn;
      mt = b > mt ? b : mt;
      fp = hash_bytes(fp, ni, sizeof(*ni) * Ki);
      fp = hash_bytes(fp, ti, sizeof(*ti) * Ki);
      off += Ki;
    }
    hit = kept.have_fp && kept.fp.a == fp.a && kept.fp.b == fp.b;
    if (!hit && spec) { /* the guess was wrong: let the device finish before the pairs go */
      (void)stb_groups_wait(kept.dev);
      spec = 0;
    }
  }
  if (!hit) {
    /* the pairs go to the device restaurant after restaurant, copied once (into pinned memory, each piece on its way
     * while the next is copied); the largest n and t fall out of the same pass */
    int bad = stb_groups_pairs_begin(kept.dev);
    if (!bad && G) {
      if (getval) bad = stb_groups_pairs_put(kept.dev, nflat, tflat, G, &mn, &mt);
      else bad = stb_groups_pairs_put_ragged(kept.dev, I, K, n, t, &mn, &mt);
    }
    if (bad) {
      fprintf(stderr, "Out of memory for S table (%s)\n", stb_last_error());
      exit(1);
    }
  }
  free(nflat);
  free(tflat);
  /* the bounds: maxn = max n + 1, maxt = max t + 1, both at least 1 (lib/samplea.c:184-208); the table aterms builds
   * is S_make(maxn,maxt,maxn,maxt) (lib/samplea.c:60) after S_make's clamps (lib/stable.c:118-129): M = max(maxt,10),
   * N = max(maxn,M) */
  ap.maxt = (int)mt + 1;
  ap.maxn = G ? (int)mn + 1 : 1;
  M = ap.maxt < 10 ? 10u : (unsigned)ap.maxt;
  N = (unsigned)ap.maxn < M ? M : (unsigned)ap.maxn;
  {
    /* The table is private to this call and its cells do not depend on its bounds: the device walks one whose bounds
     * are rounded up to multiples of 128 (at most 127 rows of 22 ns more), so that a largest count that moves a little
     * from call to call -- a Gibbs sampler's does -- finds the geometry, the workspace and the count slab of the last
     * call.  STB_SAMPLEA_QUANT=1 walks the reference's exact bounds. */
    const char *qe = getenv("STB_SAMPLEA_QUANT");
    unsigned q = qe && *qe ? (unsigned)atoi(qe) : 128u;
    if (q > 1 && q <= 4096) {
      N = (N + q - 1) / q * q;
      M = (M + q - 1) / q * q;
      if (M > N) M = N;
    }
  }
  if (hit && (kept.N != N || kept.M != M)) hit = 0; /* (cannot happen: the same pairs have the same maxima) */
  if (!hit) {
    if (stb_groups_pairs_commit(kept.dev, T, bpar, N, M)) {
      fprintf(stderr, "Out of memory for S table (%s)\n", stb_last_error());
      exit(1);
    }
    kept.N = N;
    kept.M = M;
    kept.fp = fp;
    kept.have_fp = cache;
  } else if (!spec && stb_groups_update_restaurants(kept.dev, T, bpar)) {
    fprintf(stderr, "aterms(): %s\n", stb_last_error());
    exit(1);
  }
  ap.keep = 1;
  ap.reused = hit;
  ap.npre = 0;
  if (!slice) {
    double y3[NPRE];
    int bad;
    if (spec) {
      bad = stb_groups_wait(ap.dev);
      for (i = 0; i < NPRE; i++) y3[i] = yspec[i];
    } else {
      bad = stb_groups_aterms(ap.dev, xspec, NPRE, y3);
    }
    if (bad) {
      fprintf(stderr, "aterms(): device evaluation failed: %s\n", stb_last_error());
      exit(1);
    }
    for (i = 0; i < NPRE; i++) {
      ap.xpre[i] = xspec[i];
      ap.ypre[i] = y3[i];
    }
    ap.npre = NPRE;
  }

  stb_trace_reset();
  if (!slice) {
    int code = arms_simple(3, inita, inita + 2, aterms, &ap, 0, inita + 1, &mya); /* :210 */
    stb_trace_code(code);
    if (mya < inita[0] || mya > inita[2]) {
      fprintf(stderr, "Arms_simple(apar) returned value out of bounds\n");
      exit(1);
    }
  } else {
    inita[1] = A_MAX; /* lib/samplea.c:217: the slice bracket is [lower, A_MAX] */
    if (SliceSimple(&mya, aterms, inita, rng, loops, &ap)) {
      fprintf(stderr, "SliceSimple error\n");
      exit(1);
    }
  }
  /* (the set stays with the thread, as a container; stb_sampler_cache_clear releases it) */
  return mya;
}

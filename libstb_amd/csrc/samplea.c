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

/* The device copy of the pairs of the last call: a Gibbs sampler resamples a over and over on counts
 * that change slowly or not at all, and uploading and sorting 10^6 pairs costs as much as four
 * posterior evaluations.  Kept only while the next call brings exactly the same pairs (64-bit hash
 * of K, n, t plus the shapes); T and bpar are refreshed on every call. */
static _Thread_local struct { /* (one per calling thread: samplers of different threads do not share a set) */
  stb_groups_t *dev;
  uint64_t hash;
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
 * of a whole samplea call.  Only a fingerprint that says "the same pairs as last time": any 64 bits that depend on every
 * input bit will do, they need not be the same on every machine. */
__attribute__((target("aes,sse4.1"))) static uint64_t hash_bytes_aes(uint64_t h, const void *p, size_t bytes) {
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
  return (uint64_t)_mm_extract_epi64(s0, 0) ^ (uint64_t)_mm_extract_epi64(s0, 1);
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
static uint64_t hash_bytes(uint64_t h, const void *p, size_t bytes) {
#ifdef STB_HAVE_AES_HASH
  static int have_aes = -1; /* (a benign race: every thread computes the same value) */
  if (have_aes < 0) have_aes = __builtin_cpu_supports("aes") && __builtin_cpu_supports("sse4.1") ? 1 : 0;
  if (have_aes) return hash_bytes_aes(h, p, bytes);
#endif
  return hash_bytes_mul(h, p, bytes);
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
  scnt_int *nflat;
  stcnt_int *tflat;
  size_t G = 0, g = 0;
  unsigned N, M;
  uint64_t pairs_hash = 0;
  double xspec[NPRE], yspec[NPRE];
  int i, k, keep, spec = 0;

  /* lib/samplea.c:161-177: start point nudged off the ends, move limited to +-SQUEEZEA */
  inita[1] = mya;
  if (fabs(inita[1] - A_MAX) / A_MAX < 0.00001) inita[1] = A_MAX * 0.999 + A_MIN * 0.001;
  if (fabs(inita[1] - A_MIN) / A_MIN < 0.00001) inita[1] = A_MIN * 0.999 + A_MAX * 0.001;
#ifdef SQUEEZEA
  if (inita[1] - SQUEEZEA > A_MIN) inita[0] = inita[1] - SQUEEZEA;
  if (inita[1] + SQUEEZEA < A_MAX) inita[2] = inita[1] + SQUEEZEA;
#endif

  /* the bounds: maxn = max n + 1, maxt = max t + 1, both at least 1 (lib/samplea.c:184-208), and a hash of the
   * pairs, in ONE pass over the caller's arrays; they are flattened (copied) only when the device does not hold
   * them already -- a Gibbs sampler resamples a over counts that change slowly or not at all, and the copy costs as
   * much as three posterior evaluations */
  for (i = 0; i < I; i++) G += (size_t)(K[i] > 0 ? K[i] : 0);
  nflat = NULL;
  tflat = NULL;
  ap.maxt = 1;
  ap.maxn = 1;
  ap.verbose = verbose;
  if (getval) { /* (a callback has to be asked pair by pair: flattened first, then as arrays) */
    nflat = malloc(sizeof(*nflat) * (G ? G : 1));
    tflat = malloc(sizeof(*tflat) * (G ? G : 1));
    if (!nflat || !tflat) {
      fprintf(stderr, "Out of memory for S table\n");
      exit(1);
    }
    for (i = 0; i < I; i++)
      for (k = 0; k < K[i]; k++, g++) getval(&nflat[g], &tflat[g], i, k);
  }
  /* A kept set is most likely still the right one (a Gibbs sampler's counts change slowly or not at all): its three
   * pre-evaluations (below) are queued on that guess BEFORE the host reads the caller's pairs, so the device walks the
   * tables while the host hashes 6 MB; should the hash then differ, the values are waited for and thrown away. */
  {
    const char *ce = getenv("STB_SAMPLEA_CACHE");
    keep = !(ce && strcmp(ce, "0") == 0);
  }
  for (i = 0; i < NPRE; i++) xspec[i] = inita[0] + (i + 1.0) * (inita[2] - inita[0]) / (NPRE + 1.0);
  if (keep && kept.dev && kept.I == I && kept.G == G && !use_slice())
    spec = !stb_groups_update_restaurants(kept.dev, T, bpar) && !stb_groups_aterms_async(kept.dev, xspec, NPRE, yspec, NULL);
  {
    uint64_t hh = hash_bytes(0x5eedull, K, sizeof(int) * (size_t)(I > 0 ? I : 0));
    size_t off = 0;
    for (i = 0; i < I; i++) {
      const size_t Ki = (size_t)(K[i] > 0 ? K[i] : 0);
      const scnt_int *ni = getval ? nflat + off : n[i];
      const stcnt_int *ti = getval ? tflat + off : t[i];
      const unsigned mn = max_u32(ni, Ki), mt = max_u16(ti, Ki);
      if ((int)mt >= ap.maxt) ap.maxt = (int)mt + 1;
      if (Ki && mn >= (unsigned)ap.maxn) ap.maxn = (int)mn + 1;
      hh = mix64(hash_bytes(hh, ni, sizeof(*ni) * Ki), hash_bytes(hh, ti, sizeof(*ti) * Ki));
      off += Ki;
    }
    pairs_hash = hh;
  }
  /* the table aterms builds is S_make(maxn,maxt,maxn,maxt) (lib/samplea.c:60) after S_make's
   * clamps (lib/stable.c:118-129): M = max(maxt,10), N = max(maxn,M) */
  M = ap.maxt < 10 ? 10u : (unsigned)ap.maxt;
  N = (unsigned)ap.maxn < M ? M : (unsigned)ap.maxn;
  {
    const uint64_t h = pairs_hash;
    ap.reused = 0;
    if (keep && kept.dev && kept.hash == h && kept.I == I && kept.G == G && kept.N == N && kept.M == M) {
      ap.dev = kept.dev;
      ap.reused = 1;
      if (!spec && stb_groups_update_restaurants(ap.dev, T, bpar)) {
        stb_sampler_cache_clear();
        ap.dev = NULL;
      }
    } else {
      if (spec) (void)stb_groups_wait(kept.dev); /* the guess was wrong: let the device finish before the set goes */
      spec = 0;
      stb_sampler_cache_clear();
      ap.dev = NULL;
    }
    if (!ap.dev) {
      if (!getval) { /* not on the device yet: now the pairs are copied, restaurant after restaurant */
        size_t off = 0;
        nflat = malloc(sizeof(*nflat) * (G ? G : 1));
        tflat = malloc(sizeof(*tflat) * (G ? G : 1));
        if (!nflat || !tflat) {
          fprintf(stderr, "Out of memory for S table\n");
          exit(1);
        }
        for (i = 0; i < I; i++) {
          const size_t Ki = (size_t)(K[i] > 0 ? K[i] : 0);
          memcpy(nflat + off, n[i], sizeof(*nflat) * Ki);
          memcpy(tflat + off, t[i], sizeof(*tflat) * Ki);
          off += Ki;
        }
      }
      ap.dev = stb_groups_create(I, K, T, nflat, tflat, bpar, N, M, NPRE);
    }
    if (ap.dev && keep) {
      kept.dev = ap.dev;
      kept.hash = h;
      kept.I = I;
      kept.G = G;
      kept.N = N;
      kept.M = M;
    }
    ap.keep = keep;
  }
  free(nflat);
  free(tflat);
  if (!ap.dev) {
    fprintf(stderr, "Out of memory for S table (%s)\n", stb_last_error()); /* lib/samplea.c:61-64 */
    exit(1);
  }
  ap.npre = 0;
  {
    if (!use_slice()) {
      /* ARMS starts from three abscissae it fixes before any evaluation (lib/arms.c:117-119, the
       * same expression here, so the same bits): evaluate them in ONE batched device call */
      double x3[NPRE], y3[NPRE];
      int bad;
      for (i = 0; i < NPRE; i++) x3[i] = xspec[i];
      /* (a fresh set: through stored tables, no set-up; a kept one: the fused evaluation, whose cell lists it has --
       * queued above when the set was there before the pairs were read) */
      if (spec && ap.reused && ap.dev) {
        bad = stb_groups_wait(ap.dev);
        spec = 0;
        for (i = 0; i < NPRE; i++) y3[i] = yspec[i];
      } else {
        bad = ap.reused ? stb_groups_aterms(ap.dev, x3, NPRE, y3) : stb_groups_aterms_tables(ap.dev, x3, NPRE, y3);
      }
      if (bad) {
        fprintf(stderr, "aterms(): device evaluation failed: %s\n", stb_last_error());
        exit(1);
      }
      for (i = 0; i < NPRE; i++) {
        ap.xpre[i] = x3[i];
        ap.ypre[i] = y3[i];
      }
      ap.npre = NPRE;
    }
  }

  stb_trace_reset();
  if (!use_slice()) {
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
  if (!ap.keep) stb_groups_free(ap.dev); /* else: kept for the next call */
  return mya;
}

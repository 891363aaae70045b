// groups.hip -- stb_groups_*: the (n,t) pairs and per-restaurant totals of one samplea call resident
// in HBM, and aterms (reference lib/samplea.c:46-83) evaluated on them for one discount or a grid.

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <atomic>
#include <vector>

#include "stb_common.h"

void stb_grid_tile_offsets(const grid_geom &g, std::vector<unsigned> &off);  // grid_hb.hip

#include "groups.h"

// The sweep gathers table[row(n) + t]; pairs arrive in restaurant order, i.e. random in (n,t), and a
// random 8-byte gather moves a whole 64-byte sector.  Sorting the pairs once by (n,t) (they are reused
// for every evaluation of a samplea call and for all D tables of a grid) makes neighbouring threads
// read neighbouring addresses.  The sum is order-independent up to rounding and stays deterministic.
__global__ void k_pack_pairs(const uint32_t *n, const uint16_t *t, uint64_t G, uint64_t *key) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) key[g] = ((uint64_t)n[g] << 16) | t[g];
}
__global__ void k_unpack_pairs(const uint64_t *key, uint64_t G, uint32_t *n, uint16_t *t) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) {
    n[g] = (uint32_t)(key[g] >> 16);
    t[g] = (uint16_t)(key[g] & 0xffff);
  }
}

static int sort_pairs(uint32_t *d_n, uint16_t *d_t, uint64_t G, hipStream_t st) {
  if (G < 2) return 0;
  uint64_t *k0 = nullptr, *k1 = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  int rc = 1;
  do {
    if (stb_pool_malloc((void **)&k0, sizeof(uint64_t) * G) != hipSuccess || stb_pool_malloc((void **)&k1, sizeof(uint64_t) * G) != hipSuccess) {
      stb_fail("sort_pairs: out of device memory");
      break;
    }
    const unsigned blocks = (unsigned)((G + 255) / 256);
    hipLaunchKernelGGL(k_pack_pairs, dim3(blocks), dim3(256), 0, st, d_n, d_t, G, k0);
    if (rocprim::radix_sort_keys(nullptr, tmp_bytes, k0, k1, (size_t)G, 0, 48, st) != hipSuccess) {
      stb_fail("sort_pairs: radix_sort_keys (size query) failed");
      break;
    }
    if (stb_pool_malloc(&tmp, tmp_bytes ? tmp_bytes : 1) != hipSuccess) {
      stb_fail("sort_pairs: out of device memory");
      break;
    }
    if (rocprim::radix_sort_keys(tmp, tmp_bytes, k0, k1, (size_t)G, 0, 48, st) != hipSuccess) {
      stb_fail("sort_pairs: radix_sort_keys failed");
      break;
    }
    hipLaunchKernelGGL(k_unpack_pairs, dim3(blocks), dim3(256), 0, st, k1, G, d_n, d_t);
    if (hipStreamSynchronize(st) != hipSuccess) {
      stb_fail("sort_pairs: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    rc = 0;
  } while (0);
  // (the stream was synchronised above, or nothing was launched on these buffers)
  if (rc != 0) (void)hipStreamSynchronize(st);
  stb_pool_free(k0);
  stb_pool_free(k1);
  stb_pool_free(tmp);
  return rc;
}

extern "C" void stb_groups_free(stb_groups_t *g) {
  STB_ENTRY;
  if (!g) return;
  const int prev_dev = stb_device_enter(g->dev);
  std::vector<void *> ptrs = {g->d_n, g->d_T, g->d_t, g->d_bpar, g->d_tables, g->d_S1, g->d_out, g->d_ws_fill, g->d_ws_sweep, g->d_ws_terms,
                              g->d_cnt, g->d_n2, g->d_t2, g->d_dotp, g->d_slab, g->d_icnt, g->d_ninf, g->d_scan_tmp};
  for (int w = 0; w < STB_NLISTS; w++)
    for (void *q : {(void *)g->d_item_ptr[w], (void *)g->d_ent_pos[w], (void *)g->d_ent_cnt[w], (void *)g->d_tile_off[w], (void *)g->d_dense[w],
                    (void *)g->d_tinfo[w], (void *)g->d_jobs[w], (void *)g->d_tjob[w], (void *)g->d_tnw[w], (void *)g->d_toff[w]})
      ptrs.push_back(q);
  if (g->st) (void)hipStreamSynchronize(g->st);  // nothing may still be using the buffers
  for (void *p : ptrs) stb_pool_free(p);
  stb_pool_free(g->h_out);
  stb_pool_free(g->h_pn);
  stb_pool_free(g->h_pt);
  stb_pool_free(g->h_T);
  stb_pool_free(g->h_bpar);
  if (g->ev_dep) (void)hipEventDestroy(g->ev_dep);
  if (g->ev_done) (void)hipEventDestroy(g->ev_done);
  for (auto &e : g->ev)
    if (e) (void)hipEventDestroy(e);
  if (g->st) (void)hipStreamDestroy(g->st);
  free(g);
  stb_device_leave(prev_dev);
}

#define GCHK(expr)                                                                            \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      stb_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);        \
      stb_groups_free(g);                                                                     \
      return nullptr;                                                                         \
    }                                                                                         \
  } while (0)

// occurrence count of every table cell among the pairs (same classification as k_sweep_partial)
__global__ void k_count_pairs(const uint32_t *n, const uint16_t *t, uint64_t G, unsigned N, unsigned M,
                              unsigned *cnt) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const unsigned nn = n[g], tt = t[g];
  if (nn <= 1 || nn == tt || tt <= 1 || nn < tt || tt > M || nn > N) return;
  atomicAdd(&cnt[stb_row_offset(nn, M) + (tt - 2)], 1u);
}

// out[d] += sum of the DOT kernel's partial sums of table d, in a fixed order (a table of 10^4 columns has
// 41 000 tiles in the halo-block form: 256 threads, each its stride, then a fixed tree)
__global__ __launch_bounds__(256) void k_dot_reduce(const double *dotp, int parts, double *out) {
  __shared__ dd_t red[4];
  const int d = blockIdx.x;
  dd_t acc{0.0, 0.0};
  for (int i = threadIdx.x; i < parts; i += 256) dd_add(acc, dotp[(size_t)d * parts + i]);
  const dd_t tot = block_reduce_dd(acc, red);
  if (threadIdx.x == 0) out[d] += tot.hi + tot.lo;
}

// Last launch of a fused evaluation in the halo-block or the grid form: per discount the fill's partial sums
// (mode 1: one double per tile; mode 2: per strip an exact exponent sum and a mantissa-log sum) and the restaurant
// terms' partial sums, each in a fixed order, their total written to pinned host memory together with the fill's
// error words -- so that the host waits ONCE and copies nothing.
__global__ __launch_bounds__(256) void k_eval_tail(const double *dotp, int parts, const unsigned *parts_extra, int mode, const dd_t *tpart, int nbt, unsigned long long inf,
                                                   const unsigned long long *inf_dev, const unsigned *hdr, double *out_dev, double *out_host, int Dmax,
                                                   double *out_user, double seq) {
  __shared__ dd_t red[4];
  __shared__ double ks[4];
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const int d = blockIdx.x;
  dd_t acc{0.0, 0.0};
  double k = 0.0;  // (integers below 2^53: exact in any order)
  if (parts_extra) parts += (int)*parts_extra;  // (the grid form's helper jobs: their number lives with their list, on the device)
  if (mode == 2) {
    for (int i = threadIdx.x; i < parts; i += 256) {
      k += dotp[((size_t)d * parts + i) * 2];
      dd_add(acc, dotp[((size_t)d * parts + i) * 2 + 1]);
    }
  } else {
    for (int i = threadIdx.x; i < parts; i += 256) dd_add(acc, dotp[(size_t)d * parts + i]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) k += __shfl_down(k, off, 64);
  if ((threadIdx.x & 63) == 0) ks[threadIdx.x >> 6] = k;
  dd_t tot = block_reduce_dd(acc, red);
  dd_t tv{0.0, 0.0};
  for (int b = threadIdx.x; b < nbt; b += 256) dd_merge(tv, tpart[(size_t)d * nbt + b]);
  tv = block_reduce_dd(tv, red);
  if (threadIdx.x == 0) {
    if (mode == 2) {
      k = (ks[0] + ks[1]) + (ks[2] + ks[3]);
      // k ln2 in two pieces: k < 2^38 and LN2_HI has 32 trailing zero bits, so the first product rounds at 2^-53 relative
      dd_add(tot, k * LN2_LO);
      dd_add(tot, k * LN2_HI);
    }
    if (inf_dev) inf += *inf_dev;  // (lists built from the count slab: counted on the device)
    const double dots = inf ? -HUGE_VAL : tot.hi + tot.lo;
    const double terms = tv.hi + tv.lo;
    out_dev[d] = dots;
    out_dev[Dmax + d] = terms;
    out_host[d] = terms + dots;
    if (out_user) out_user[d] = terms + dots;  // (stb_groups_aterms_device: the caller's device buffer)
    if (d == 0) {
      out_host[2 * Dmax] = (double)hdr[1];
      out_host[2 * Dmax + 1] = (double)hdr[2];
      // (what the walk took, first workgroup's start to here, in 100 MHz ticks: the host compares it with what the
      // geometry should take -- a GPU shared with other processes shows there and nowhere else)
      const unsigned long long t0 = *reinterpret_cast<const unsigned long long *>(hdr + STB_HDR_T0);
      out_host[2 * Dmax + 3] = t0 ? (double)((unsigned long long)wall_clock64() - t0) : 0.0;
      // (one discount: everything the host waits for is written -- it spins on this word instead of waiting for the
      // launch's completion signal, which reaches it microseconds later; seq = 0: nobody spins)
      if (seq != 0.0 && gridDim.x == 1) {
        __threadfence_system();
        __hip_atomic_store(out_host + 2 * Dmax + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

int stb_restaurant_partials(const double *x_host, int D, const uint32_t *d_T, const double *d_bpar, uint64_t I, void *d_ws,
                            size_t ws_bytes, double *a_out, const dd_t **partial_out, int *nb_out, hipStream_t st);  // sweep_terms.hip

// What depends on the table bounds.  A Gibbs sampler's largest count moves from call to call, and with it the bounds
// of the table samplea builds (lib/samplea.c:186-208): everything below is given back to the buffer cache and asked
// for again in the new sizes (the cache hands a buffer of up to 5/4 the size out again).
int stb_groups_set_bounds(stb_groups_t *g, unsigned N, unsigned M) {
  if (g->have_bounds && g->N == N && g->M == M) return 0;
  if (N < 1 || M < 1) return stb_fail("stb_groups: table bounds N=%u M=%u", N, M);
  HIPCHK(hipStreamSynchronize(g->st));
  g->have_bounds = 0;  // (until everything below has its new size: a failed allocation must not leave a set that looks ready)
  stb_lists_drop(g, false);
  void **ptrs[] = {(void **)&g->d_tables, (void **)&g->d_S1, &g->d_ws_fill};
  for (void **p : ptrs) {
    stb_pool_free(*p);
    *p = nullptr;
  }
  g->N = N;
  g->M = M;
  g->hb_sum_C = stb_hb_sum_C(N, M, g->Dmax);
  g->tstride = (stb_table_elems(N, M) + 31) & ~31ull;
  g->ws_fill = stb_fill_workspace_bytes(N, M, g->Dmax);
  g->ws_zero = 0;
  HIPCHK(stb_pool_malloc((void **)&g->d_ws_fill, g->ws_fill));
  g->fused = stb_env_int("STB_ATERMS_FUSED", 1) && N >= 3 && N < (1u << 27);  // set up on first use
  g->have_bounds = 1;
  return 0;
}

// The tables of the evaluations that store them (one discount on a set without lists, the fallback of a fused
// evaluation that gave up, stb_groups_aterms_tables) are asked for when the first of those comes: a grid of 64
// discounts at N = M = 10^4 is 27 GB of tables that the fused evaluation never touches.
static int ensure_tables(stb_groups_t *g) {
  if (g->d_tables && g->d_S1 && g->d_ws_sweep) return 0;
  if (!g->d_tables && stb_pool_malloc((void **)&g->d_tables, sizeof(double) * g->tstride * g->Dmax) != hipSuccess)
    return stb_fail("stb_groups_aterms: out of device memory for %d tables of %u x %u", g->Dmax, g->N, g->M);
  if (!g->d_S1 && stb_pool_malloc((void **)&g->d_S1, sizeof(double) * (size_t)g->N * g->Dmax) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
  if (!g->d_ws_sweep) {
    g->ws_sweep = stb_sweep_workspace_bytes(g->G, g->Dmax);
    if (stb_pool_malloc((void **)&g->d_ws_sweep, g->ws_sweep) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
  }
  return 0;
}

static stb_groups_t *groups_create_here(int I, const int *K, const uint32_t *T, const uint32_t *nflat,
                                        const uint16_t *tflat, const double *bpar, unsigned N, unsigned M, int Dmax) {
  if (stb_device_count() < 1) {
    stb_fail("stb_groups_create: no HIP device (libstb_amd has no CPU path)");
    return nullptr;
  }
  if (Dmax < 1 || Dmax > STB_TERMS_DMAX) {
    stb_fail("stb_groups_create: Dmax=%d (1..%d)", Dmax, STB_TERMS_DMAX);
    return nullptr;
  }
  stb_groups_t *g = (stb_groups_t *)calloc(1, sizeof(*g));
  if (!g) {
    stb_fail("stb_groups_create: out of host memory");
    return nullptr;
  }
  uint64_t G = 0;
  for (int i = 0; i < I; i++) G += (uint64_t)(K[i] > 0 ? K[i] : 0);
  if (hipGetDevice(&g->dev) != hipSuccess) {
    stb_fail("stb_groups_create: %s", hipGetErrorString(hipGetLastError()));
    free(g);
    return nullptr;
  }
  g->I = I;
  g->G = G;
  g->Dmax = Dmax;
  GCHK(hipStreamCreate(&g->st));
  for (auto &e : g->ev) GCHK(hipEventCreate(&e));
  GCHK(stb_pool_malloc((void **)&g->d_n, sizeof(uint32_t) * (G ? G : 1)));
  GCHK(stb_pool_malloc((void **)&g->d_t, sizeof(uint16_t) * (G ? G : 1)));
  GCHK(stb_pool_malloc((void **)&g->d_T, sizeof(uint32_t) * (I > 0 ? I : 1)));
  GCHK(stb_pool_malloc((void **)&g->d_bpar, sizeof(double) * (I > 0 ? I : 1)));
  GCHK(stb_pool_malloc((void **)&g->h_T, sizeof(uint32_t) * (I > 0 ? I : 1), 1));
  GCHK(stb_pool_malloc((void **)&g->h_bpar, sizeof(double) * (I > 0 ? I : 1), 1));
  GCHK(stb_pool_malloc((void **)&g->d_out, sizeof(double) * 2 * Dmax));
  GCHK(stb_pool_malloc((void **)&g->h_out, sizeof(double) * (2 * Dmax + 4), 1));
  GCHK(hipHostGetDevicePointer((void **)&g->h_out_dev, g->h_out, 0));
  GCHK(hipEventCreateWithFlags(&g->ev_dep, hipEventDisableTiming));
  GCHK(hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming));
  g->ws_terms = stb_terms_workspace_bytes((uint64_t)I, Dmax);
  GCHK(stb_pool_malloc((void **)&g->d_ws_terms, g->ws_terms));
  if (N && M && stb_groups_set_bounds(g, N, M)) {
    stb_groups_free(g);
    return nullptr;
  }
  // (a set may be made empty -- no pairs, or no bounds either -- and filled with stb_groups_pairs_begin / _put / _commit)
  if (nflat && tflat) {
    if (!g->have_bounds) {
      stb_fail("stb_groups_create: pairs without table bounds");
      stb_groups_free(g);
      return nullptr;
    }
    int bad = stb_groups_pairs_begin(g);
    for (uint64_t o = 0; !bad && o < G; o += (1u << 18)) bad = stb_groups_pairs_put(g, nflat + o, tflat + o, G - o < (1u << 18) ? G - o : (1u << 18), nullptr, nullptr);
    if (!bad) bad = stb_groups_pairs_commit(g, T, bpar, N, M);
    if (bad) {
      stb_groups_free(g);
      return nullptr;
    }
  } else if (T && bpar && I > 0 && stb_groups_update_restaurants(g, T, bpar)) {
    stb_groups_free(g);
    return nullptr;
  }
  if (T && bpar) g->reused = 0;  // (what stb_groups_update_restaurants marks is a set that serves call after call)
  return g;
}

// the pairs in (n, t) order, for the gather over stored tables (on first need: see lists.hip)
int stb_groups_sort_pairs(stb_groups_t *g) {
  if (g->sorted || !stb_env_int("STB_SORT_PAIRS", 1)) return 0;
  if (sort_pairs(g->d_n, g->d_t, g->G, g->st)) return 1;
  g->sorted = 1;
  return 0;
}

// The set lives on the device stb_get_device() names (stb_set_device / STB_DEVICE / the runtime's
// current device); the caller's current device is put back before returning.
extern "C" stb_groups_t *stb_groups_create(int I, const int *K, const uint32_t *T, const uint32_t *nflat,
                                           const uint16_t *tflat, const double *bpar, unsigned N, unsigned M, int Dmax) {
  STB_ENTRY;
  const int prev_dev = stb_device_enter(stb_get_device());
  stb_groups_t *g = groups_create_here(I, K, T, nflat, tflat, bpar, N, M, Dmax);
  stb_device_leave(prev_dev);
  return g;
}

// ---- sparse set-up of the fused evaluation, all on the device -----------------------------------
#define STB_KEY_OTHER 0xfffffffffffffffeull  // a pair that addresses no table cell (t = 1, t = n, out of bounds)
#define STB_KEY_SKIP 0xffffffffffffffffull   // n <= 1: contributes nothing (lib/samplea.c:78)

// key of a pair: (item index << 9) | (row in trip << 6) | column in slice, item = trip * nsg + slice
__global__ void k_item_keys(const uint32_t *n, const uint16_t *t, uint64_t G, unsigned N, unsigned M, unsigned nsg,
                            unsigned col0, uint64_t *key, uint32_t *payload) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const unsigned nn = n[g], tt = t[g];
  uint64_t k;
  if (nn <= 1) k = STB_KEY_SKIP;
  else if (nn == tt || tt <= 1 || nn < tt || tt > M || nn > N) k = STB_KEY_OTHER;
  else {
    const unsigned trip = (nn - 3) >> 3, u = (nn - 3) & 7, sg = (tt - col0) >> 6, ln = (tt - col0) & 63;  // (tt >= 2)
    k = (((uint64_t)trip * nsg + sg) << 9) | (u << 6) | ln;
  }
  key[g] = k;
  payload[g] = (uint32_t)g;
}

// the same for the tiles of k_fill_hb: key = (item << 11) | (row in group << 8) | element of the wave (halo included),
// item = (record index of tile (strip, block)) * NQ + group of H.G rows; row n belongs to block (n - 2) / R
__global__ void k_item_keys_hb(const uint32_t *n, const uint16_t *t, uint64_t G, unsigned N, unsigned M, hb_dot_info H,
                               uint64_t *key, uint32_t *payload) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const unsigned nn = n[g], tt = t[g];
  uint64_t k;
  // (column 1 -- t = 1 -- is a cell too: the last element of strip 0's halo, which that strip's tiles compute along;
  // a pair with t = n contributes log 1 = 0; what is left as "other" has S_S = log 0, lib/stable.c:948-949)
  if (nn <= 1 || nn == tt) k = STB_KEY_SKIP;
  else if (tt == 0 || nn < tt || tt > M || nn > N) k = STB_KEY_OTHER;
  else {
    unsigned j = 0, cw = (unsigned)H.HC - 1u;  // t = 1
    if (tt >= 2) {
      const unsigned e = tt - 2;
      j = e / (unsigned)H.UC;
      cw = (unsigned)H.HC + (e - j * (unsigned)H.UC);
    }
    const unsigned b = (nn - 2) / (unsigned)H.R, r = (nn - 2) - b * (unsigned)H.R;
    const unsigned b0 = (unsigned)(((unsigned long long)j * (unsigned)H.UC) / (unsigned)H.R);
    const unsigned rec = H.rec_off[j + 1] + (b - b0);  // (b >= b0: the cell lies on or below the diagonal)
    const unsigned q = r / (unsigned)H.G;
    k = ((((uint64_t)rec * (unsigned)H.NQ) + q) << 11) | ((uint64_t)(r - q * (unsigned)H.G) << 8) | cw;
  }
  key[g] = k;
  payload[g] = (uint32_t)g;
}

// ... and for the strips of the grid form (grid_hb.hip): key = (item << 13) | (row in group << 8) | element of the wave,
// item = (record index of tile (strip, block)) * NQ + group of G rows.  Column 1 (t = 1) is a cell too -- the last
// element of strip 0's halo, which that strip computes along -- and a pair with t = n contributes log 1 = 0: what is
// left as "other" are the pairs whose S_S is log 0 (lib/stable.c:948-949).
__global__ void k_item_keys_hb2(const uint32_t *n, const uint16_t *t, uint64_t G, unsigned N, unsigned M, hb_dot_info H, int posbits,
                                uint64_t *key, uint32_t *payload) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const unsigned nn = n[g], tt = t[g];
  uint64_t k;
  if (nn <= 1 || nn == tt) k = STB_KEY_SKIP;
  else if (tt == 0 || nn < tt || tt > M || nn > N) k = STB_KEY_OTHER;
  else {
    unsigned j = 0, cw = (unsigned)H.HC - 1u;  // t = 1
    if (tt >= 2) {
      const unsigned e = tt - 2;
      j = e / (unsigned)H.UC;
      cw = (unsigned)H.HC + (e - j * (unsigned)H.UC);
    }
    const unsigned b = (nn - 2) / (unsigned)H.R, r = (nn - 2) - b * (unsigned)H.R;
    const unsigned b0 = (unsigned)(((unsigned long long)j * (unsigned)H.UC) / (unsigned)H.R);
    const unsigned rec = H.rec_off[j + 1] + (b - b0);
    const unsigned q = r / (unsigned)H.G;
    k = ((((uint64_t)rec * (unsigned)H.NQ) + q) << (posbits + 5)) | ((uint64_t)(r - q * (unsigned)H.G) << posbits) | cw;
  }
  key[g] = k;
  payload[g] = (uint32_t)g;
}

// counts[0] = keys below STB_KEY_OTHER (table cells), counts[1] = keys equal to it
__global__ void k_key_bounds(const uint64_t *key, uint64_t G, uint64_t *counts) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  uint64_t lo = 0, hi = G;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) / 2;
    if (key[mid] < STB_KEY_OTHER) lo = mid + 1;
    else hi = mid;
  }
  const uint64_t a = lo;
  hi = G;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) / 2;
    if (key[mid] <= STB_KEY_OTHER) lo = mid + 1;
    else hi = mid;
  }
  counts[0] = a;
  counts[1] = lo - a;
}

__global__ void k_gather_pairs(const uint32_t *n, const uint16_t *t, const uint32_t *idx, uint64_t cnt, uint32_t *n2,
                               uint16_t *t2) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < cnt) {
    n2[g] = n[idx[g]];
    t2[g] = t[idx[g]];
  }
}

__global__ void k_split_runs(const uint64_t *ukey, const unsigned *runs, unsigned short *pos, unsigned *item, int posbits) {
  const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < *runs) {
    pos[r] = (unsigned short)(ukey[r] & ((1u << posbits) - 1u));
    item[r] = (unsigned)(ukey[r] >> posbits);
  }
}

// The grid form's dense layout.  A listed cell is one word, position | count << 13; a (tile, group of rows) holds its
// cells in NW words per lane -- NW the same for the NQ groups of a tile: the most any of them needs -- at an address that
// needs no look-up beyond the tile's own entry of `tinfo` (first word / 64 << 6 | NW), which the walking wave asks for a
// block ahead: a group's cells are coalesced loads issued ahead of their use, whether it holds 20 cells or 2000 (the
// pairs of a 10^4 x 10^4 table thin out like 1 / n: its first 4000 rows hold more than 63 cells per group of 12 x 208).
// NW = 63 marks a tile taken from the CSR lists instead (a count of 2^19 or more, or more words than STB_DENSE_NWMAX).
#define STB_DENSE_NWMAX 40
#define STB_DENSE_CSR 63u
__global__ void k_tile_words(const unsigned *item_ptr, const unsigned *cnt, unsigned n_tiles, unsigned NQ, int wbits, unsigned *tnw, unsigned *twords) {
  const unsigned t = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
  if (t >= n_tiles) return;
  const unsigned e0 = item_ptr[(size_t)t * NQ], e1 = item_ptr[(size_t)t * NQ + NQ];
  unsigned nw = 0;
  for (unsigned q = 0; q < NQ; q++) {
    const unsigned c = item_ptr[(size_t)t * NQ + q + 1] - item_ptr[(size_t)t * NQ + q];
    nw = max(nw, (c + 63) / 64);
  }
  bool big = false;
  for (unsigned e = e0 + lane; e < e1; e += 64) big = big || cnt[e] >= (1u << (32 - wbits));  // (a word is position | count << wbits)
  if (__any(big) || nw > STB_DENSE_NWMAX) nw = STB_DENSE_CSR;
  if (lane == 0) {
    tnw[t] = nw;
    twords[t] = (nw == STB_DENSE_CSR) ? 0u : nw * NQ;
  }
}

__global__ void k_dense_fill(const unsigned *item_ptr, const unsigned short *pos, const unsigned *cnt, unsigned nitems, unsigned NQ, int wbits,
                             const unsigned *tnw, const unsigned *toff, unsigned *dense, unsigned *tinfo) {
  const unsigned i = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
  if (i >= nitems) return;
  const unsigned t = i / NQ, q = i - t * NQ;
  const unsigned nw = tnw[t];
  if (q == 0 && lane == 0) tinfo[t] = (toff[t] << 6) | nw;
  if (nw == STB_DENSE_CSR) return;
  const unsigned e0 = item_ptr[i], e1 = item_ptr[i + 1];
  unsigned *dst = dense + ((size_t)toff[t] + (size_t)q * nw) * 64;
  for (unsigned k = lane; k < nw * 64; k += 64) dst[k] = (e0 + k < e1) ? ((unsigned)pos[e0 + k] | (cnt[e0 + k] << wbits)) : 0u;
}

// item_ptr[i] = first run whose item index is >= i
__global__ void k_item_ptr(const unsigned *item, const unsigned *runs, unsigned nitems, unsigned *ptr) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nitems) return;
  unsigned lo = 0, hi = *runs;
  while (lo < hi) {
    const unsigned mid = (lo + hi) / 2;
    if (item[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  ptr[i] = lo;
}

// partial sums of the fused forms: (column blocks of 64) x 16 waves per table for the chain form, one per tile for the
// halo-block form, two per strip and per helper job for the grid form
int stb_groups_alloc_dotp(stb_groups_t *g) {
  const unsigned N = g->N, M = g->M;
  g->dotp_elems = (size_t)g->Dmax * ((size_t)(M + 63) / 64 + 1) * 16;
  const size_t ckp = stb_launch_ck ? (size_t)g->Dmax * stb_ck_dot_parts(N, M, g->Dmax) : 0;
  if (ckp > g->dotp_elems) g->dotp_elems = ckp;
  hb_dot_info H2;
  if (stb_hb_dot_info(N, M, g->Dmax, &H2, g->hb_sum_C) == 0 && (size_t)g->Dmax * H2.n_tiles > g->dotp_elems) g->dotp_elems = (size_t)g->Dmax * H2.n_tiles;
  // (the self-summing form: two sums per strip, strips of 80 columns at the narrowest)
  size_t hb2 = (size_t)g->Dmax * ((size_t)M / 64 + 8) * 2;
  {
    // (... and two per tile left to helper waves: at most grid_geom::job_cap of them)
    grid_geom gj;
    for (int dd : {1, g->Dmax})
      if (stb_grid_geometry(N, M, dd, &gj) == 0 && (size_t)g->Dmax * ((size_t)gj.JW + gj.job_cap) * 2 > hb2)
        hb2 = (size_t)g->Dmax * ((size_t)gj.JW + gj.job_cap) * 2;
  }
  if (hb2 > g->dotp_elems) g->dotp_elems = hb2;
  if (stb_pool_malloc((void **)&g->d_dotp, sizeof(double) * g->dotp_elems) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
  return 0;
}

// The same choice made ON the device (round 5: a set whose pairs are new gets its lists without a host round trip): one
// workgroup; the tiles' words per group in, the job list in the layout of grid_hb.hip out (queue starts, number of jobs,
// jobs in tile order = strip after strip, a strip's by block) together with every tile's place in it.
__global__ __launch_bounds__(1024) void k_jobs_build(const unsigned *tnw, unsigned n_tiles, const unsigned *tile_off, int JW, int jlim, int R, int UC, int NB,
                                                     unsigned cap, unsigned nwh0, unsigned *jobs, unsigned *tjob) {
  __shared__ unsigned hist[64], s_nwh, s_scan[17], s_off[1024];
  const unsigned tid = threadIdx.x;
  if (tid < 64) hist[tid] = 0;
  // (the strips' first tiles, looked at a dozen times per tile below: in LDS where they fit)
  const bool off_lds = JW + 2 <= 1024;
  if (off_lds)
    for (int i = (int)tid; i < JW + 2; i += 1024) s_off[i] = tile_off[i];
  __syncthreads();
  const unsigned *toffp = off_lds ? s_off : tile_off;
  for (unsigned t = tid; t < n_tiles; t += 1024) atomicAdd(&hist[min(tnw[t], 63u)], 1u);
  __syncthreads();
  if (tid == 0) {
    // the smallest threshold from nwh0 on whose tiles fit (the tiles at or above a threshold grow as it falls): one pass
    // from the top; 64 = none fits
    unsigned nwh = 64, c = 0;
    for (int k = 63; k >= (int)nwh0; k--) {
      c += hist[k];
      if (c > cap) break;
      nwh = (unsigned)k;
    }
    s_nwh = nwh;
  }
  __syncthreads();
  const unsigned nwh = s_nwh;
  // strip of a tile: tile_off[j + 1] is the first tile of strip j (entry 0 holds the total)
  auto strip_of = [&](unsigned t) -> int {
    int lo = 0, hi = JW - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) / 2;
      if (toffp[mid + 1] <= t) lo = mid;
      else hi = mid - 1;
    }
    return lo;
  };
  const int jend = JW < jlim ? JW : jlim;  // strips that may have jobs
  const unsigned per = (n_tiles + 1023u) / 1024u, t0 = tid * per, t1 = min(t0 + per, n_tiles);
  unsigned mine = 0;
  if (nwh <= 63)
    for (unsigned t = t0; t < t1; t++) mine += (tnw[t] >= nwh && strip_of(t) < jend) ? 1u : 0u;
  unsigned total = 0;
  unsigned k = stb_block_exclusive_1024(mine, s_scan, &total);
  for (unsigned t = t0; t < t1; t++) {
    const int j = strip_of(t);
    if (t == toffp[j + 1] && j <= GH_JQ_HOST) jobs[j] = (j < jend) ? k : total;  // the first tile of strip j: its queue starts here
    if (nwh <= 63 && tnw[t] >= nwh && j < jend) {
      const int b = (int)(((long long)j * UC) / R) + (int)(t - toffp[j + 1]);
      jobs[128 + k] = (unsigned)j | ((unsigned)b << 16);
      tjob[t] = k;
      k++;
    } else {
      tjob[t] = 0xffffffffu;
    }
  }
  // queue starts of the strips there are not (or that may have no jobs), and the number of jobs
  for (int j = (int)tid; j <= GH_JQ_HOST; j += 1024)
    if (j >= jend) jobs[j] = total;
  if (tid == 0) jobs[GH_JQ_HOST + 1] = total;
}

// queued on the set's stream behind the kernels that produced `tnw`; no host synchronisation
int stb_lists_jobs_device(stb_groups_t *g, int which, int D, const grid_geom &gg, const unsigned *tnw) {
  const unsigned n_tiles = gg.n_tiles;
  g->n_jobs[which] = 0;
  const unsigned cap = stb_grid_job_cap(gg.C, g->Dmax, n_tiles, gg.phases);
  if (!cap || !n_tiles) {
    if (g->d_jobs[which]) HIPCHK(hipMemsetAsync(g->d_jobs[which], 0, 4 * 128, g->st));
    return 0;
  }
  unsigned nwh = (unsigned)stb_env_int("STB_GRID_HELP_NW", 0);
  if (nwh < 1) {
    const double waves_per_simd = (double)gg.JW * D / (4.0 * stb_cu_count());
    nwh = waves_per_simd <= 1.9 ? 4 : (waves_per_simd <= 2.7 ? 6 : 8);
  }
  int jlim = ((gg.JW - 1) / gg.P) * gg.P;
  if (jlim > 64) jlim = 64;
  if (!g->d_jobs[which] && stb_pool_malloc((void **)&g->d_jobs[which], 4 * ((size_t)n_tiles + 128 + 1)) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
  if (!g->d_tjob[which] && stb_pool_malloc((void **)&g->d_tjob[which], 4 * (size_t)(n_tiles + 1)) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
  hipLaunchKernelGGL(k_jobs_build, dim3(1), dim3(1024), 0, g->st, tnw, n_tiles, g->d_tile_off[which], gg.JW, jlim, gg.R, gg.U * gg.C, gg.NB, cap, nwh,
                     g->d_jobs[which], g->d_tjob[which]);
  HIPCHK(hipGetLastError());
  g->n_jobs[which] = cap;  // (at most; the number itself stays on the device)
  return 0;
}

// The tiles whose listed cells the strip's own wave does not look up: every strip of a table moves at the pace of
// the strips to its left, and those hold most of the pairs (t uniform below n: columns like log) -- several
// passes a group where the average strip has one.  Such a tile is only walked by its strip, which leaves the
// state of its wave before the block as a record; waves whose strips the diagonal has not reached yet take the
// tiles as jobs once their own strips have ended.  Which: all tiles of NWH passes a group and more (the lists in CSR
// form among them), NWH chosen below and raised until the records fit (grid_geom::job_cap at Dmax discounts).
// h_nw: words per group of every tile (host).  The lists go to the device on the set's stream.
int stb_lists_jobs_from(stb_groups_t *g, int which, int D, const grid_geom &gg, const unsigned *h_nw) {
  const unsigned n_tiles = gg.n_tiles;
  g->n_jobs[which] = 0;
  const unsigned cap = stb_grid_job_cap(gg.C, g->Dmax, n_tiles, gg.phases);
  if (!cap || !n_tiles) {
    // (no jobs: a list left by an earlier build must say so -- the kernels read the number of jobs from it)
    if (g->d_jobs[which] && (hipMemsetAsync(g->d_jobs[which], 0, 4 * 128, g->st) != hipSuccess || hipStreamSynchronize(g->st) != hipSuccess))
      return stb_fail("stb_groups_aterms: %s", hipGetErrorString(hipGetLastError()));
    return 0;
  }
  unsigned hist[65] = {0};  // (an upper bound: the tiles of all strips)
  for (unsigned t = 0; t < n_tiles; t++) hist[h_nw[t] > 63 ? 63 : h_nw[t]]++;
  // NWH at the least: what a job takes off the critical path it adds to the chip's work (the tile is walked twice),
  // which the fuller chip can afford less (MI355X, 10^6 pairs, N = 10^4, kernel ms without jobs / NWH = 3 / 4 / 6 /
  // 8 / 12: 32 discounts 1.21 / 1.00 / 0.96 / 1.01 / 1.04 / 1.09, 40: 1.26 / 1.10 / 1.02 / 1.05 / 1.08 / 1.12,
  // 48: 1.26 / 1.30 / 1.14 / 1.09 / 1.10 / 1.15, 64: 1.36 / 1.59 / 1.39 / 1.29 / 1.28 / 1.29)
  unsigned nwh = (unsigned)stb_env_int("STB_GRID_HELP_NW", 0);
  if (nwh < 1) {
    const double waves_per_simd = (double)gg.JW * D / (4.0 * stb_cu_count());
    nwh = waves_per_simd <= 1.9 ? 4 : (waves_per_simd <= 2.7 ? 6 : 8);
  }
  for (; nwh <= 63; nwh++) {
    unsigned c = 0;
    for (unsigned k = nwh; k <= 63; k++) c += hist[k];
    if (c <= cap) break;
  }
  std::vector<unsigned> jobs, tjob(n_tiles, 0xffffffffu), off;
  stb_grid_tile_offsets(gg, off);
  const int UCg = gg.U * gg.C;
  // (a job is taken by a wave of a workgroup further right -- one with a higher ticket, for which the strip's own
  // workgroup is running or through: the strips of a table's last workgroup have no jobs; GH_JQ = 64 queues)
  int jlim = ((gg.JW - 1) / gg.P) * gg.P;
  if (jlim > 64) jlim = 64;
  std::vector<unsigned> qoff(64 + 1, 0u);
  if (nwh <= 63) {
    // strip after strip, a strip's tiles by block: a queue per strip
    for (int j = 0; j < gg.JW && j < jlim; j++) {
      const int b00 = (int)(((long long)j * UCg) / gg.R);
      qoff[j] = (unsigned)jobs.size();
      for (int b = b00; b < gg.NB; b++)
        if (h_nw[off[j + 1] + (unsigned)(b - b00)] >= nwh) {
          tjob[off[j + 1] + (unsigned)(b - b00)] = (unsigned)jobs.size();
          jobs.push_back((unsigned)j | ((unsigned)b << 16));
        }
    }
    for (int j = (gg.JW < jlim ? gg.JW : jlim); j <= 64; j++) qoff[j] = (unsigned)jobs.size();
    if (jobs.size() > cap) {  // (the bound above counted every strip's tiles: cannot happen)
      jobs.clear();
      std::fill(tjob.begin(), tjob.end(), 0xffffffffu);
      std::fill(qoff.begin(), qoff.end(), 0u);
    }
  }
  const size_t nj = jobs.size();
  {
    // the list as the kernels read it (grid_hb.hip: GH_JQ + 1 queue starts, the number of jobs at word 65, jobs from word 128)
    std::vector<unsigned> lay(128, 0u);
    for (int j = 0; j <= 64; j++) lay[(size_t)j] = qoff[(size_t)j];
    lay[65] = (unsigned)nj;
    lay.insert(lay.end(), jobs.begin(), jobs.end());
    jobs.swap(lay);
  }
  // (buffers for the most there can be, kept from set to set; the copies are synchronous: the vectors above go away)
  if (!g->d_jobs[which] && stb_pool_malloc((void **)&g->d_jobs[which], 4 * ((size_t)n_tiles + 128 + 1)) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
  if (!g->d_tjob[which] && stb_pool_malloc((void **)&g->d_tjob[which], 4 * (size_t)(n_tiles + 1)) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
  if (hipMemcpyAsync(g->d_tjob[which], tjob.data(), 4 * (size_t)n_tiles, hipMemcpyHostToDevice, g->st) != hipSuccess ||
      hipMemcpyAsync(g->d_jobs[which], jobs.data(), 4 * jobs.size(), hipMemcpyHostToDevice, g->st) != hipSuccess || hipStreamSynchronize(g->st) != hipSuccess)
    return stb_fail("stb_groups_aterms: %s", hipGetErrorString(hipGetLastError()));
  g->n_jobs[which] = (unsigned)nj;
  return 0;
}

// Returns 0 and sets g->sparse = 1 when the sparse form was built, 0 with g->sparse = 0 when the
// pairs are too dense for it to pay (the caller then builds the count slab), non-zero on error.
static int groups_fused_setup_sparse(stb_groups_t *g, int which, int D) {
  const unsigned N = g->N, M = g->M;
  const uint64_t G = g->G;
  hb_dot_info H;
  memset(&H, 0, sizeof(H));
  if (which == 2 && stb_hb_dot_info(N, M, g->Dmax, &H, g->hb_sum_C)) return stb_fail("stb_groups_aterms: no halo-block geometry for N=%u M=%u", N, M);
  grid_geom gg;
  memset(&gg, 0, sizeof(gg));
  if (which >= 3) {
    // (the strip shape of the grid form depends on the number of discounts: a list per shape)
    if (stb_grid_geometry(N, M, D, &gg)) return stb_fail("stb_groups_aterms: no grid geometry for N=%u M=%u", N, M);
    which = stb_grid_which(gg.C);
    H.R = gg.R;
    H.UC = gg.U * gg.C;
    H.HC = gg.HL * gg.C;
    H.NB = gg.NB;
    H.JW = gg.JW;
    H.NQ = gg.NQ;
    H.G = gg.G;
    H.C = gg.C;
    H.n_tiles = gg.n_tiles;
    H.n_rec = gg.n_tiles;
    if ((g->lists_ready[which] || g->d_tile_off[which]) && (g->list_R[which] != H.R || g->list_G[which] != H.G)) {
      (void)hipStreamSynchronize(g->st);
      void **olds[] = {(void **)&g->d_item_ptr[which], (void **)&g->d_ent_pos[which], (void **)&g->d_ent_cnt[which], (void **)&g->d_tile_off[which],
                       (void **)&g->d_dense[which],    (void **)&g->d_tinfo[which],   (void **)&g->d_jobs[which],    (void **)&g->d_tjob[which],
                       (void **)&g->d_tnw[which],      (void **)&g->d_toff[which]};
      for (void **q : olds) {
        stb_pool_free(*q);
        *q = nullptr;
      }
      g->n_jobs[which] = 0;
      g->ent_cap[which] = 0;
      g->dense_cap[which] = 0;
      g->lists_ready[which] = 0;
    }
    if (!g->lists_ready[which] && !g->d_tile_off[which]) {
      std::vector<unsigned> off;
      stb_grid_tile_offsets(gg, off);
      if (stb_pool_malloc((void **)&g->d_tile_off[which], off.size() * sizeof(unsigned)) != hipSuccess ||
          hipMemcpy(g->d_tile_off[which], off.data(), off.size() * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess)
        return stb_fail("stb_groups_aterms: out of device memory");
    }
    H.rec_off = g->d_tile_off[which];
  }
  if (g->lists_ready[which]) return 0;
  {
    // from the count slab (lists.hip): no sort, no host round trip; where it does not apply, the sort below
    const int rc = stb_lists_slab_build(g, which, D, H, gg);
    if (rc != 2) return rc;
  }
  if (!g->fused_ready) g->sparse = 0;
  if (G == 0 || G >= 0xffffffffull) return 0;
  const unsigned nsg = (M + 63) / 64 + 4;
  const unsigned trips = (N - 2 + 7) / 8;
  const uint64_t nitems64 = (which >= 2) ? (uint64_t)H.n_rec * (unsigned)H.NQ : (uint64_t)trips * nsg;
  if (nitems64 >= (1ull << 31)) return 0;
  const unsigned nitems = (unsigned)nitems64;
  uint64_t *k0 = nullptr, *k1 = nullptr, *uk = nullptr, *d_counts = nullptr;
  uint32_t *p0 = nullptr, *p1 = nullptr;
  unsigned *cnt = nullptr, *runs = nullptr, *item = nullptr;
  void *tmp = nullptr;
  int rc = 1;
  const unsigned blocks = (unsigned)((G + 255) / 256);
  do {
    if (stb_pool_malloc((void **)&k0, 8 * G) != hipSuccess || stb_pool_malloc((void **)&k1, 8 * G) != hipSuccess ||
        stb_pool_malloc((void **)&p0, 4 * G) != hipSuccess || stb_pool_malloc((void **)&p1, 4 * G) != hipSuccess ||
        stb_pool_malloc((void **)&uk, 8 * G) != hipSuccess || stb_pool_malloc((void **)&cnt, 4 * G) != hipSuccess ||
        stb_pool_malloc((void **)&item, 4 * G) != hipSuccess || stb_pool_malloc((void **)&runs, 64) != hipSuccess ||
        stb_pool_malloc((void **)&d_counts, 64) != hipSuccess) {
      stb_fail("stb_groups_aterms: out of device memory");
      break;
    }
    if (which >= 3)
      hipLaunchKernelGGL(k_item_keys_hb2, dim3(blocks), dim3(256), 0, g->st, g->d_n, g->d_t, G, N, M, H, stb_pos_bits(H.C), k0, p0);
    else if (which == 2)
      hipLaunchKernelGGL(k_item_keys_hb, dim3(blocks), dim3(256), 0, g->st, g->d_n, g->d_t, G, N, M, H, k0, p0);
    else
      hipLaunchKernelGGL(k_item_keys, dim3(blocks), dim3(256), 0, g->st, g->d_n, g->d_t, G, N, M, nsg, which ? 2u : 1u, k0, p0);
    size_t b1 = 0, b2 = 0;
    if (rocprim::radix_sort_pairs(nullptr, b1, k0, k1, p0, p1, (size_t)G, 0, 64, g->st) != hipSuccess) break;
    if (rocprim::run_length_encode(nullptr, b2, k1, (unsigned)G, uk, cnt, runs, g->st) != hipSuccess) break;
    const size_t tmp_bytes = b1 > b2 ? b1 : b2;
    if (stb_pool_malloc(&tmp, tmp_bytes ? tmp_bytes : 1) != hipSuccess) {
      stb_fail("stb_groups_aterms: out of device memory");
      break;
    }
    size_t bb = tmp_bytes;
    if (rocprim::radix_sort_pairs(tmp, bb, k0, k1, p0, p1, (size_t)G, 0, 64, g->st) != hipSuccess) break;
    hipLaunchKernelGGL(k_key_bounds, dim3(1), dim3(1), 0, g->st, k1, G, d_counts);
    uint64_t h_counts[2] = {0, 0};
    if (hipMemcpyAsync(h_counts, d_counts, 16, hipMemcpyDeviceToHost, g->st) != hipSuccess ||
        hipStreamSynchronize(g->st) != hipSuccess)
      break;
    const uint64_t n_in = h_counts[0], n_other = h_counts[1];
    // dense enough that every cell's log might as well be computed: leave it to the count slab
    if (n_in * 3 > stb_table_cells(N, M)) {
      rc = 0;
      break;
    }
    // the pairs outside the table, in their sorted order (the same whichever layout is built first; the
    // self-summing form has none but those whose S_S is log 0, which it only counts)
    if (which >= 2) {
      g->n_inf = n_other;
    } else if (!g->d_n2) {
      g->G2 = n_other;
      if (stb_pool_malloc((void **)&g->d_n2, 4 * (n_other ? n_other : 1)) != hipSuccess ||
          stb_pool_malloc((void **)&g->d_t2, 2 * (n_other ? n_other : 1)) != hipSuccess) {
        stb_fail("stb_groups_aterms: out of device memory");
        break;
      }
      if (n_other)
        hipLaunchKernelGGL(k_gather_pairs, dim3((unsigned)((n_other + 255) / 256)), dim3(256), 0, g->st, g->d_n, g->d_t,
                           p1 + n_in, n_other, g->d_n2, g->d_t2);
    }
    // distinct cells with their counts
    unsigned h_runs = 0;
    if (n_in) {
      bb = tmp_bytes;
      if (rocprim::run_length_encode(tmp, bb, k1, (unsigned)n_in, uk, cnt, runs, g->st) != hipSuccess) break;
      if (hipMemcpyAsync(&h_runs, runs, 4, hipMemcpyDeviceToHost, g->st) != hipSuccess ||
          hipStreamSynchronize(g->st) != hipSuccess)
        break;
    } else if (hipMemsetAsync(runs, 0, 4, g->st) != hipSuccess) {
      break;
    }
    {
      void **olds[] = {(void **)&g->d_ent_pos[which], (void **)&g->d_ent_cnt[which], (void **)&g->d_item_ptr[which], (void **)&g->d_dense[which],
                       (void **)&g->d_tinfo[which], (void **)&g->d_jobs[which], (void **)&g->d_tjob[which]};
      for (void **q : olds) {
        stb_pool_free(*q);
        *q = nullptr;
      }
      g->ent_cap[which] = 0;
      g->dense_cap[which] = 0;
    }
    if (stb_pool_malloc((void **)&g->d_ent_pos[which], 2 * (size_t)(h_runs ? h_runs : 1)) != hipSuccess ||
        stb_pool_malloc((void **)&g->d_ent_cnt[which], 4 * (size_t)(h_runs ? h_runs : 1)) != hipSuccess ||
        stb_pool_malloc((void **)&g->d_item_ptr[which], 4 * ((size_t)nitems + 2)) != hipSuccess) {
      stb_fail("stb_groups_aterms: out of device memory");
      break;
    }
    if (h_runs) {
      hipLaunchKernelGGL(k_split_runs, dim3((h_runs + 255) / 256), dim3(256), 0, g->st, uk, runs, g->d_ent_pos[which], item,
                         which >= 3 ? stb_pos_bits(H.C) + 5 : (which == 2 ? 11 : 9));
      if (hipMemcpyAsync(g->d_ent_cnt[which], cnt, 4 * (size_t)h_runs, hipMemcpyDeviceToDevice, g->st) != hipSuccess) break;
    }
    hipLaunchKernelGGL(k_item_ptr, dim3((nitems + 1 + 255) / 256), dim3(256), 0, g->st, item, runs, nitems, g->d_item_ptr[which]);
    if (which >= 3) {
      // the dense layout the walk reads: words per tile, their prefix sum, the words
      const unsigned n_tiles = H.n_tiles, NQ = (unsigned)H.NQ;
      unsigned *tnw = nullptr, *twords = nullptr, *toff = nullptr;
      void *tmp2 = nullptr;
      size_t b3 = 0;
      bool ok = stb_pool_malloc((void **)&tnw, 4 * (size_t)(n_tiles + 1)) == hipSuccess &&
                stb_pool_malloc((void **)&twords, 4 * (size_t)(n_tiles + 1)) == hipSuccess &&
                stb_pool_malloc((void **)&toff, 4 * (size_t)(n_tiles + 1)) == hipSuccess &&
                stb_pool_malloc((void **)&g->d_tinfo[which], 4 * (size_t)(n_tiles + 1)) == hipSuccess;
      unsigned h_last[2] = {0, 0};
      if (ok && n_tiles) {
        hipLaunchKernelGGL(k_tile_words, dim3((n_tiles + 3) / 4), dim3(256), 0, g->st, g->d_item_ptr[which], g->d_ent_cnt[which], n_tiles, NQ,
                           stb_pos_bits(H.C) + 5, tnw, twords);
        ok = rocprim::exclusive_scan(nullptr, b3, twords, toff, 0u, (size_t)n_tiles, rocprim::plus<unsigned>(), g->st) == hipSuccess &&
             stb_pool_malloc(&tmp2, b3 ? b3 : 1) == hipSuccess &&
             rocprim::exclusive_scan(tmp2, b3, twords, toff, 0u, (size_t)n_tiles, rocprim::plus<unsigned>(), g->st) == hipSuccess &&
             hipMemcpyAsync(&h_last[0], toff + (n_tiles - 1), 4, hipMemcpyDeviceToHost, g->st) == hipSuccess &&
             hipMemcpyAsync(&h_last[1], twords + (n_tiles - 1), 4, hipMemcpyDeviceToHost, g->st) == hipSuccess &&
             hipStreamSynchronize(g->st) == hipSuccess;
      }
      const size_t words = (size_t)h_last[0] + h_last[1];  // in units of 64 dwords
      ok = ok && words < (1u << 26);                       // (a tile's first word / 64 goes into 26 bits of its entry)
      ok = ok && stb_pool_malloc((void **)&g->d_dense[which], 256 * (words ? words : 1)) == hipSuccess;
      if (ok && nitems)
        hipLaunchKernelGGL(k_dense_fill, dim3((nitems + 3) / 4), dim3(256), 0, g->st, g->d_item_ptr[which], g->d_ent_pos[which],
                           g->d_ent_cnt[which], nitems, NQ, stb_pos_bits(H.C) + 5, tnw, toff, g->d_dense[which], g->d_tinfo[which]);
      if (ok) ok = hipStreamSynchronize(g->st) == hipSuccess;
      // The tiles whose listed cells the strip's own wave does not look up: every strip of a table moves at the pace of
      // the strips to its left, and those hold most of the pairs (t uniform below n: columns like log) -- several
      // passes a group where the average strip has one.  Such a tile is only walked by its strip, which leaves the
      // state of its wave before the block as a record; waves whose strips the diagonal has not reached yet take the
      // tiles as jobs once their own strips have ended.  Which: all tiles of NWH passes a group and more (the lists in CSR
      // form among them), NWH chosen below and raised until the records fit (grid_geom::job_cap at Dmax discounts).
      g->n_jobs[which] = 0;
      if (ok && n_tiles) {
        std::vector<unsigned> h_nw(n_tiles);
        ok = hipMemcpy(h_nw.data(), tnw, 4 * (size_t)n_tiles, hipMemcpyDeviceToHost) == hipSuccess && stb_lists_jobs_from(g, which, D, gg, h_nw.data()) == 0;
      }
      stb_pool_free(tnw);
      stb_pool_free(twords);
      stb_pool_free(toff);
      stb_pool_free(tmp2);
      if (!ok) {
        stb_fail("stb_groups_aterms: out of device memory (or a dense list beyond 2^32 bytes)");
        break;
      }
    }
    if (!g->d_dotp && stb_groups_alloc_dotp(g)) break;
    if (hipStreamSynchronize(g->st) != hipSuccess || hipGetLastError() != hipSuccess) break;
    g->nsg = nsg;
    g->sparse = 1;
    g->lists_ready[which] = 1;
    g->list_R[which] = H.R;
    g->list_G[which] = H.G;
    g->fused_ready = 1;
    rc = 0;
  } while (0);
  if (rc != 0) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) stb_fail("stb_groups_aterms: sparse set-up failed: %s", hipGetErrorString(e));
  }
  (void)hipStreamSynchronize(g->st);
  stb_pool_free(k0);
  stb_pool_free(k1);
  stb_pool_free(p0);
  stb_pool_free(p1);
  stb_pool_free(uk);
  stb_pool_free(cnt);
  stb_pool_free(item);
  stb_pool_free(runs);
  stb_pool_free(d_counts);
  stb_pool_free(tmp);
  return rc;
}

// Lazy set-up of the fused evaluation (first call with more than one discount): occurrence count
// per table cell, and the pairs that address no cell.  The pairs come back from the device in their
// sorted order, so the result does not depend on the order the caller supplied them in.
static int groups_fused_setup(stb_groups_t *g) {
  const unsigned N = g->N, M = g->M;
  const uint64_t G = g->G;
  const uint64_t elems = stb_table_elems(N, M);
  HIPCHK(stb_pool_malloc((void **)&g->d_cnt, sizeof(unsigned) * (elems ? elems : 1)));
  HIPCHK(hipMemsetAsync(g->d_cnt, 0, sizeof(unsigned) * (elems ? elems : 1), g->st));
  if (G)
    hipLaunchKernelGGL(k_count_pairs, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, g->st, g->d_n, g->d_t, G, N,
                       M, g->d_cnt);
  uint32_t *hn = (uint32_t *)malloc(sizeof(uint32_t) * (G ? G : 1));
  uint16_t *ht = (uint16_t *)malloc(sizeof(uint16_t) * (G ? G : 1));
  if (!hn || !ht) {
    free(hn);
    free(ht);
    return stb_fail("stb_groups_aterms: out of host memory");
  }
  hipError_t e1 = hipSuccess, e2 = hipSuccess;
  if (G) {
    e1 = hipMemcpyAsync(hn, g->d_n, sizeof(uint32_t) * G, hipMemcpyDeviceToHost, g->st);
    e2 = hipMemcpyAsync(ht, g->d_t, sizeof(uint16_t) * G, hipMemcpyDeviceToHost, g->st);
  }
  if (e1 == hipSuccess && e2 == hipSuccess) e1 = hipStreamSynchronize(g->st);
  uint64_t G2 = 0;
  if (e1 == hipSuccess && e2 == hipSuccess) {
    for (uint64_t q = 0; q < G; q++) {  // compact in place: the pairs outside the table
      const unsigned nn = hn[q], tt = ht[q];
      if (nn > 1 && (nn == tt || tt <= 1 || nn < tt || tt > M || nn > N)) {
        hn[G2] = nn;
        ht[G2] = (uint16_t)tt;
        G2++;
      }
    }
    g->G2 = G2;
    e1 = stb_pool_malloc((void **)&g->d_n2, sizeof(uint32_t) * (G2 ? G2 : 1));
    if (e1 == hipSuccess) e2 = stb_pool_malloc((void **)&g->d_t2, sizeof(uint16_t) * (G2 ? G2 : 1));
    if (e1 == hipSuccess && e2 == hipSuccess && G2) {
      e1 = hipMemcpy(g->d_n2, hn, sizeof(uint32_t) * G2, hipMemcpyHostToDevice);
      e2 = hipMemcpy(g->d_t2, ht, sizeof(uint16_t) * G2, hipMemcpyHostToDevice);
    }
  }
  free(hn);
  free(ht);
  if (e1 != hipSuccess || e2 != hipSuccess) return stb_fail("stb_groups_aterms: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  // partial sums: at most (column blocks of 64) x 16 waves per table
  g->dotp_elems = (size_t)g->Dmax * ((size_t)(M + 63) / 64 + 1) * 16;
  HIPCHK(stb_pool_malloc((void **)&g->d_dotp, sizeof(double) * g->dotp_elems));
  HIPCHK(hipGetLastError());
  g->fused_ready = 1;
  return 0;
}

// one evaluation in two halves: queue everything on the set's stream, the sums ending in pinned host
// memory; then wait, check the fill and hand the values over.  aterms_finish returns 0, 1 (error) or 2
// (the fused chain fill gave up waiting: the caller repeats the evaluation through stored tables).
// The lean flow of a fused evaluation in the halo-block (tile workers sum) or the grid form (walking waves sum):
//   launch 1  restaurant partial sums; the abscissae go to the device as kernel arguments
//   launch 2  the table walk (its workspace is zero already: the previous evaluation left it so)
//   launch 3  k_eval_tail: every reduction, the total and the walk's error words to pinned host memory
//   then, behind what the host waits for, the workspace is zeroed for the next evaluation.
// ONE wait (an event after launch 3), no copy in either direction, no S1 vector, no gather pass.
// fused evaluations that gave up waiting and were repeated through stored tables (process-wide; stb_groups_fallbacks)
static std::atomic<unsigned> g_fused_giveups{0};
extern "C" unsigned stb_groups_fallbacks(void) { return g_fused_giveups.load(); }

static int aterms_issue_lean(stb_groups_t *g, const double *x_host, int D, double *out_host, int which, bool timed) {
  g->pending = 0;
  char *ws0 = (char *)g->d_ws_fill;
  double *a_dev = (double *)ws0;
  char *ws = ws0 + STB_WS_FORM;
  const size_t ws_left = g->ws_fill - STB_WS_FORM;
  if (timed) HIPCHK(hipEventRecord(g->ev[0], g->st));
  const dd_t *tpart = nullptr;
  int nbt = 0;
  if (stb_restaurant_partials(x_host, D, g->d_T, g->d_bpar, (uint64_t)g->I, g->d_ws_terms, g->ws_terms, a_dev, &tpart, &nbt, g->st))
    return 1;
  dot_request req;
  req.item_ptr = g->d_item_ptr[which];
  req.ent_pos = g->d_ent_pos[which];
  req.ent_cnt = g->d_ent_cnt[which];
  req.nsg = g->nsg;
  req.col0 = which >= 3 ? 4 : 3;
  if (which == 2) req.geom_C = g->hb_sum_C;  // (the strip shape the set's lists were built for)
  if (which >= 3) {
    req.geom_C = stb_which_C(which);
    req.geom_R = g->list_R[which];
    req.geom_G = g->list_G[which];
    req.tile_off = g->d_tile_off[which];
    req.dense = g->d_dense[which];
    req.tinfo = g->d_tinfo[which];
    req.jobs = g->d_jobs[which];
    req.tjob = g->d_tjob[which];
  }
  req.dotp = g->d_dotp;
  req.dotp_cap = g->dotp_elems;
  req.ws_zero = g->ws_zero;
  req.no_s1 = 1;
  fill_args A;
  memset(&A, 0, sizeof(A));
  A.a = a_dev;
  A.N = g->N;
  A.M = g->M;
  A.tables = g->d_tables;
  A.tstride = g->tstride;
  A.S1 = g->d_S1;
  A.s1stride = g->N;
  if (stb_logtab(&A.lt)) return 1;
  unsigned *hdr = nullptr;
  g->ws_zero = 0;  // (whatever happens below, the workspace is no longer known to be zero)
  g->pend_seq = 0.0;
  if (D == 1 && !timed && stb_env_int("STB_SPIN_WAIT", 1)) {
    g->seq += 1.0;
    g->pend_seq = g->seq;
    ((volatile double *)g->h_out)[2 * g->Dmax + 2] = 0.0;
  }
  if (which >= 3 ? stb_launch_grid(A, D, ws, ws_left, &req, &hdr, g->st) : stb_launch_hb(A, D, ws, ws_left, &req, &hdr, g->st)) return 1;
  // (the partial sums' room was checked inside the launch functions, before anything was queued: dot_request::dotp_cap)
  if (timed) HIPCHK(hipEventRecord(g->ev[1], g->st));
  if (timed) HIPCHK(hipEventRecord(g->ev[2], g->st));
  hipLaunchKernelGGL(k_eval_tail, dim3(D), dim3(256), 0, g->st, g->d_dotp, req.parts_per_table, req.parts_extra_dev, which >= 3 ? 2 : 1, tpart, nbt,
                     (unsigned long long)g->n_inf, g->ent_cap[which] ? g->d_ninf : nullptr, hdr, g->d_out, g->h_out_dev, g->Dmax, g->pend_user,
                     g->pend_seq);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(g->ev[3], g->st));
  // zero for the next evaluation, behind the event the host waits for
  if (req.zero_bytes) {
    HIPCHK(hipMemsetAsync(ws, 0, req.zero_bytes, g->st));
    g->ws_zero = req.zero_bytes;
  }
  g->pending = 1;
  g->pend_lean = 1;
  g->pend_D = D;
  g->pend_fuse = 1;
  g->pend_v = STB_FILL_HB;
  g->pend_out = out_host;
  memcpy(g->pend_x, x_host, sizeof(double) * D);
  return 0;
}

static int aterms_issue(stb_groups_t *g, const double *x_host, int D, double *out_host, bool fuse, int v, bool timed = true) {
  if (fuse && v == STB_FILL_HB && g->sparse && g->sel_which >= 2)
    return aterms_issue_lean(g, x_host, D, out_host, g->sel_which, timed);
  g->pend_lean = 0;
  g->ws_zero = 0;  // (this flow's fills zero and use the workspace themselves)
  if (ensure_tables(g)) return 1;
  if (!fuse && stb_groups_sort_pairs(g)) return 1;  // (the gather's sum is then independent of the caller's order)
  // (what is left for this flow: the chain form, v = STB_FILL_CK with fuse: the summing checkpointed form and its cell
  // lists, and every evaluation through stored tables)
  const int which = (fuse && v == STB_FILL_CK) ? 1 : 0;
  g->pending = 0;
  HIPCHK(hipEventRecord(g->ev[0], g->st));
  if (fuse) {
    // the chain form as a DOT kernel: sum over table cells of count * log S, no table in memory;
    // then the few pairs that address no cell (t = 1 -> S1, t = n -> 0, out of bounds -> -inf)
    dot_request req;
    if (g->sparse) {
      req.item_ptr = g->d_item_ptr[which];
      req.ent_pos = g->d_ent_pos[which];
      req.ent_cnt = g->d_ent_cnt[which];
      req.nsg = g->nsg;
      req.col0 = which + 1;
    } else {
      req.cnt = g->d_cnt;
    }
    req.dotp = g->d_dotp;
    req.dotp_cap = g->dotp_elems;
    stb_set_dot_request(&req);
    const int rc = stb_fill_S(x_host, D, g->N, g->M, g->d_tables, g->tstride, g->d_S1, g->N, g->d_ws_fill,
                              g->ws_fill, which ? STB_FILL_CK : STB_FILL_CHAIN, g->st);
    stb_set_dot_request(nullptr);
    if (rc) return 1;
    stb_fill_last(&g->pend_fill);
    if ((size_t)D * req.parts_per_table > g->dotp_elems) return stb_fail("stb_groups_aterms: partial-sum buffer too small");
    HIPCHK(hipEventRecord(g->ev[1], g->st));
    if (stb_sweep_S(g->d_tables, g->tstride, g->d_S1, g->N, D, g->N, g->M, g->d_n2, g->d_t2, g->G2,
                    g->d_out, g->d_ws_sweep, g->ws_sweep, g->st))
      return 1;
    hipLaunchKernelGGL(k_dot_reduce, dim3(D), dim3(256), 0, g->st, g->d_dotp, req.parts_per_table, g->d_out);
  } else {
    if (stb_fill_S(x_host, D, g->N, g->M, g->d_tables, g->tstride, g->d_S1, g->N, g->d_ws_fill,
                   g->ws_fill, v, g->st))
      return 1;
    stb_fill_last(&g->pend_fill);
    HIPCHK(hipEventRecord(g->ev[1], g->st));
    if (stb_sweep_S(g->d_tables, g->tstride, g->d_S1, g->N, D, g->N, g->M, g->d_n, g->d_t, g->G,
                    g->d_out, g->d_ws_sweep, g->ws_sweep, g->st))
      return 1;
  }
  HIPCHK(hipEventRecord(g->ev[2], g->st));
  if (stb_restaurant_terms(x_host, D, g->d_T, g->d_bpar, (uint64_t)g->I, g->d_out + g->Dmax,
                           g->d_ws_terms, g->ws_terms, g->st))
    return 1;
  HIPCHK(hipEventRecord(g->ev[3], g->st));
  HIPCHK(hipMemcpyAsync(g->h_out, g->d_out, sizeof(double) * 2 * g->Dmax, hipMemcpyDeviceToHost, g->st));
  g->pending = 1;
  g->pend_D = D;
  g->pend_fuse = fuse ? 1 : 0;
  g->pend_v = v;
  g->pend_out = out_host;
  memcpy(g->pend_x, x_host, sizeof(double) * D);
  return 0;
}

static int aterms_finish(stb_groups_t *g, float *ms_fill, float *ms_sweep, float *ms_terms) {
  if (!g->pending) return stb_fail("stb_groups_wait: nothing queued");
  g->pending = 0;
  const int D = g->pend_D;
  if (g->pend_lean) {
    bool there = false;
    if (g->pend_seq != 0.0 && !ms_fill && !ms_sweep && !ms_terms) {
      volatile double *flag = g->h_out + 2 * g->Dmax + 2;
      for (unsigned n = 0; n < 4000000u && *flag != g->pend_seq; n++) __builtin_ia32_pause();  // (bounded: the event below otherwise)
      there = *flag == g->pend_seq;
    }
    if (!there) HIPCHK(hipEventSynchronize(g->ev[3]));
    const unsigned code = (unsigned)g->h_out[2 * g->Dmax], detail = (unsigned)g->h_out[2 * g->Dmax + 1];
    if (code != 0) {
      stb_fail("stb_groups_aterms: the fused evaluation gave up waiting for a neighbour block (code 0x%x, block %u of table %u)", code,
               detail & 0xffffu, detail >> 16);
      g_fused_giveups.fetch_add(1u);
      return 2;  // (the caller repeats the evaluation through stored tables)
    }
    for (int d = 0; d < D; d++) g->pend_out[d] = g->h_out[d];
    // (the walk's own span against what D chains of N rows should take: stb_note_span, abi.hip)
    stb_note_span(g->h_out[2 * g->Dmax + 3] * 1e-5, 1e-3 * (100.0 + 0.05 * (double)g->N * (1.0 + (double)D / 16.0)), "a fused evaluation (stb_groups_aterms)");
    if (ms_fill) HIPCHK(hipEventElapsedTime(ms_fill, g->ev[0], g->ev[1]));
    if (ms_sweep) HIPCHK(hipEventElapsedTime(ms_sweep, g->ev[1], g->ev[2]));
    if (ms_terms) HIPCHK(hipEventElapsedTime(ms_terms, g->ev[2], g->ev[3]));
    return 0;
  }
  HIPCHK(hipStreamSynchronize(g->st));
  if (stb_fill_status_of(&g->pend_fill)) {
    if (g->pend_fuse) g_fused_giveups.fetch_add(1u);
    return g->pend_fuse ? 2 : 1;
  }
  // (whether THIS fill was repeated in the producer/consumer form -- not a counter of the calling thread, which other
  // sets' fills move too and which the waiting thread need not share with the issuing one; a fused evaluation stores no
  // table to sweep again)
  if (g->pend_fill.fell_back && !g->pend_fuse) {
    // the one-launch fill gave up and was repeated with the producer/consumer form: the sum above
    // read unfinished tables, take it again
    if (stb_sweep_S(g->d_tables, g->tstride, g->d_S1, g->N, D, g->N, g->M, g->d_n, g->d_t, g->G,
                    g->d_out, g->d_ws_sweep, g->ws_sweep, g->st))
      return 1;
    HIPCHK(hipMemcpyAsync(g->h_out, g->d_out, sizeof(double) * g->Dmax, hipMemcpyDeviceToHost, g->st));
    HIPCHK(hipStreamSynchronize(g->st));
  }
  for (int d = 0; d < D; d++) g->pend_out[d] = g->h_out[g->Dmax + d] + g->h_out[d];
  if (ms_fill) HIPCHK(hipEventElapsedTime(ms_fill, g->ev[0], g->ev[1]));
  if (ms_sweep) HIPCHK(hipEventElapsedTime(ms_sweep, g->ev[1], g->ev[2]));
  if (ms_terms) HIPCHK(hipEventElapsedTime(ms_terms, g->ev[2], g->ev[3]));
  return 0;
}

static int aterms_once(stb_groups_t *g, const double *x_host, int D, double *out_host, bool fuse, int v,
                       float *ms_fill, float *ms_sweep, float *ms_terms) {
  if (aterms_issue(g, x_host, D, out_host, fuse, v, ms_fill || ms_sweep || ms_terms)) return 1;
  return aterms_finish(g, ms_fill, ms_sweep, ms_terms);
}

// The summing halo-block form (tile workers sum) holds while its spine workgroups leave the workers a quarter of the
// chip: 25/32 of the compute units (200 of an MI355X's 256: 28 discounts of 10^4 columns at 7 strips a workgroup;
// measured 28 discounts 0.99 ms against the grid form's 1.21, 30: 1.55, 32: 2.03 against 1.23).  STB_ATERMS_HB_MAX_SPINE overrides.
static unsigned hb_max_spine() {
  const int e = stb_env_int("STB_ATERMS_HB_MAX_SPINE", 0);
  return e > 0 ? (unsigned)e : (unsigned)(stb_cu_count() * 25 / 32);
}

// which form an evaluation of D discounts takes, with the one-off set-up of the fused form done
static int aterms_prepare(stb_groups_t *g, int D, bool allow_fuse, bool *fuse_out, int *v_out) {
  if (!g) return stb_fail("stb_groups_aterms: null group set");
  if (!g->have_bounds || (g->G && !g->have_pairs)) return stb_fail("stb_groups_aterms: the set has no pairs yet (stb_groups_pairs_begin / _put / _commit)");
  if (g->putting) return stb_fail("stb_groups_aterms: stb_groups_pairs_commit has not been called");
  if (D < 1 || D > g->Dmax) return stb_fail("stb_groups_aterms: D=%d outside 1..%d", D, g->Dmax);
  if (g->pending == 1) return stb_fail("stb_groups_aterms: an evaluation queued with stb_groups_aterms_async has not been waited for");
  const int v = stb_default_variant();
  // one discount: the gather over a stored table needs no set-up; a grid: fused -- and so is a single discount once
  // the set is known to be used again and again (stb_groups_update_restaurants has been called on it: samplea's kept
  // set), where the one-off set-up is paid back: an evaluation is then three launches and one wait whatever D is
  // (round 5: the lists of the halo-block and grid forms come from a count slab in four launches -- lists.hip -- so a
  // single discount on a fresh set is fused too wherever those forms apply; STB_ATERMS_FUSE1=0 restores the old rule)
  bool fuse1 = g->reused;
  if (!fuse1 && D == 1 && stb_env_int("STB_ATERMS_FUSE1", 1) && stb_env_int("STB_LISTS_SLAB", 1) && g->G * 3 <= stb_table_cells(g->N, g->M)) {
    hb_dot_info H1;
    fuse1 = stb_env_int("STB_ATERMS_HB", 1) && stb_hb_dot_info(g->N, g->M, 1, &H1, g->hb_sum_C) == 0;
  }
  // (a GPU shared with other processes: stored tables and the gather, no waits between workgroups -- abi.hip)
  const bool fuse = allow_fuse && !stb_shared_gpu() && g->fused && (D >= 2 || fuse1) &&
                    (v == STB_FILL_SCALED || v == STB_FILL_CHAIN || v == STB_FILL_CK || v == STB_FILL_HB);
  // The summing fill also exists in the checkpointed form (recurrence-only spine + tile workers that walk
  // a tile again and sum its listed cells; STB_ATERMS_CK=1, or variant STB_FILL_CK), usable while its spine
  // workgroups all fit on the chip.  It is not the default: MI355X, 10^6 pairs, N = M = 10^4, its 0.66 ms
  // at 2-4 discounts and 0.85-0.96 at 8 against the chain form's 0.73-0.77 and 0.87-0.97 (tools/ablation/time_grid.py)
  // -- the tiles' requests for edges, checkpoints and cell lists stretch the spine's hand-offs between
  // workgroups from 2 to ~17 us, which eats what the lighter spine gains.
  int which = 0;
  if (fuse && stb_launch_ck && (stb_env_int("STB_ATERMS_CK", 0) || v == STB_FILL_CK) && v != STB_FILL_CHAIN && stb_ck_eligible(g->N, g->M, D) &&
      stb_ck_dot_spine(g->N, g->M, D) <= (unsigned)stb_env_int("STB_ATERMS_CK_MAX_SPINE", 208))
    which = 1;
  // ... and in the halo-block form (a spine that walks blocks of rows alone + tile workers that sum their
  // tiles' listed cells): the default for a grid while the chip holds its spine -- 49 strips per table of 10^4
  // columns, four to a workgroup up to 17 discounts, seven beyond: up to 24 discounts (1200 strips in all); STB_ATERMS_HB=0 switches
  // it off.  (MI355X, tools/ablation/time_grid.py, 10^6 pairs, N = M = 10^4, kernel / wall ms: 2 discounts 0.36 / 0.45
  // against 0.76 / 0.83 chain, 4: 0.36 / 0.45 against 0.78 / 0.86, 8: 0.44 / 0.53 against 0.89 / 0.97, 16: 0.76 /
  // 0.86 against 0.98 / 1.07, 24: 1.07 / 1.15 against 1.25 / 1.34, 28: 1.22 against 1.32, 30: 1.55 against 1.37,
  // 64: 2.7 against 2.4.)
  // ... and with the walking waves themselves summing their strips' listed cells, no tile workers at all (grid_hb.hip):
  // what a grid beyond the halo-block form's range takes instead of the chain form while the pairs are sparse in the
  // table (its look-ups take one listed cell per lane and group of rows: 10^6 pairs over a 10^4 x 10^4 table, 2 % of
  // the cells, fill them; 12 % -- the same pairs with n < 4000 -- overflow every group).  STB_ATERMS_GRID=1 / 0 forces
  // it on (wherever its geometry exists) / off.  (MI355X, tools/ab_grid.py, 10^6 pairs, N = M = 10^4, kernel ms: 32
  // discounts 1.44 = chain, 48: 1.56 against 1.88, 64: 1.62 against 2.39; N = M = 4000, 64 discounts: 1.35 against 0.57.)
  if (fuse && which == 0 && (v == STB_FILL_HB || v == STB_FILL_SCALED)) {
    const int force = stb_env_int("STB_ATERMS_GRID", -1);
    grid_geom gg;
    if (force != 0 && stb_grid_geometry(g->N, g->M, D, &gg) == 0) {
      hb_dot_info H;
      const bool hb_range = stb_hb_dot_info(g->N, g->M, D, &H, g->hb_sum_C) == 0 && H.n_spine <= hb_max_spine();
      const bool sparse_pairs = (double)g->G <= 0.04 * (double)stb_table_cells(g->N, g->M);
      if (force > 0 || (!hb_range && sparse_pairs && stb_env_int("STB_ATERMS_HB", 1))) which = stb_grid_which(gg.C);
    }
  }
  if (fuse && which == 0 && (v == STB_FILL_HB || (v == STB_FILL_SCALED && stb_env_int("STB_ATERMS_HB", 1)))) {
    hb_dot_info H;
    if (stb_hb_dot_info(g->N, g->M, D, &H, g->hb_sum_C) == 0 && H.n_spine <= hb_max_spine()) which = 2;
  }
  if (fuse && stb_env_int("STB_ATERMS_SPARSE", 1) && (!g->fused_ready || g->sparse) && groups_fused_setup_sparse(g, which, D)) return 1;
  if (fuse && !g->fused_ready && groups_fused_setup(g)) return 1;
  if (fuse && !g->sparse) which = 0;  // (dense pair sets: the count slab, chain form only)
  bool fuse2 = fuse;
  if (fuse && g->sparse && !g->lists_ready[which]) fuse2 = false;  // (no list in this layout: stored tables + gather)
  g->sel_which = which;
  *fuse_out = fuse2;
  *v_out = !fuse2 ? v : (which >= 2 ? STB_FILL_HB : (which ? STB_FILL_CK : STB_FILL_CHAIN));
  return 0;
}

static int groups_aterms(stb_groups_t *g, const double *x_host, int D, double *out_host, bool allow_fuse, float *ms_fill,
                         float *ms_sweep, float *ms_terms) {
  if (!g) return stb_fail("stb_groups_aterms: null group set");
  const int prev_dev = stb_device_enter(g->dev);
  bool fuse = false;
  int v = 0;
  int rc = aterms_prepare(g, D, allow_fuse, &fuse, &v);
  if (!rc) rc = aterms_once(g, x_host, D, out_host, fuse, v, ms_fill, ms_sweep, ms_terms);
  if (rc == 2)  // no waits between workgroups in this form
    rc = aterms_once(g, x_host, D, out_host, false, STB_FILL_PC, ms_fill, ms_sweep, ms_terms) ? 1 : 0;
  stb_device_leave(prev_dev);
  return rc;
}

// The same evaluation in two calls, so that one host thread can keep several GPUs (or several group
// sets) busy: _async queues it on the set's own stream -- behind whatever `stream` holds at this moment,
// when `stream` is not NULL -- and returns; stb_groups_wait blocks until it is through and only then
// writes out_host[0..D-1].  x_host may be reused at once; out_host must stay valid until the wait.
// One evaluation per set at a time.
extern "C" int stb_groups_aterms_async(stb_groups_t *g, const double *x_host, int D, double *out_host, void *stream) {
  STB_ENTRY;
  if (!g) return stb_fail("stb_groups_aterms_async: null group set");
  const int prev_dev = stb_device_enter(g->dev);
  bool fuse = false;
  int v = 0;
  int rc = aterms_prepare(g, D, true, &fuse, &v);
  if (!rc && stream) {
    if (hipEventRecord(g->ev_dep, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(g->st, g->ev_dep, 0) != hipSuccess)
      rc = stb_fail("stb_groups_aterms_async: %s", hipGetErrorString(hipGetLastError()));
  }
  if (!rc) rc = aterms_issue(g, x_host, D, out_host, fuse, v, false);
  stb_device_leave(prev_dev);
  return rc;
}

extern "C" int stb_groups_wait(stb_groups_t *g) {
  STB_ENTRY;
  if (!g) return stb_fail("stb_groups_wait: null group set");
  if (g->pending == 2) {  // (stb_groups_aterms_device in a flow that finished inside the call)
    g->pending = 0;
    return 0;
  }
  const int prev_dev = stb_device_enter(g->dev);
  double *out = g->pend_out;
  const int D = g->pend_D;
  double x[STB_TERMS_DMAX];
  memcpy(x, g->pend_x, sizeof(x));
  double *user = g->pend_user;
  const bool lean = g->pend_lean != 0;
  g->pend_user = nullptr;
  int rc = aterms_finish(g, nullptr, nullptr, nullptr);
  const bool redo = rc == 2;
  if (rc == 2) rc = aterms_once(g, x, D, out, false, STB_FILL_PC, nullptr, nullptr, nullptr) ? 1 : 0;
  // (stb_groups_aterms_device: only the lean flow's last launch writes the caller's device buffer itself)
  if (!rc && user && (redo || !lean)) {
    if (hipMemcpyAsync(user, out, sizeof(double) * D, hipMemcpyHostToDevice, g->st) != hipSuccess || hipStreamSynchronize(g->st) != hipSuccess)
      rc = stb_fail("stb_groups_wait: %s", hipGetErrorString(hipGetLastError()));
  }
  stb_device_leave(prev_dev);
  return rc;
}

// The same evaluation with the D log-posteriors left on the DEVICE, in d_out[0..D): what a caller that hands them to a
// collective wants (the discount axis sharded over GPUs: every rank all-gathers its share) -- no copy to
// the host and back.  Queued on the set's stream behind what `stream` holds now; `stream` in turn waits (on the device)
// for the values, so work queued on it afterwards sees them.  stb_groups_wait(g) must still be called -- before the
// values are trusted: it reports a walk that gave up waiting, and then re-evaluates through stored tables and rewrites
// d_out.
extern "C" int stb_groups_aterms_device(stb_groups_t *g, const double *x_host, int D, double *d_out, void *stream) {
  STB_ENTRY;
  if (!g || !d_out) return stb_fail("stb_groups_aterms_device: null argument");
  const int prev_dev = stb_device_enter(g->dev);
  bool fuse = false;
  int v = 0;
  int rc = aterms_prepare(g, D, true, &fuse, &v);
  if (!rc && stream) {
    if (hipEventRecord(g->ev_dep, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(g->st, g->ev_dep, 0) != hipSuccess)
      rc = stb_fail("stb_groups_aterms_device: %s", hipGetErrorString(hipGetLastError()));
  }
  if (!rc) {
    g->pend_user = d_out;
    rc = aterms_issue(g, x_host, D, g->pend_host, fuse, v, false);
    if (rc) g->pend_user = nullptr;
  }
  if (!rc && !g->pend_lean) {
    // (only the lean flow's last launch writes a device buffer: the other flows are finished here and the totals
    // copied, on the set's stream; the wait that must follow has nothing left to do)
    g->pend_user = nullptr;
    rc = aterms_finish(g, nullptr, nullptr, nullptr);
    if (rc == 2) rc = aterms_once(g, x_host, D, g->pend_host, false, STB_FILL_PC, nullptr, nullptr, nullptr) ? 1 : 0;
    if (!rc && hipMemcpyAsync(d_out, g->pend_host, sizeof(double) * D, hipMemcpyHostToDevice, g->st) != hipSuccess)
      rc = stb_fail("stb_groups_aterms_device: %s", hipGetErrorString(hipGetLastError()));
    if (!rc) g->pending = 2;
  }
  if (!rc && (hipEventRecord(g->ev_done, g->st) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, g->ev_done, 0) != hipSuccess))
    rc = stb_fail("stb_groups_aterms_device: %s", hipGetErrorString(hipGetLastError()));
  stb_device_leave(prev_dev);
  return rc;
}

extern "C" int stb_groups_aterms_timed(stb_groups_t *g, const double *x_host, int D, double *out_host, float *ms_fill,
                                       float *ms_sweep, float *ms_terms) {
  STB_ENTRY;
  return groups_aterms(g, x_host, D, out_host, true, ms_fill, ms_sweep, ms_terms);
}

extern "C" int stb_groups_aterms(stb_groups_t *g, const double *x_host, int D, double *out_host) {
  STB_ENTRY;
  return groups_aterms(g, x_host, D, out_host, true, nullptr, nullptr, nullptr);
}

// the same values through stored tables and the sorted gather whatever D is: no set-up, which is
// what a handful of abscissae evaluated once (ARMS' three starting points) want
// ---- the node from ONE host thread (SURVEY 8e for a C caller; the reference's callers are C: lib/samplea.c:155,
// test/demo.c:478-480).  k group sets -- normally one per device, each made after stb_set_device(dev) from the same pairs
// (stb_groups_create_node does that) -- take contiguous blocks of the D abscissae, sizes differing by at most one, all are
// queued before any is waited for, and out_host[0..D) comes back in the grid's order.  No collective: a C caller's
// "gather" is D doubles arriving in pinned host memory, 8 bytes a discount; RCCL is the transport of the one-process-per-GPU
// layout (libstb_amd/shard.py, bench.py), where the values stay on the devices.
extern "C" int stb_groups_aterms_multi(stb_groups_t *const *sets, int k, const double *x_host, int D, double *out_host) {
  STB_ENTRY;
  if (!sets || k < 1 || !x_host || !out_host || D < 1) return stb_fail("stb_groups_aterms_multi: bad argument (k=%d D=%d)", k, D);
  if (k > D) k = D;  // (a set without a discount sits the call out)
  for (int s = 0; s < k; s++)
    if (!sets[s]) return stb_fail("stb_groups_aterms_multi: set %d is null", s);
  int rc = 0, queued = 0;
  char first_err[512] = "";
  for (int s = 0; s < k && !rc; s++) {
    const int lo = (int)((long long)D * s / k), hi = (int)((long long)D * (s + 1) / k);
    rc = stb_groups_aterms_async(sets[s], x_host + lo, hi - lo, out_host + lo, nullptr);
    if (!rc) queued++;
  }
  if (rc) snprintf(first_err, sizeof(first_err), "%s", stb_last_error());
  for (int s = 0; s < queued; s++)  // (whatever was queued is waited for, error or not)
    if (stb_groups_wait(sets[s]) && !rc) {
      rc = 1;
      snprintf(first_err, sizeof(first_err), "%s", stb_last_error());
    }
  return rc ? stb_fail("stb_groups_aterms_multi: %s", first_err) : 0;
}

// one set per device for the call above: min(ndev, stb_device_count()) sets of the same pairs on devices 0, 1, ..., each able to
// take Dmax discounts (so that any split of a grid of k * Dmax fits); returns how many were made (0: failure, nothing is left)
extern "C" int stb_groups_create_node(int ndev, int I, const int *K, const uint32_t *T, const uint32_t *nflat, const uint16_t *tflat,
                                      const double *bpar, unsigned N, unsigned M, int Dmax, stb_groups_t **sets_out) {
  STB_ENTRY;
  const int have = stb_device_count();
  if (!sets_out || ndev < 1 || have < 1) {
    stb_fail("stb_groups_create_node: %s", have < 1 ? "no HIP device" : "bad argument");
    return 0;
  }
  const int k = ndev < have ? ndev : have;
  for (int s = 0; s < k; s++) {
    const int prev = stb_device_enter(s);
    sets_out[s] = groups_create_here(I, K, T, nflat, tflat, bpar, N, M, Dmax);
    stb_device_leave(prev);
    if (!sets_out[s]) {
      char msg[512];
      snprintf(msg, sizeof(msg), "%s", stb_last_error());
      for (int q = 0; q < s; q++) stb_groups_free(sets_out[q]);
      stb_fail("stb_groups_create_node: device %d: %s", s, msg);
      return 0;
    }
  }
  return k;
}

extern "C" int stb_groups_aterms_tables(stb_groups_t *g, const double *x_host, int D, double *out_host) {
  STB_ENTRY;
  return groups_aterms(g, x_host, D, out_host, false, nullptr, nullptr, nullptr);
}

// new per-restaurant totals and concentrations for the same (n,t) pairs (they change with every
// sweep of a Gibbs sampler while the pairs -- sorted once -- may not)
extern "C" int stb_groups_update_restaurants(stb_groups_t *g, const uint32_t *T, const double *bpar) {
  STB_ENTRY;
  if (!g) return stb_fail("stb_groups_update_restaurants: null group set");
  if (g->pending == 1) return stb_fail("stb_groups_update_restaurants: an evaluation queued with stb_groups_aterms_async has not been waited for");
  const int prev_dev = stb_device_enter(g->dev);
  int rc = 0;
  g->reused = 1;
  if (g->I > 0) {
    // through pinned memory: the caller's arrays are free again on return, and nothing waits for the copy but the
    // evaluation queued behind it (the wait at the START is for an earlier copy out of the same staging area)
    if (hipStreamSynchronize(g->st) != hipSuccess) rc = 1;
    if (!rc) {
      memcpy(g->h_T, T, sizeof(uint32_t) * (size_t)g->I);
      memcpy(g->h_bpar, bpar, sizeof(double) * (size_t)g->I);
    }
    if (rc || hipMemcpyAsync(g->d_T, g->h_T, sizeof(uint32_t) * g->I, hipMemcpyHostToDevice, g->st) != hipSuccess ||
        hipMemcpyAsync(g->d_bpar, g->h_bpar, sizeof(double) * g->I, hipMemcpyHostToDevice, g->st) != hipSuccess)
      rc = stb_fail("stb_groups_update_restaurants: %s", hipGetErrorString(hipGetLastError()));
  }
  stb_device_leave(prev_dev);
  return rc;
}

extern "C" int stb_groups_shape(const stb_groups_t *g, int *I, uint64_t *G, unsigned *N, unsigned *M, int *Dmax) {
  if (!g) return 1;
  if (I) *I = g->I;
  if (G) *G = g->G;
  if (N) *N = g->N;
  if (M) *M = g->M;
  if (Dmax) *Dmax = g->Dmax;
  return 0;
}

// lists.hip -- what a group set needs when its (n,t) pairs are NEW: the pairs' way to the device and the cell lists
// of the fused evaluation, rebuilt on the device without a sort and without a host round trip.
//
// Why.  The reference's callers change the counts between two samplea calls (test/demo.c:405-445 rewrites t[j][i] and
// T[j] in every Gibbs iteration, then :478-480 resamples a): a set of pairs is used for ONE samplea call, i.e. ~8
// posterior evaluations, and whatever is done once per set is paid on every call.  Rounds 2-4 sorted the pairs twice
// per set (rocPRIM: 23 launches, 0.24 ms, by (n,t) for the gather; 21 launches + run-length encoding, 0.33 ms, by
// (tile, group, cell) for the lists of the fused form) with four host round trips in between, and the host copied the
// ragged arrays twice (flat, then to the device through the runtime's own staging): 3.1-3.8 ms a call where an
// unchanged set took 1.57.  Here:
//
//   * stb_groups_pairs_begin / _put / _put_ragged / _commit: the caller's arrays are copied ONCE, into pinned memory (the
//     maxima of n and t -- the table bounds of lib/samplea.c:186-208 -- fall out of the same pass); a whole set at once is
//     copied by a few host threads of the library's own while the calling thread hands the finished pieces to the DMA
//     engine (one core of the GPU box's host moves the 6 MB of 10^6 pairs in 0.41 ms, four in 0.18).
//   * the lists come from a COUNT SLAB: one word per cell of every (tile, group of rows) item in the order the walk
//     looks cells up, i.e. the table's cells once over (33 MB for N = M = 4096, 205 MB for 10^4), zero between builds.
//     k_count_cells adds every pair to its word -- ONE device-scope atomic per pair: this part hands out 24 G of them a
//     second whatever they return and however large the slab (tools/ubench/scatter.hip: 42 us for 10^6), an integer, so
//     the result does not depend on the order of the pairs; k_item_count reads the slab for every item's distinct cells;
//     a prefix sum of those is the CSR row pointer; k_emit_cells has a wave per item compact the item's words -- in position
//     order, the order the sort produced, so the lists are THE SAME BYTES as the sort-based builder's -- and puts every
//     word it read back to zero: the slab is clean for the next set without a memset.  For the grid form the prefix sums and
//     the tiles' entries are one single-workgroup kernel (k_scan_lists), the compaction writes the dense words the walk
//     reads along with the CSR entries (k_emit_both), and the tiles left to helper jobs are chosen by k_jobs_build
//     (groups.hip), which leaves their number in the list for the kernels to read.  No host synchronisation anywhere: the
//     evaluation is queued right behind (N = 4096, halo-block form: 75 us; N = 10^4, grid form: 166 us; the radix sort +
//     run-length encoding they replace: 330-600 us and four host round trips).
//   * pairs are no longer sorted by (n,t) when a set is made: the fused evaluation does not read them, and the gather
//     over a stored table of 4000 x 4000 (64 MB: it lives in the Infinity Cache) is as fast on unsorted pairs
//     (0.035 ms either way, profiles/r05_fresh_before.txt).  The sort is made on first need by the evaluations through
//     stored tables, whose sum then does not depend on the caller's order, as before.
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include "groups.h"

void stb_grid_tile_offsets(const grid_geom &g, std::vector<unsigned> &off);  // grid_hb.hip

// ------------------------------------------------------------------------------------------------
// count slab -> CSR lists

// the word of the slab a pair falls on: 0 nothing (n <= 1 or t = n: log 1), 1 its S_S is log 0 (lib/stable.c:948-949), 2 a cell
__device__ __forceinline__ int slab_word(unsigned nn, unsigned tt, unsigned N, unsigned M, const slab_info &H, unsigned &item, size_t &idx) {
  if (nn <= 1 || nn == tt) return 0;
  if (tt == 0 || nn < tt || tt > M || nn > N) return 1;
  unsigned j = 0, col = 0;  // t = 1: the last element of strip 0's halo, word 0 of the row
  if (tt >= 2) {
    const unsigned e = tt - 2;
    j = e / (unsigned)H.UC;
    col = 1 + (e - j * (unsigned)H.UC);
  }
  const unsigned b = (nn - 2) / (unsigned)H.R, r = (nn - 2) - b * (unsigned)H.R;
  const unsigned b0 = (j * (unsigned)H.UC) / (unsigned)H.R;  // (j < 2^16, UC <= 256)
  const unsigned rec = H.rec_off[j + 1] + (b - b0);  // (b >= b0: the cell lies on or below the diagonal)
  const unsigned q = r / (unsigned)H.G;
  item = rec * (unsigned)H.NQ + q;
  idx = ((size_t)item * (unsigned)H.G + (r - q * (unsigned)H.G)) * H.UCp + col;
  return 2;
}

// (device-scope atomics run at 24 G a second on this part whatever they return and however large the slab
// -- tools/ubench/scatter.hip, profiles/r05_ubench_scatter.txt: 42 us for 10^6 -- so there is ONE per pair: the items'
// distinct cells are counted from the slab afterwards, by reading it)
__global__ __launch_bounds__(256) void k_count_cells(const uint32_t *n, const uint16_t *t, uint64_t G, unsigned N, unsigned M, slab_info H,
                                                     unsigned *slab, unsigned long long *ninf) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int kind = 0;
  unsigned item = 0;
  size_t idx = 0;
  if (g < G) kind = slab_word(n[g], t[g], N, M, H, item, idx);
  if (kind == 2) __hip_atomic_fetch_add(&slab[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long m = __ballot(kind == 1);
  if (m && (threadIdx.x & 63) == 0) atomicAdd(ninf, (unsigned long long)__popcll(m));
}

// a wave per item: how many of its words are not zero
__global__ __launch_bounds__(256) void k_item_count(const unsigned *slab, unsigned nitems, unsigned len, unsigned *icnt) {
  const unsigned item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (item >= nitems) return;
  const unsigned *w = slab + (size_t)item * len;
  unsigned c = 0;
  unsigned k = lane * 4;
  // (four words a lane and load while the item's start is 16-byte aligned -- len is odd for most shapes: word by word then)
  if ((((size_t)item * len) & 3u) == 0) {
    for (; k + 3 < len; k += 256) {
      const uint4 v = *reinterpret_cast<const uint4 *>(w + k);
      c += (v.x != 0u) + (v.y != 0u) + (v.z != 0u) + (v.w != 0u);
    }
    for (unsigned i = (len & ~3u) + lane; i < len; i += 64) c += w[i] != 0u;
  } else {
    for (unsigned i = lane; i < len; i += 64) c += w[i] != 0u;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) icnt[item] = c;
}

// a wave per item: its words, in order, become its list; what was read goes back to zero.  A lane takes four
// consecutive words a step (one 16-byte load where the item starts on a 16-byte boundary: its length is a multiple of 8
// words), the next step's load in flight while this one's words are placed; the order of the entries is the order of the
// words.
__global__ __launch_bounds__(256) void k_emit_cells(unsigned *slab, const unsigned *item_ptr, unsigned nitems, slab_info H, unsigned short *ent_pos,
                                                    unsigned *ent_cnt) {
  const unsigned item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (item >= nitems) return;
  unsigned out = item_ptr[item];
  const unsigned end = item_ptr[item + 1];
  if (out == end) return;  // (nothing of this item occurs: its words are zero, nobody reads them)
  const unsigned len = (unsigned)H.G * H.UCp;
  unsigned *w = slab + (size_t)item * len;
  const unsigned long long below = (1ull << lane) - 1ull;
  if ((((size_t)item * len) & 3u) == 0 && (len & 3u) == 0) {
    auto load4 = [&](unsigned i) -> uint4 { return (i + 3 < len) ? *reinterpret_cast<const uint4 *>(w + i) : uint4{0u, 0u, 0u, 0u}; };
    uint4 nx = load4(lane * 4);
    for (unsigned k = 0; k < len && out < end; k += 256) {
      const unsigned i = k + lane * 4;
      const uint4 c = nx;
      if (k + 256 < len) nx = load4(i + 256);
      const unsigned cc[4] = {c.x, c.y, c.z, c.w};
      // entries before this lane's first word: the non-zero words of the lanes below it
      const unsigned mine = (cc[0] != 0u) + (cc[1] != 0u) + (cc[2] != 0u) + (cc[3] != 0u);
      unsigned before = 0, total = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned long long m = __ballot(cc[j] != 0u);
        before += (unsigned)__popcll(m & below);
        total += (unsigned)__popcll(m);
      }
      if (mine) {
        unsigned o = out + before;
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (cc[j] != 0u) {
            const unsigned ii = i + j, r = ii / H.UCp, col = ii - r * H.UCp;
            ent_pos[o] = (unsigned short)((r << H.PB) | ((unsigned)H.HC - 1u + col));
            ent_cnt[o] = cc[j];
            o++;
            any = true;
          }
        if (any) *reinterpret_cast<uint4 *>(w + i) = uint4{0u, 0u, 0u, 0u};
      }
      out += total;
    }
    return;
  }
  // (an item that does not start on a 16-byte boundary: word by word, two pieces of 64 words in flight)
  unsigned c0 = (lane < len) ? w[lane] : 0u;
  for (unsigned k = 0; k < len && out < end; k += 64) {
    const unsigned i = k + lane;
    const unsigned c = c0;
    if (k + 64 < len) c0 = (i + 64 < len) ? w[i + 64] : 0u;
    const unsigned long long m = __ballot(c != 0u);
    if (c != 0u) {
      const unsigned o = out + (unsigned)__popcll(m & below);
      const unsigned r = i / H.UCp, col = i - r * H.UCp;
      ent_pos[o] = (unsigned short)((r << H.PB) | ((unsigned)H.HC - 1u + col));
      ent_cnt[o] = c;
      w[i] = 0u;
    }
    out += (unsigned)__popcll(m);
  }
}

// ---- the grid form: CSR lists AND the dense words the walk reads, in one pass over the slab ----
#define STB_DENSE_NWMAX 40  // (as in groups.hip)

// One workgroup: the exclusive scan of the items' distinct cells (the CSR row pointer) and, per tile, the words per lane
// its groups get in the dense layout (the most any of them needs; 63: more than the layout takes, the tile is read from
// the CSR lists), their prefix sum and the tile's entry -- what two library scans, k_tile_words and a launch gap did.
__global__ __launch_bounds__(1024) void k_scan_lists(const unsigned *icnt, unsigned nitems, unsigned n_tiles, unsigned NQ, unsigned *item_ptr, unsigned *tnw,
                                                     unsigned *toff, unsigned *tinfo) {
  __shared__ unsigned s_scan[17];
  const unsigned tid = threadIdx.x;
  auto block_exclusive = [&](unsigned mine) -> unsigned { return stb_block_exclusive_1024(mine, s_scan, nullptr); };
  {
    const unsigned per = (nitems + 1 + 1023u) / 1024u, i0 = tid * per, i1 = min(i0 + per, nitems + 1);
    unsigned sum = 0;
    for (unsigned i = i0; i < i1; i++) sum += (i < nitems) ? icnt[i] : 0u;
    unsigned run = block_exclusive(sum);
    for (unsigned i = i0; i < i1; i++) {
      item_ptr[i] = run;
      run += (i < nitems) ? icnt[i] : 0u;
    }
  }
  {
    const unsigned per = (n_tiles + 1023u) / 1024u, t0 = tid * per, t1 = min(t0 + per, n_tiles);
    unsigned sum = 0;
    for (unsigned t = t0; t < t1; t++) {
      unsigned nw = 0;
      for (unsigned q = 0; q < NQ; q++) nw = max(nw, (icnt[(size_t)t * NQ + q] + 63u) / 64u);
      if (nw > STB_DENSE_NWMAX) nw = 63u;
      tnw[t] = nw;
      sum += (nw == 63u) ? 0u : nw * NQ;
    }
    unsigned run = block_exclusive(sum);
    for (unsigned t = t0; t < t1; t++) {
      const unsigned nw = tnw[t];
      toff[t] = run;
      tinfo[t] = (run << 6) | nw;
      run += (nw == 63u) ? 0u : nw * NQ;
    }
  }
}

// a wave per item, as k_emit_cells; every entry also goes to its place among the tile's dense words (position | count <<
// wbits, group q of the tile at words [first + q NW, first + (q + 1) NW) x 64 lanes, filled from the front, zeros behind).
// A count that does not fit a word marks its tile as one read from the CSR lists (NW := 63).
__global__ __launch_bounds__(256) void k_emit_both(unsigned *slab, const unsigned *item_ptr, unsigned nitems, slab_info H, unsigned short *ent_pos, unsigned *ent_cnt,
                                                   unsigned NQ, int wbits, unsigned *tnw, const unsigned *toff, unsigned *tinfo, unsigned *dense) {
  const unsigned item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (item >= nitems) return;
  const unsigned tile = item / NQ, q = item - tile * NQ;
  const unsigned first = item_ptr[item], end = item_ptr[item + 1];
  const unsigned nw = tinfo[tile] & 63u;  // (as k_scan_lists left it: another wave's "63" for a big count comes later or never)
  unsigned *dw = (nw != 0u && nw != 63u) ? dense + ((size_t)toff[tile] + (size_t)q * nw) * 64 : nullptr;
  const unsigned room = nw * 64u;
  if (first != end) {
    unsigned out = first;
    const unsigned len = (unsigned)H.G * H.UCp;
    unsigned *w = slab + (size_t)item * len;
    const unsigned long long below = (1ull << lane) - 1ull;
    const bool wide = (((size_t)item * len) & 3u) == 0 && (len & 3u) == 0;
    auto place = [&](unsigned o, unsigned ii, unsigned c) {
      const unsigned r = ii / H.UCp, col = ii - r * H.UCp;
      const unsigned pos = (r << H.PB) | ((unsigned)H.HC - 1u + col);
      ent_pos[o] = (unsigned short)pos;
      ent_cnt[o] = c;
      if (dw) dw[o - first] = pos | (c << wbits);
      if (c >= (1u << (32 - wbits))) {  // (does not fit a dense word: this tile is read from the CSR lists)
        atomicOr(&tinfo[tile], 63u);
        atomicMax(&tnw[tile], 63u);
      }
    };
    if (wide) {
      auto load4 = [&](unsigned i) -> uint4 { return (i + 3 < len) ? *reinterpret_cast<const uint4 *>(w + i) : uint4{0u, 0u, 0u, 0u}; };
      uint4 nx = load4(lane * 4);
      for (unsigned k = 0; k < len && out < end; k += 256) {
        const unsigned i = k + lane * 4;
        const uint4 c = nx;
        if (k + 256 < len) nx = load4(i + 256);
        const unsigned cc[4] = {c.x, c.y, c.z, c.w};
        unsigned before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned long long m = __ballot(cc[j] != 0u);
          before += (unsigned)__popcll(m & below);
          total += (unsigned)__popcll(m);
        }
        if ((cc[0] | cc[1] | cc[2] | cc[3]) != 0u) {
          unsigned o = out + before;
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (cc[j] != 0u) place(o++, i + j, cc[j]);
          *reinterpret_cast<uint4 *>(w + i) = uint4{0u, 0u, 0u, 0u};
        }
        out += total;
      }
    } else {
      for (unsigned k = 0; k < len && out < end; k += 64) {
        const unsigned i = k + lane;
        const unsigned c = (i < len) ? w[i] : 0u;
        const unsigned long long m = __ballot(c != 0u);
        if (c != 0u) {
          place(out + (unsigned)__popcll(m & below), i, c);
          w[i] = 0u;
        }
        out += (unsigned)__popcll(m);
      }
    }
  }
  // the words behind the group's last cell: zero (the first empty word ends a group's look-ups)
  if (dw)
    for (unsigned k = (end - first) + lane; k < room; k += 64) dw[k] = 0u;
}

int stb_lists_jobs_from(stb_groups_t *g, int which, int D, const grid_geom &gg, const unsigned *h_nw);

static void free_z(void *&p) {
  stb_pool_free(p);
  p = nullptr;
}
#define FREE_Z(x) free_z(reinterpret_cast<void *&>(x))

// New pairs (keep_capacity) or new table bounds: what was built from the old ones goes.  With keep_capacity the buffers
// the slab builder sized for the most a set of G pairs can need stay; those the sort-based builder sized exactly go.
void stb_lists_drop(stb_groups_t *g, bool keep_capacity) {
  for (int w = 0; w < STB_NLISTS; w++) {
    g->lists_ready[w] = 0;
    g->n_jobs[w] = 0;
    if (keep_capacity && g->ent_cap[w]) continue;
    FREE_Z(g->d_item_ptr[w]);
    FREE_Z(g->d_ent_pos[w]);
    FREE_Z(g->d_ent_cnt[w]);
    FREE_Z(g->d_tile_off[w]);
    FREE_Z(g->d_dense[w]);
    FREE_Z(g->d_tinfo[w]);
    FREE_Z(g->d_jobs[w]);
    FREE_Z(g->d_tjob[w]);
    FREE_Z(g->d_tnw[w]);
    FREE_Z(g->d_toff[w]);
    g->ent_cap[w] = 0;
    g->dense_cap[w] = 0;
  }
  // the forms that read the pairs outside the table and the dense count slab of the chain form
  FREE_Z(g->d_n2);
  FREE_Z(g->d_t2);
  FREE_Z(g->d_cnt);
  g->G2 = 0;
  g->n_inf = 0;
  g->fused_ready = 0;
  g->sparse = 0;
  if (!keep_capacity) {
    FREE_Z(g->d_slab);
    FREE_Z(g->d_icnt);
    FREE_Z(g->d_scan_tmp);
    FREE_Z(g->d_dotp);
    g->slab_elems = 0;
    g->slab_which = 0;
    g->scan_tmp_bytes = 0;
    g->dotp_elems = 0;
  }
}

static size_t slab_limit_bytes() { return (size_t)stb_env_int("STB_SLAB_MB", 4096) << 20; }

// Returns 0 with the lists of layout `which` queued on the set's stream (lists_ready set: an evaluation may be queued
// behind them), 2 when this builder does not apply (the caller takes the sort-based one), 1 on error.
int stb_lists_slab_build(stb_groups_t *g, int which, int D, const hb_dot_info &H, const grid_geom &gg) {
  if (which < 2 || !stb_env_int("STB_LISTS_SLAB", 1)) return 2;
  const unsigned N = g->N, M = g->M;
  const uint64_t G = g->G;
  if (G == 0 || G >= 0xffffffffull) return 2;
  // (pairs that could fill a third of the table: the sort-based builder decides by the distinct cells whether the
  // count slab of the chain form serves better)
  if (G * 3 > stb_table_cells(N, M)) return 2;
  slab_info S;
  S.R = H.R;
  S.G = H.G;
  S.NQ = H.NQ;
  S.UC = H.UC;
  S.HC = H.HC;
  S.UCp = (unsigned)H.UC + 1u;
  S.rec_off = H.rec_off;
  const uint64_t nitems64 = (uint64_t)H.n_rec * (unsigned)H.NQ;
  const uint64_t elems64 = nitems64 * (unsigned)H.G * S.UCp;
  if (nitems64 >= (1ull << 31) || elems64 * 4 > slab_limit_bytes()) return 2;
  S.PB = which >= 3 ? stb_pos_bits(H.C) : 8;
  if (H.G > 32 || H.HC - 1 + (int)S.UCp > (1 << S.PB)) return 2;  // (a position is row << PB | element)
  // (the grid form's dense words are sized for the worst case; where that is beyond a 32-bit offset the sort-based builder,
  // which sizes exactly, takes over -- decided here, before anything is queued)
  if (which >= 3 && (size_t)H.NQ * ((size_t)(g->G / 64) + H.n_tiles + 1) >= (1u << 26)) return 2;
  const unsigned nitems = (unsigned)nitems64;
  const size_t elems = (size_t)elems64;
  hipStream_t st = g->st;
  // the slab: laid out for ONE layout at a time (a set is evaluated in one form call after call; another form asks
  // for another slab)
  if (!g->d_slab || g->slab_elems < elems || g->slab_which != which || g->slab_R != H.R || g->slab_G != H.G || g->slab_UCp != (int)S.UCp ||
      g->slab_items != nitems) {
    if (g->d_slab) HIPCHK(hipStreamSynchronize(st));
    if (!g->d_slab || g->slab_elems < elems) {
      FREE_Z(g->d_slab);
      g->slab_elems = 0;
      if (stb_pool_malloc((void **)&g->d_slab, 4 * elems) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory (count slab, %zu MB)", (4 * elems) >> 20);
      g->slab_elems = elems;
      g->slab_clean = 0;
    }
    FREE_Z(g->d_icnt);
    if (stb_pool_malloc((void **)&g->d_icnt, 4 * ((size_t)nitems + 2)) != hipSuccess || (!g->d_ninf && stb_pool_malloc((void **)&g->d_ninf, 64) != hipSuccess))
      return stb_fail("stb_groups_aterms: out of device memory");
    HIPCHK(hipMemsetAsync(g->d_icnt, 0, 4 * ((size_t)nitems + 2), st));
    size_t need = 0;
    if (rocprim::exclusive_scan(nullptr, need, g->d_icnt, g->d_icnt, 0u, (size_t)nitems + 1, rocprim::plus<unsigned>(), st) != hipSuccess)
      return stb_fail("stb_groups_aterms: exclusive_scan (size query) failed");
    if (which >= 3) {  // (... and the scan over the tiles' word counts)
      size_t need2 = 0;
      if (rocprim::exclusive_scan(nullptr, need2, g->d_icnt, g->d_icnt, 0u, (size_t)H.n_tiles + 1, rocprim::plus<unsigned>(), st) != hipSuccess)
        return stb_fail("stb_groups_aterms: exclusive_scan (size query) failed");
      if (need2 > need) need = need2;
    }
    if (need > g->scan_tmp_bytes) {
      FREE_Z(g->d_scan_tmp);
      if (stb_pool_malloc(&g->d_scan_tmp, need ? need : 1) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
      g->scan_tmp_bytes = need ? need : 1;
    }
    g->slab_which = which;
    g->slab_R = H.R;
    g->slab_G = H.G;
    g->slab_UCp = (int)S.UCp;
    g->slab_items = nitems;
  }
  if (!g->slab_clean) HIPCHK(hipMemsetAsync(g->d_slab, 0, 4 * g->slab_elems, st));
  // the list buffers, for the most G pairs can ask for
  if (g->ent_cap[which] < G || !g->d_item_ptr[which]) {
    if (g->d_ent_pos[which]) HIPCHK(hipStreamSynchronize(st));
    FREE_Z(g->d_ent_pos[which]);
    FREE_Z(g->d_ent_cnt[which]);
    FREE_Z(g->d_item_ptr[which]);
    if (stb_pool_malloc((void **)&g->d_ent_pos[which], 2 * (size_t)G + 256) != hipSuccess || stb_pool_malloc((void **)&g->d_ent_cnt[which], 4 * (size_t)G + 256) != hipSuccess ||
        stb_pool_malloc((void **)&g->d_item_ptr[which], 4 * ((size_t)nitems + 2)) != hipSuccess)
      return stb_fail("stb_groups_aterms: out of device memory");
    g->ent_cap[which] = (size_t)G;
  }
  g->slab_clean = 0;
  // (counting the pieces of a hand-over while the next ones still arrive -- on the guess that layout and bounds stay --
  // was built and measured: in the copies' stream the DMA engine waits for every count, the hand-over ends 55 us LATER; on a
  // stream of its own it ends 20 us later still (tools/ab_spec.py in the history of this file): removed)
  HIPCHK(hipMemsetAsync(g->d_ninf, 0, 8, st));
  hipLaunchKernelGGL(k_count_cells, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, g->d_n, g->d_t, G, N, M, S, g->d_slab, g->d_ninf);
  hipLaunchKernelGGL(k_item_count, dim3((nitems + 3) / 4), dim3(256), 0, st, g->d_slab, nitems, (unsigned)H.G * S.UCp, g->d_icnt);
  // (the log-0 pairs of a layout built here are counted on the device, d_ninf, and added by k_eval_tail to g->n_inf, which
  // is what a layout built by the sort counted on the host and stays: the two may count the same pairs twice, which is
  // still "not zero"; stb_lists_drop zeroes it with the pairs)
  if (which < 3) {
    size_t tb = g->scan_tmp_bytes;
    if (rocprim::exclusive_scan(g->d_scan_tmp, tb, g->d_icnt, g->d_item_ptr[which], 0u, (size_t)nitems + 1, rocprim::plus<unsigned>(), st) != hipSuccess)
      return stb_fail("stb_groups_aterms: exclusive_scan failed");
    hipLaunchKernelGGL(k_emit_cells, dim3((nitems + 3) / 4), dim3(256), 0, st, g->d_slab, g->d_item_ptr[which], nitems, S, g->d_ent_pos[which],
                       g->d_ent_cnt[which]);
    HIPCHK(hipGetLastError());
    g->slab_clean = 1;  // (once the stream has come this far; every later use is queued behind it)
  } else {
    // the grid form: CSR lists and the dense words the walk reads in ONE pass over the slab, the scans by one workgroup,
    // then the tiles left to helper jobs -- all on the device, the evaluation queued right behind
    const unsigned n_tiles = H.n_tiles, NQ = (unsigned)H.NQ;
    const size_t words_cap = (size_t)NQ * ((size_t)(G / 64) + n_tiles + 1);  // in units of 64 words: sum over tiles of NQ * ceil(most cells of a group / 64)
    if (words_cap >= (1u << 26)) return stb_fail("stb_groups_aterms: a dense list beyond 2^32 bytes");
    {
      unsigned **bufs[] = {&g->d_tnw[which], &g->d_toff[which], &g->d_tinfo[which]};
      for (unsigned **q : bufs)
        if (!*q && stb_pool_malloc((void **)q, 4 * (size_t)(n_tiles + 2)) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
    }
    if (g->dense_cap[which] < words_cap || !g->d_dense[which]) {
      if (g->d_dense[which]) HIPCHK(hipStreamSynchronize(st));
      FREE_Z(g->d_dense[which]);
      if (stb_pool_malloc((void **)&g->d_dense[which], 256 * words_cap) != hipSuccess) return stb_fail("stb_groups_aterms: out of device memory");
      g->dense_cap[which] = words_cap;
    }
    hipLaunchKernelGGL(k_scan_lists, dim3(1), dim3(1024), 0, st, g->d_icnt, nitems, n_tiles, NQ, g->d_item_ptr[which], g->d_tnw[which], g->d_toff[which],
                       g->d_tinfo[which]);
    hipLaunchKernelGGL(k_emit_both, dim3((nitems + 3) / 4), dim3(256), 0, st, g->d_slab, g->d_item_ptr[which], nitems, S, g->d_ent_pos[which],
                       g->d_ent_cnt[which], NQ, S.PB + 5, g->d_tnw[which], g->d_toff[which], g->d_tinfo[which], g->d_dense[which]);
    HIPCHK(hipGetLastError());
    g->slab_clean = 1;
    if (stb_lists_jobs_device(g, which, D, gg, g->d_tnw[which])) return 1;
  }
  if (!g->d_dotp && stb_groups_alloc_dotp(g)) return 1;
  g->sparse = 1;
  g->lists_ready[which] = 1;
  g->list_R[which] = H.R;
  g->list_G[which] = H.G;
  g->fused_ready = 1;
  g->nsg = (M + 63) / 64 + 4;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// the pairs' way to the device

#define STB_PUT_CHUNK (256u * 1024u)  // pairs per piece handed to the stream (1.5 MB)

extern "C" int stb_groups_pairs_begin(stb_groups_t *g) {
  STB_ENTRY;
  if (!g) return stb_fail("stb_groups_pairs_begin: null group set");
  if (g->pending == 1) return stb_fail("stb_groups_pairs_begin: an evaluation queued with stb_groups_aterms_async has not been waited for");
  const int prev = stb_device_enter(g->dev);
  int rc = 0;
  // (whatever still reads the device copy of the old pairs, or copies out of the staging area, must be through)
  if (hipStreamSynchronize(g->st) != hipSuccess) rc = stb_fail("stb_groups_pairs_begin: %s", hipGetErrorString(hipGetLastError()));
  if (!rc && g->G) {  // (each staging area for itself: one may be there from a call that failed on the other)
    if ((!g->h_pn && stb_pool_malloc((void **)&g->h_pn, 4 * (size_t)g->G, 1) != hipSuccess) ||
        (!g->h_pt && stb_pool_malloc((void **)&g->h_pt, 2 * (size_t)g->G, 1) != hipSuccess))
      rc = stb_fail("stb_groups_pairs_begin: out of pinned host memory");
  }
  g->put_n = g->flushed_n = 0;
  g->put_maxn = g->put_maxt = 0;
  g->putting = rc ? 0 : 1;
  stb_device_leave(prev);
  return rc;
}

// copy and maxima in one pass (the compiler turns both loops into vector code)
static unsigned copy_max_u32(uint32_t *__restrict d, const uint32_t *__restrict s, size_t n, unsigned m) {
  for (size_t i = 0; i < n; i++) {
    const uint32_t v = s[i];
    d[i] = v;
    m = v > m ? v : m;
  }
  return m;
}
static unsigned copy_max_u16(uint16_t *__restrict d, const uint16_t *__restrict s, size_t n, unsigned m) {
  uint16_t mm = 0;
  for (size_t i = 0; i < n; i++) {
    const uint16_t v = s[i];
    d[i] = v;
    mm = v > mm ? v : mm;
  }
  return mm > m ? mm : m;
}

static int put_flush(stb_groups_t *g, bool all) {
  while (g->put_n - g->flushed_n >= (all ? 1u : STB_PUT_CHUNK)) {
    const uint64_t c = all ? g->put_n - g->flushed_n : STB_PUT_CHUNK;
    const uint64_t o = g->flushed_n;
    HIPCHK(hipMemcpyAsync(g->d_n + o, g->h_pn + o, 4 * c, hipMemcpyHostToDevice, g->st));
    HIPCHK(hipMemcpyAsync(g->d_t + o, g->h_pt + o, 2 * c, hipMemcpyHostToDevice, g->st));
    g->flushed_n += c;
  }
  return 0;
}

extern "C" int stb_groups_pairs_put(stb_groups_t *g, const uint32_t *n, const uint16_t *t, uint64_t count, unsigned *maxn, unsigned *maxt) {
  // (no STB_ENTRY: called a thousand times per set, and nothing below can reach the runtime's initialisation -- the
  // set exists, so the runtime is up)
  if (!g || !g->putting) return stb_fail("stb_groups_pairs_put: stb_groups_pairs_begin comes first");
  if (g->put_n + count > g->G) return stb_fail("stb_groups_pairs_put: %llu pairs for a set of %llu", (unsigned long long)(g->put_n + count), (unsigned long long)g->G);
  if (count) {
    g->put_maxn = copy_max_u32(g->h_pn + g->put_n, n, count, g->put_maxn);
    g->put_maxt = copy_max_u16(g->h_pt + g->put_n, t, count, g->put_maxt);
    g->put_n += count;
  }
  if (maxn) *maxn = g->put_maxn;
  if (maxt) *maxt = g->put_maxt;
  if (g->put_n - g->flushed_n < STB_PUT_CHUNK) return 0;
  const int prev = stb_device_enter(g->dev);
  const int rc = put_flush(g, false);
  stb_device_leave(prev);
  return rc;
}

extern "C" int stb_groups_pairs_commit(stb_groups_t *g, const uint32_t *T, const double *bpar, unsigned N, unsigned M) {
  STB_ENTRY;
  if (!g || !g->putting) return stb_fail("stb_groups_pairs_commit: stb_groups_pairs_begin comes first");
  if (g->put_n != g->G) return stb_fail("stb_groups_pairs_commit: %llu of the set's %llu pairs supplied", (unsigned long long)g->put_n, (unsigned long long)g->G);
  const int prev = stb_device_enter(g->dev);
  g->putting = 0;
  int rc = put_flush(g, true);
  if (!rc && (N == 0 || M == 0)) {
    if (!g->have_bounds) rc = stb_fail("stb_groups_pairs_commit: the set has no table bounds yet");
    N = g->N;
    M = g->M;
  }
  if (!rc) rc = stb_groups_set_bounds(g, N, M);  // (drops what depends on the bounds when they change)
  if (!rc) {
    stb_lists_drop(g, true);
    g->sorted = 0;
    g->have_pairs = 1;
    g->reused = 0;
  }
  if (!rc && T && bpar && g->I > 0) {
    memcpy(g->h_T, T, sizeof(uint32_t) * (size_t)g->I);
    memcpy(g->h_bpar, bpar, sizeof(double) * (size_t)g->I);
    if (hipMemcpyAsync(g->d_T, g->h_T, sizeof(uint32_t) * (size_t)g->I, hipMemcpyHostToDevice, g->st) != hipSuccess ||
        hipMemcpyAsync(g->d_bpar, g->h_bpar, sizeof(double) * (size_t)g->I, hipMemcpyHostToDevice, g->st) != hipSuccess)
      rc = stb_fail("stb_groups_pairs_commit: %s", hipGetErrorString(hipGetLastError()));
  }
  stb_device_leave(prev);
  return rc;
}

// ---- a whole set at once, copied by several host threads while the calling thread hands the finished pieces to the
// stream.  One core copies 6 MB into pinned memory in 0.41 ms (14.6 GB/s: what a core of the GPU box's host moves),
// which was the largest single piece of a samplea call on new pairs; four copy 1.5 MB each while the DMA engine takes
// what is done.  The workers are the library's own (started on first use, STB_PUT_THREADS of them, default 4; 0: the
// calling thread copies); they touch host memory only.
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

namespace {
struct put_slice {
  // restaurants [i0, i1) of a ragged set, or one flat range when K is null; the slice's pairs start at `off`
  int i0, i1;
  uint64_t off, count;
  std::atomic<uint64_t> done;  // pairs of the slice copied so far
  unsigned maxn, maxt;
};
struct put_job {
  const int *K;
  uint32_t *const *n;
  uint16_t *const *t;
  const uint32_t *nflat;
  const uint16_t *tflat;
  uint32_t *dn;
  uint16_t *dt;
  put_slice *slices;
  int nslices;
  std::atomic<int> next;  // the next slice nobody has taken
};
struct put_pool {
  std::mutex use;  // one hand-over at a time has the workers; another thread's meanwhile is copied by that thread itself
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::thread> th;
  put_job *job = nullptr;
  unsigned long long gen = 0;
  int busy = 0;
};
put_pool *g_put_pool = nullptr;  // (never destroyed: its threads sleep until the process ends)
std::once_flag g_put_once;

// one slice, if any is left: copied by whoever calls (a worker of the pool, or the calling thread while it waits for them)
bool put_step(put_job *J) {
  {
    const int k = J->next.fetch_add(1);
    if (k >= J->nslices) return false;
    put_slice &S = J->slices[k];
    unsigned mn = 0, mt = 0;
    uint64_t o = S.off, since = 0;
    if (J->K) {
      for (int i = S.i0; i < S.i1; i++) {
        const uint64_t c = (uint64_t)(J->K[i] > 0 ? J->K[i] : 0);
        mn = copy_max_u32(J->dn + o, J->n[i], c, mn);
        mt = copy_max_u16(J->dt + o, J->t[i], c, mt);
        o += c;
        since += c;
        if (since >= 32768) {  // (publish: the calling thread hands finished pieces to the stream)
          S.done.store(o - S.off, std::memory_order_release);
          since = 0;
        }
      }
    } else {
      for (uint64_t c0 = 0; c0 < S.count; c0 += 32768) {
        const uint64_t c = S.count - c0 < 32768 ? S.count - c0 : 32768;
        mn = copy_max_u32(J->dn + o, J->nflat + o, c, mn);
        mt = copy_max_u16(J->dt + o, J->tflat + o, c, mt);
        o += c;
        S.done.store(o - S.off, std::memory_order_release);
      }
    }
    S.maxn = mn;
    S.maxt = mt;
    S.done.store(S.count, std::memory_order_release);
  }
  return true;
}
void put_run(put_job *J) {
  while (put_step(J)) {
  }
}

// (a fork()ed child has the pool's pointer but none of its threads: it copies on its own thread)
void put_atfork_child() { g_put_pool = nullptr; }

void put_worker(put_pool *P) {
  unsigned long long seen = 0;
  for (;;) {
    put_job *J;
    {
      std::unique_lock<std::mutex> lk(P->mu);
      P->cv.wait(lk, [&] { return P->gen != seen; });
      seen = P->gen;
      J = P->job;
      if (!J) continue;
      P->busy++;
    }
    put_run(J);
    {
      std::lock_guard<std::mutex> lk(P->mu);
      P->busy--;
    }
    P->cv.notify_all();
  }
}
}  // namespace

// all G pairs of the set, ragged (K, n[i], t[i]) or flat (K null: nflat, tflat), between begin and commit
static int put_all(stb_groups_t *g, int I, const int *K, uint32_t *const *n, uint16_t *const *t, const uint32_t *nflat, const uint16_t *tflat,
                   unsigned *maxn, unsigned *maxt) {
  if (!g || !g->putting || g->put_n != 0) return stb_fail("stb_groups_pairs_put_ragged: between stb_groups_pairs_begin and the first put only");
  uint64_t G = 0;
  if (K) {
    for (int i = 0; i < I; i++) G += (uint64_t)(K[i] > 0 ? K[i] : 0);
    if (G != g->G) return stb_fail("stb_groups_pairs_put_ragged: %llu pairs for a set of %llu", (unsigned long long)G, (unsigned long long)g->G);
  } else {
    G = g->G;
  }
  int W = stb_env_int("STB_PUT_THREADS", 4);
  // (the cores this process may run on, not those of the machine: under taskset to one core the calling thread copies alone)
  unsigned hw = std::thread::hardware_concurrency();
  {
    cpu_set_t cs;
    CPU_ZERO(&cs);
    if (sched_getaffinity(0, sizeof(cs), &cs) == 0 && CPU_COUNT(&cs) > 0) hw = (unsigned)CPU_COUNT(&cs);
  }
  if (hw && W > (int)hw - 1) W = (int)hw - 1;
  if (W > 16) W = 16;
  if (G < 131072) W = 0;  // (small sets: the calling thread is through before a worker is awake)
  // slices of about 64 K pairs, taken in order by whoever is free
  const uint64_t per = 65536;
  const size_t cap = (size_t)(G / per + 2);  // (every slice but the last holds at least `per` pairs)
  std::unique_ptr<put_slice[]> slices(new put_slice[cap]);
  int ns = 0;
  auto add = [&](int i0, int i1, uint64_t off, uint64_t cnt) {
    put_slice &S = slices[(size_t)ns++];
    S.i0 = i0;
    S.i1 = i1;
    S.off = off;
    S.count = cnt;
    S.done.store(0);
    S.maxn = S.maxt = 0;
  };
  if (K) {
    int i0 = 0;
    uint64_t off = 0, cnt = 0;
    for (int i = 0; i < I; i++) {
      cnt += (uint64_t)(K[i] > 0 ? K[i] : 0);
      if (cnt >= per || i == I - 1) {
        add(i0, i + 1, off, cnt);
        off += cnt;
        cnt = 0;
        i0 = i + 1;
      }
    }
  } else {
    for (uint64_t off = 0; off < G; off += per) add(0, 0, off, G - off < per ? G - off : per);
  }
  put_job J;
  J.K = K;
  J.n = n;
  J.t = t;
  J.nflat = nflat;
  J.tflat = tflat;
  J.dn = g->h_pn;
  J.dt = g->h_pt;
  J.slices = slices.get();
  J.nslices = ns;
  J.next.store(0);
  put_pool *P = nullptr;
  std::unique_lock<std::mutex> use_lk;
  if (W > 0 && J.nslices > 1) {
    std::call_once(g_put_once, [W] {
      put_pool *np = new put_pool;
      try {
        for (int w = 0; w < W; w++) np->th.emplace_back(put_worker, np);
      } catch (...) {  // (no more threads to be had: those that started serve; none: the callers copy themselves)
      }
      for (auto &th : np->th) th.detach();
      g_put_pool = np;
      (void)pthread_atfork(nullptr, nullptr, put_atfork_child);
    });
    P = (g_put_pool && !g_put_pool->th.empty()) ? g_put_pool : nullptr;
    if (P) {
      // (samplers of different host threads share the workers: whoever finds them busy copies on its own thread)
      use_lk = std::unique_lock<std::mutex>(P->use, std::try_to_lock);
      if (!use_lk.owns_lock()) P = nullptr;
    }
  }
  if (P) {
    {
      std::lock_guard<std::mutex> lk(P->mu);
      P->job = &J;
      P->gen++;
    }
    P->cv.notify_all();
  }
  int rc = 0;
  const int prev = stb_device_enter(g->dev);
  if (!P) {
    put_run(&J);  // (the calling thread copies; the pieces go out below)
  }
  // hand finished pieces to the stream, in order, while the workers copy
  {
    uint64_t flushed = 0;
    int k = 0;
    while (k < J.nslices && !rc) {
      put_slice &S = slices[(size_t)k];
      const uint64_t d = S.done.load(std::memory_order_acquire);
      const uint64_t upto = S.off + d;
      if (d == S.count) k++;
      if (upto - flushed >= 131072 || (k == J.nslices && upto > flushed)) {
        if (hipMemcpyAsync(g->d_n + flushed, g->h_pn + flushed, 4 * (upto - flushed), hipMemcpyHostToDevice, g->st) != hipSuccess ||
            hipMemcpyAsync(g->d_t + flushed, g->h_pt + flushed, 2 * (upto - flushed), hipMemcpyHostToDevice, g->st) != hipSuccess)
          rc = stb_fail("stb_groups_pairs_put_ragged: %s", hipGetErrorString(hipGetLastError()));
        flushed = upto;
      } else if (d != S.count) {
        // (nothing to hand over yet: take a slice too -- the workers may be busy, few, or, in a fork()ed child, gone)
        if (!put_step(&J)) __builtin_ia32_pause();
      }
    }
  }
  stb_device_leave(prev);
  if (P) {  // (the job lives on this stack: every worker must have let go of it)
    std::unique_lock<std::mutex> lk(P->mu);
    P->job = nullptr;
    P->cv.wait(lk, [&] { return P->busy == 0; });
  }
  if (rc) return rc;
  unsigned mn = 0, mt = 0;
  for (int k = 0; k < ns; k++) {
    mn = slices[(size_t)k].maxn > mn ? slices[(size_t)k].maxn : mn;
    mt = slices[(size_t)k].maxt > mt ? slices[(size_t)k].maxt : mt;
  }
  g->put_n = g->flushed_n = G;
  g->put_maxn = mn;
  g->put_maxt = mt;
  if (maxn) *maxn = mn;
  if (maxt) *maxt = mt;
  return 0;
}

extern "C" int stb_groups_pairs_put_ragged(stb_groups_t *g, int I, const int *K, uint32_t *const *n, uint16_t *const *t, unsigned *maxn, unsigned *maxt) {
  if (!K || !n || !t) return stb_fail("stb_groups_pairs_put_ragged: null argument");
  return put_all(g, I, K, n, t, nullptr, nullptr, maxn, maxt);
}

// new pairs for the same shape and the same table bounds, from flat arrays
extern "C" int stb_groups_update_pairs(stb_groups_t *g, const uint32_t *nflat, const uint16_t *tflat) {
  if (!g) return stb_fail("stb_groups_update_pairs: null group set");
  if (!g->have_bounds) return stb_fail("stb_groups_update_pairs: the set has no table bounds yet (stb_groups_pairs_commit sets them)");
  if (stb_groups_pairs_begin(g)) return 1;
  if (g->G && put_all(g, 0, nullptr, nullptr, nullptr, nflat, tflat, nullptr, nullptr)) return 1;
  return stb_groups_pairs_commit(g, nullptr, nullptr, 0, 0);
}

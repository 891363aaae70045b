// groups.h -- the device-resident group set (stb_groups_t) as the translation units that work on it see it:
// groups.hip (creation, evaluation), lists.hip (the cell lists of the fused evaluation built from a count slab,
// the pairs' way to the device).
#ifndef STB_GROUPS_H
#define STB_GROUPS_H

#include "stb_common.h"

#define STB_TERMS_DMAX 64  // (as in sweep_terms.hip: abscissae per stb_restaurant_terms call)
#define STB_NLISTS 6
#define GH_JQ_HOST 64  // (grid_hb.hip: GH_JQ, the strips from the left whose tiles can be helper jobs)
#define STB_WS_FORM 4096  // lean flow: the discounts at the start of d_ws_fill, the form's workspace from here

// the count slab's geometry (lists.hip): what k_count_cells needs to find a pair's word
struct slab_info {
  int R, G, NQ, UC, HC;  // rows of a block / of a group, groups per item base, own columns of a strip, halo columns
  int PB;                // a position is row in group << PB | element of the wave
  unsigned UCp;          // words of a row of an item: column 1 (strip 0 only) + the strip's own columns
  const unsigned *rec_off;
};

// ------------------------------------------------------------------------------------------------
// device-resident group set

struct stb_groups {
  int dev;  // the device everything below lives on
  int I;
  uint64_t G;
  unsigned N, M;
  int Dmax;
  uint32_t *d_n, *d_T;
  uint16_t *d_t;
  double *d_bpar;
  double *d_tables, *d_S1, *d_out;  // d_out: [2][Dmax]
  uint64_t tstride;
  void *d_ws_fill, *d_ws_sweep, *d_ws_terms;
  size_t ws_fill, ws_sweep, ws_terms;
  hipStream_t st;
  hipEvent_t ev[4];
  // fused evaluation (stb_groups_aterms with the chain form): occurrence count per table cell, the
  // pairs that do not address a table cell (t = 1, t = n, out of bounds), partial sums of the fill
  unsigned *d_cnt;
  uint32_t *d_n2;
  uint16_t *d_t2;
  uint64_t G2;
  double *d_dotp;
  size_t dotp_elems;
  int fused, fused_ready;
  // sparse form of the fused evaluation: CSR of the occurring cells per item, in several layouts, each built when
  // first needed: [0] (trip, 64-column slice from column 1) for k_fill_chain, [1] the same from column 2 for
  // k_fill_ck, [2] (tile, group of 4 rows) for k_fill_hb's tile workers, [3] / [4] (strip, block, group of G rows)
  // for k_grid_hb's self-summing spine with 2 / 4 columns per lane, [5] with 8 (column 1 -- the pairs with t = 1 -- included)
  unsigned *d_item_ptr[STB_NLISTS];
  unsigned short *d_ent_pos[STB_NLISTS];
  unsigned *d_ent_cnt[STB_NLISTS];
  unsigned nsg;
  int lists_ready[STB_NLISTS];
  int list_R[STB_NLISTS], list_G[STB_NLISTS];  // [3], [4]: the block and group length the list was built for
  unsigned *d_tile_off[STB_NLISTS];            // [3], [4]: first tile of every strip (grid_hb.hip)
  unsigned *d_dense[STB_NLISTS];               // [3], [4]: the listed cells as words per lane, group after group: what the walk reads
  unsigned *d_tinfo[STB_NLISTS];               // [3], [4]: per tile, where its words start and how many a group has
  unsigned *d_jobs[STB_NLISTS];                // [3], [4]: the tiles whose cells are left to helper waves, in the order they become ready
  unsigned *d_tjob[STB_NLISTS];                // [3], [4]: per tile its place in d_jobs, or 0xffffffff
  unsigned n_jobs[STB_NLISTS];
  int hb_sum_C;                                // columns per lane of the halo-block summing form's strips for this set (stb_hb_sum_C at its creation)
  uint64_t n_inf;                              // pairs whose S_S is log 0 (t = 0, t > n, outside the bounds)
  int sparse;
  int reused;  // stb_groups_update_restaurants has been called: the same pairs serve call after call
  // an evaluation that has been queued and not yet waited for (stb_groups_aterms_async / stb_groups_wait)
  double *h_out;  // pinned, [2][Dmax] + 2: what the stream copies the sums to (lean flow: totals, then the fill's error words)
  double *h_out_dev;           // the device's address of it
  size_t ws_zero;              // bytes from the start of d_ws_fill + STB_WS_FORM known to be zero (lean flow)
  int pend_lean;               // the queued evaluation took the lean flow
  double seq, pend_seq;        // one-discount lean evaluations: the word k_eval_tail writes last to pinned memory (0: the host waits for the event)
  double *pend_user;           // stb_groups_aterms_device: where the totals go on the device (or null)
  double pend_host[STB_TERMS_DMAX];  // ... and where stb_groups_wait puts them on the host meanwhile
  hipEvent_t ev_done;
  hipEvent_t ev_dep;
  int pending, pend_D, pend_fuse, pend_v;
  int sel_which;  // the list layout aterms_prepare chose for a fused evaluation in the halo-block form
  double *pend_out;
  double pend_x[STB_TERMS_DMAX];
  last_fill pend_fill;
  // ---- the pairs' way to the device (lists.hip).  A caller whose counts change between calls -- every Gibbs sampler:
  // the reference's own loop rewrites t[j][i] / T[j] each iteration, test/demo.c:405-445 -- hands the new pairs over
  // piece by piece (stb_groups_pairs_put); each piece is copied into pinned memory and goes to the device while the
  // caller is still copying the next.
  uint32_t *h_pn;              // pinned [G]
  uint16_t *h_pt;              // pinned [G]
  uint32_t *h_T;               // pinned [I]
  double *h_bpar;              // pinned [I]
  uint64_t put_n, flushed_n;   // pairs staged so far / handed to the stream
  unsigned put_maxn, put_maxt; // the largest n and t among them
  int putting;                 // between stb_groups_pairs_begin and _commit
  int have_pairs;              // the set holds pairs (a set may be created empty)
  int have_bounds;             // N, M are set and what depends on them is allocated
  int sorted;                  // d_n / d_t are in (n, t) order -- what the gather over stored tables wants; made so on first need
  // ---- cell lists from a count slab (lists.hip): a word per cell of the (tile, group) items, zero between builds
  unsigned *d_slab;
  size_t slab_elems;
  int slab_which;              // the layout the slab is laid out for (2, 3, 4), with list_R / list_G / Dmax-dependent geometry below
  int slab_R, slab_G, slab_UCp;
  unsigned slab_items;
  int slab_clean;              // every word of it is zero
  unsigned *d_icnt;            // [items + 1] distinct cells per item (zero between builds)
  unsigned long long *d_ninf;  // pairs whose S_S is log 0, counted on the device
  void *d_scan_tmp;
  size_t scan_tmp_bytes;
  size_t ent_cap[STB_NLISTS];  // entries the list buffers of a layout hold (0: sized exactly by the sort-based builder)
  size_t dense_cap[STB_NLISTS];
  unsigned *d_tnw[STB_NLISTS], *d_toff[STB_NLISTS];  // [3] .. [5]: words per group and first word of every tile (the dense layout), kept
};

// the list layout of the grid form with C columns per lane, and back; a position is row << stb_pos_bits(C) | element of the wave
static inline int stb_grid_which(int C) { return C == 2 ? 3 : (C == 4 ? 4 : 5); }
static inline int stb_which_C(int which) { return which == 3 ? 2 : (which == 4 ? 4 : 8); }
static inline int stb_pos_bits(int C) { return C == 8 ? 9 : 8; }

#if defined(__HIPCC__)
// exclusive prefix sum of one value per thread over a workgroup of 1024 threads (16 waves): shuffles inside a wave, the
// waves' totals through LDS (`lds`: 17 words); *total receives the sum over the workgroup.  Two barriers.
__device__ __forceinline__ unsigned stb_block_exclusive_1024(unsigned mine, unsigned *lds, unsigned *total) {
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned v = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned u = __shfl_up(v, o, 64);
    if ((int)lane >= o) v += u;
  }
  if (lane == 63) lds[wave] = v;
  __syncthreads();
  if (tid == 0) {
    unsigned run = 0;
    for (int w = 0; w < 16; w++) {
      const unsigned t = lds[w];
      lds[w] = run;
      run += t;
    }
    lds[16] = run;
  }
  __syncthreads();
  const unsigned ex = lds[wave] + v - mine;
  if (total) *total = lds[16];
  __syncthreads();
  return ex;
}
#endif

// lists.hip
int stb_lists_slab_build(stb_groups_t *g, int which, int D, const hb_dot_info &H, const grid_geom &gg);  // 0 built (queued), 1 error, 2 not applicable
void stb_lists_drop(stb_groups_t *g, bool keep_capacity);   // new pairs or new bounds: what was built from the old ones goes
int stb_groups_set_bounds(stb_groups_t *g, unsigned N, unsigned M);
int stb_groups_sort_pairs(stb_groups_t *g);                  // groups.hip
int stb_groups_alloc_dotp(stb_groups_t *g);
int stb_lists_jobs_device(stb_groups_t *g, int which, int D, const grid_geom &gg, const unsigned *tnw);


#endif

// grid_hb.hip -- the fused grid evaluation of aterms' table part: for every discount a_d of a grid,
//     sum over the pairs (n, t), n > 1, of S_S(n, t; a_d)            (reference lib/samplea.c:68-80)
// without ever storing a table.  The table recurrence of S_remake_part (reference lib/stable.c:321-388),
//     S^n_m = (n-1-m a) S^{n-1}_m + S^{n-1}_{m-1},
// is walked as in the halo-block fill (fill_hb.hip: a wave owns a strip of columns of one table for all rows,
// block-floating cells, blocks of R rows behind a halo of R columns, hand-overs through LDS inside a workgroup and
// through records in HBM between workgroups), but here the walking wave itself sums its strip's listed cells:
// no tile workers, a second walk only of the few tiles that go to helper jobs, no S1 vector (column 1 is the last
// element of strip 0's halo).
//
// What it is built around (MI355X, tools/ubench/rowpace.hip and the timelines in profiles/):
//   * a lone wave walks a row of 2 columns per lane in 32 cycles, of 4 in 50, and one walking wave keeps a SIMD's
//     vector issue busy: a second one on the same SIMD shares it, and every strip of a table moves at the pace
//     of the slowest.  While the batch has no more strips than the chip has SIMDs every strip gets its own
//     (a workgroup per compute unit, 4 strips each); beyond that the launch is cut into PHASES of blocks so that
//     the strips the diagonal has not reached yet do not hold places (strip state travels between phases through
//     HBM: 64 C doubles a strip);
//   * a row of 64 C doubles staged in LDS costs 13 cycles of a compute unit's LDS store path per kilobyte, which
//     four walking waves saturate: only every other row is staged, and a listed cell of a row in between is
//     taken from the staged row above it by one step of the recurrence, in the look-up;
//   * a look-up pass costs the same for 1 or 64 cells: groups of G rows are sized so that a pass is mostly full;
//   * the log: exponent field + lane exponent go to an exact integer sum, the mantissa part to a double;
//   * every strip of a table moves at the pace of the strips to its left, which hold most of the pairs: a tile that
//     needs many look-up passes is only walked by its strip, which leaves its wave's state as a record, and summed --
//     as a job -- by a wave whose own strip has ended (see run_job).
#include <mutex>
#include <vector>

#include "fill_chain.h"

#define GH_PMAX 7
#define GH_NWMAX 8
#define GH_MAXR 48            // rows of a block at most
#define GH_SL 4               // ring of hand-overs between two spine waves of a workgroup (blocks)
#define GH_FSL 4              // ring of hand-overs from the fetcher to spine wave 0
#define GH_EOFF32 (1u << 30)
#define GH_WRITTEN 0x8000000000000000ull
#define GH_SPIN 48
#define GH_MAXPH 16           // phases at most
#define GH_JOB_MB 192          // room for the records of helper jobs
#define GH_JOBS 1500           // jobs of a table at most
#define GH_JQ 64               // strips (from the left) whose tiles can be jobs: a queue each
// the job list in device memory: words [0, GH_JQ] the first job of every strip's queue (the last: the number of jobs),
// word GH_JNUM the number of jobs -- read by the kernels, so that a list built ON the device needs no host round trip --,
// jobs from word GH_JBASE on
#define GH_JNUM (GH_JQ + 1)
#define GH_JBASE 128
// a listed cell's position: row in group << GH_PB(C) | element of the wave (64 C of them, halo included); a dense word is
// position | count << (GH_PB(C) + 5)
#define GH_PB(C) ((C) == 8 ? 9 : 8)
#ifndef GH_NOSTORE
#define GH_NOSTORE 0          // diagnostic build: 1 nothing is staged (results wrong)
#endif

struct gh_args {
  const double *a;             // [D] discounts
  const double2 *lt;           // log table
  unsigned *hdr;               // [1] error code, [2] error detail (shared by the phases)
  unsigned *ticket;            // this phase's role ticket
  unsigned long long *ck_v;    // [D][B][NB][HL * C] records: the rightmost HL lanes of a workgroup's last strip before a block
  unsigned *ck_e;              // [D][B][NB][HL]     ... and their lane exponents + GH_EOFF32
  double *state_v;             // [D][JW][64 * C]  a strip's significands between two phases
  int *state_e;                // [D][JW][64]      ... and lane exponents
  const unsigned *tile_off;    // [JW + 2], entry j + 1: tiles of the strips before strip j (a strip's tiles are its blocks from its first one on)
  const unsigned *tinfo;       // [n_tiles] first word / 64 << 6 | NW: where a tile's listed cells start in `dense` and how many
                               // words per lane each of its NQ groups of G rows has (63: taken from the three lists below)
  const unsigned *dense;       // a listed cell is a word: (row in group << 8 | element of the wave, halo included) | count << 13,
                               // 0 for none; group q of a tile: words [first + q NW, first + (q + 1) NW) x 64 lanes
  const unsigned *item_ptr;    // [n_tiles * NQ + 1] first list entry of every (tile, group)
  const unsigned short *ent_pos;  // row in group << 8 | element of the wave
  const unsigned *ent_cnt;     // occurrence count
  double *dotp;                // [D][JW + n_jobs][2]: sum of count x binary exponent (an integer), sum of count x log of the mantissa part; a strip's, then a job's
  const unsigned *jobs;        // the job list (GH_JQ + 1 queue starts, the number of jobs at GH_JNUM, from GH_JBASE on the tiles left
                               // to helper waves: strip | block << 16, strip after strip, a strip's by block), or null
  const unsigned *tjob;        // [n_tiles] a tile's place in `jobs`, or 0xffffffff
  unsigned job_cap;            // jobs a table has at most (what the records below are sized for)
  unsigned *jticket;           // [GH_JQ] x 16 words: per strip the next (job, table) to hand out, on a line of its own
  unsigned *jflag;             // [D][n_jobs] non-zero: the record is written
  int *jrec_e;                 // [D][n_jobs][64]      a job's record: the lane exponents of the tile's own wave before the block
  double *jrec_v;              // [D][n_jobs][64 * C]  ... and its significands
  unsigned N, M;
  int D, B, JW, NB;            // tables, workgroups per table (all phases), strips per table, blocks
  int P, R, HL, U;             // spine waves per workgroup, rows per block, halo lanes, own lanes
  int JWa, b_begin, b_end;     // this phase: strips whose first block lies before b_end, blocks [b_begin, b_end)
  unsigned long long timeout;  // wall_clock64 ticks a wait may last
  int poll_nap;
  int diag;                    // STB_GRID_DIAG: 4 no look-ups, 8 nothing staged either, 32 nobody takes the jobs (results wrong)
  unsigned long long *dbg;     // STB_HB_TIMELINE: table 0, [JW][NB + 2] (start, block starts, end)
};

typedef double gh_double2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void gh_store_wt16(unsigned long long *p, unsigned long long a, unsigned long long b) {
  typedef unsigned long long gh_u64x2 __attribute__((ext_vector_type(2)));
  const gh_u64x2 v2 = {a, b};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v2) : "memory");
}

template <int C>
__device__ __forceinline__ void gh_renorm(double (&v)[C], int &ep) {
  double vmax = v[0];
#pragma unroll
  for (int i = 1; i < C; i++) vmax = fmax(vmax, v[i]);
  const int sh = (vmax != 0.0) ? __builtin_amdgcn_frexp_exp(vmax) + PC_BIAS : 0;
#pragma unroll
  for (int i = 0; i < C; i++) v[i] = ldexp(v[i], -sh);
  ep += sh;
}

template <int C>
__device__ __forceinline__ void gh_row(double (&v)[C], double (&coef)[C], double s) {
  const double t0 = wave_shr1_zero(v[C - 1]) * s;
#pragma unroll
  for (int i = C - 1; i >= 1; i--) v[i] = fma(coef[i], v[i], v[i - 1]);
  v[0] = fma(coef[0], v[0], t0);
#pragma unroll
  for (int i = 0; i < C; i++) coef[i] += 1.0;
}

__host__ __device__ static inline int gh_first_block(int j, int UC, int R) { return (int)(((long long)j * UC) / R); }

// One look-up pass: every lane with `valid` takes one listed cell of the group just walked.  Every K-th row of the
// group is staged (rows 0, K, 2K, ..: row r at stage + (r / K) WS); a cell of a row in between is one to K - 1 steps
// of the recurrence away from the staged row above it -- the very operations the walking wave itself performs for that
// cell, on the K cells of the staged row it depends on (they lie in the cell's own lane and, at most, the lane to its
// left: K <= C).
template <int C, int K>
__device__ __forceinline__ void gh_lookup(bool valid, unsigned pos, unsigned cnt, const double *stage, const int *se,
                                          const double2 *lt, double a, int mE0, int nrow0, int one_hi, long long &accK, double &accF) {
  static_assert(K == 2 || (K == 4 && C >= 4), "rows between two staged ones");
  constexpr int WS = 64 * C;
  if (valid) {
    const int cw = (int)(pos & ((1u << GH_PB(C)) - 1u)), r = (int)(pos >> GH_PB(C)) & 31;
    const int j = r & (K - 1);  // steps below the staged row
    const double *row = stage + (r / K) * WS;
    const int ln = cw / C;
    const int e = se[ln];
    // (what lives in the lane to the left is under that lane's exponent: the factor the walk uses)
    const int dl = (ln > 0 ? se[ln - 1] : e) - e;
    const double sfac = ldexp(1.0, min(max(dl, -1100), 220));
    const int ci = cw & (C - 1);  // the cell's place in its lane
    // coefficient n' - 1 - m' a of row n0 + 1 (n0: the staged row) at the cell's column m = mE0 + cw; a column to the
    // left adds a, a row down adds 1
    const double c10 = fma(-(double)(mE0 + cw), a, (double)(nrow0 + r - j));
    double x;
    if constexpr (K == 2) {
      const double v0 = row[cw];
      const double v1 = row[cw - 1 < 0 ? 0 : cw - 1] * (ci < 1 ? sfac : 1.0);
      x = j ? fma(c10, v0, v1) : v0;
    } else {
      const double v0 = row[cw];
      const double v1 = row[cw - 1 < 0 ? 0 : cw - 1] * (ci < 1 ? sfac : 1.0);
      const double v2 = row[cw - 2 < 0 ? 0 : cw - 2] * (ci < 2 ? sfac : 1.0);
      const double v3 = row[cw - 3 < 0 ? 0 : cw - 3] * (ci < 3 ? sfac : 1.0);
      const double a2 = a + a;
      const double w0 = fma(c10, v0, v1), w1 = fma(c10 + a, v1, v2), w2 = fma(c10 + a2, v2, v3);  // row n0 + 1: columns m, m - 1, m - 2
      const double c20 = c10 + 1.0;
      const double u0 = fma(c20, w0, w1), u1 = fma(c20 + a, w1, w2);                               // row n0 + 2: columns m, m - 1
      const double t0 = fma(c20 + 1.0, u0, u1);                                                    // row n0 + 3
      x = (j == 0) ? v0 : ((j == 1) ? w0 : ((j == 2) ? u0 : t0));
    }
    const int hi = __double2hiint(x);
    const double2 t = lt[(hi >> 13) & 127];
    const double z = __hiloint2double(mantissa_of_one(hi, one_hi), __double2loint(x));
    const double rr = fma(z, t.x, -1.0);
    double pl = fma(rr, 0.2, -0.25);
    pl = fma(rr, pl, 1.0 / 3.0);
    pl = fma(rr, pl, -0.5);
    pl = fma(rr, pl, 1.0);
    const int kx = (int)((hi >> 20) & 0x7ff) - 1023 + e;
    accK += (long long)(int)cnt * (long long)kx;
    accF = fma((double)cnt, fma(rr, pl, t.y), accF);
  }
}

// K: every K-th row of a group is staged (2, or 4 with 4 columns per lane)
#ifndef GH_C8_WAVES
#define GH_C8_WAVES 2  // waves per SIMD the 8-column kernels are compiled for (191 registers; with 3 the compiler spills into the block loop: bare walk 1.32 against 1.04 ms)
#endif
template <int C, int G, int K>
__global__ __launch_bounds__(64 * GH_NWMAX, (C == 8 ? GH_C8_WAVES : 2)) void k_grid_hb(gh_args X) {
  static_assert(C == 2 || C == 4 || C == 8, "columns per lane");
  static_assert(G % K == 0 && G >= K && G <= 32, "rows per group");
  constexpr int WS = 64 * C, SR = G / K;  // doubles of a staged row, staged rows of a group
  constexpr int MHL = GH_MAXR / C;
  constexpr int SL = GH_SL, SLH = SL / 2, FSL = GH_FSL;
  __shared__ double2 lt[128];
  __shared__ __attribute__((aligned(16))) double xv[GH_PMAX][SL][MHL * C];
  __shared__ int xe[GH_PMAX][SL][MHL];
  __shared__ __attribute__((aligned(16))) double fv[FSL][MHL * C];
  __shared__ int fe[FSL][MHL];
  __shared__ int posted[GH_NWMAX], taken[GH_NWMAX], fetched, s_abort, s_awake;
  __shared__ unsigned s_ticket;
  __shared__ int w_se[GH_PMAX][64];
  __shared__ unsigned s_jq[GH_JQ + 1];  // first job of every strip's queue
  extern __shared__ __attribute__((aligned(16))) double gh_dyn[];  // per spine wave SR staged rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_ticket = atomicAdd(X.ticket, 1u);
  if (tid < 128) lt[tid] = X.lt[tid];
  if (X.jobs && tid <= GH_JQ) s_jq[tid] = X.jobs[tid];
  __syncthreads();
  // (the number of jobs comes from device memory: the list may have been built on the device a moment ago)
  const unsigned n_jobs = X.jobs ? min((unsigned)__builtin_amdgcn_readfirstlane((int)X.jobs[GH_JNUM]), X.job_cap) : 0u;
  const unsigned ticket = s_ticket;
  // (when the walk started -- its first launch's first workgroup -- for the host's look at what it took: stb_note_span)
  if (ticket == 0 && tid == 0) atomicCAS(reinterpret_cast<unsigned long long *>(X.hdr + STB_HDR_T0), 0ull, (unsigned long long)wall_clock64());
  const int R = X.R, HL = X.HL, U = X.U, NB = X.NB, P = X.P;
  const int UC = U * C;
  const int bB = X.b_begin, bE = X.b_end;
  // strip-major tickets: a strip only ever waits for strips with smaller tickets, which are running or done
  const int j = (int)(ticket / (unsigned)X.D);
  const int d = (int)(ticket % (unsigned)X.D);
  const int jw0 = j * P;
  if (tid < GH_NWMAX) {
    const int jw = jw0 + tid;
    const int b0 = (jw < X.JWa) ? max(gh_first_block(jw, UC, R), bB) : bE;
    posted[tid] = b0;
    taken[tid] = b0;
  }
  if (tid == 0) {
    fetched = max(gh_first_block(jw0, UC, R), bB);
    s_abort = 0;
    s_awake = (j == 0) ? 1 : 0;
  }
  __syncthreads();
  const unsigned who = (unsigned)(j | (d << 16));

  if (wave < P) {
    // ================= spine waves =================
    const int w = wave;
    const int jw = jw0 + w;
    if (jw >= X.JWa) return;
    bool aborted = false;
    auto wait_ge = [&](const int *cnt, int need, unsigned code) {
      if (aborted) return;
      for (int k = 0; k < GH_SPIN; k++) {
        if (lds_peek(cnt) >= need) {
          asm volatile("" ::: "memory");
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!chain_wait_slow(cnt, need, &s_abort, X.hdr, X.timeout, code, who, 1)) aborted = true;
      asm volatile("" ::: "memory");
    };
    const int b00 = gh_first_block(jw, UC, R);  // the strip's own first block
    const int b0 = max(b00, bB);                // ... and its first one in this phase
    // what the group loop below works on: the strip's own wave, or (after the strip has ended) a tile of another strip, of
    // any table, taken as a job
    double a = X.a[d];
    int mE0 = 2 + (jw * U - HL) * C;            // column of the wave's first element (halo included; <= 0 in strip 0's halo)
    int m0 = mE0 + lane * C;
    const size_t strip = ((size_t)d * (size_t)(X.JW + (int)n_jobs) + jw);  // the strip's pair of sums
    double v[C], coef[C];
    int ep = 1 + PC_BIAS;
#pragma unroll
    for (int i = 0; i < C; i++) {
      v[i] = 0.0;
      coef[i] = 0.0;
    }
    int one_hi = 0x3ff00000;
    asm volatile("" : "+v"(one_hi));
    const bool has_next = (w + 1 < P) && (jw + 1 < X.JWa);
    const int *left_cnt = (w == 0) ? &fetched : &posted[w > 0 ? w - 1 : 0];
    const double *left_v = (w == 0) ? &fv[0][0] : &xv[w > 0 ? w - 1 : 0][0][0];
    const int *left_e = (w == 0) ? &fe[0][0] : &xe[w > 0 ? w - 1 : 0][0][0];
    const int left_mask = (w == 0) ? FSL - 1 : SL - 1;
    unsigned long long *dbg = (X.dbg && d == 0 && lane == 0) ? X.dbg + (size_t)jw * (NB + 2) : nullptr;
    // records: only the next workgroup's fetcher reads them, and only the rightmost HL lanes of this workgroup's last strip
    const bool rec = (w == P - 1) && (jw + 1 < X.JWa) && lane >= U;
    const size_t rec_base = ((size_t)d * X.B + j) * (size_t)NB;
    unsigned long long *rec_v = X.ck_v + (rec_base * HL + (lane - U)) * C;
    unsigned *rec_e = X.ck_e + rec_base * HL + (lane - U);
    double s = 1.0;
    double *stage = gh_dyn + (size_t)w * (size_t)(SR * WS);
    int *se = &w_se[w][0];
    const int NQ = R / G;
    long long accK = 0;
    double accF = 0.0;
    // ---- a block's rows: the groups of G rows walked, every K-th row staged, the listed cells looked up.  `ti` is the
    // tile's entry, `wcur` the first word of its first group (asked for a group ago), `tin` the NEXT tile's entry (asked
    // for a block ago; 0: none), whose first word is asked for during the last group and left in `wcur`.  `only_walk`:
    // the tile's cells are somebody else's (a helper's). ----
    auto block_rows = [&](int b, unsigned ti, unsigned &wcur, unsigned tin, size_t item_blk, bool only_walk) {
      {
        const int dl = wave_shr1(ep, ep) - ep;
        s = ldexp(1.0, min(max(dl, -1100), 220));
      }
      // (the coefficient n - 1 - m a of the next row, from its closed form at every block: + 1 a row drifts)
#pragma unroll
      for (int i = 0; i < C; i++) coef[i] = (double)(1 + b * R) - (double)(m0 + i) * a;
      se[lane] = ep;
      const unsigned nw = ti & 63u;
      const unsigned *tw = X.dense + (size_t)(ti >> 6) * 64 + lane;  // this tile's words, this lane's
      for (int q = 0; q < NQ; q++) {
        // (this group's list is looked at BEFORE the next one is asked for: the compiler waits for every load under
        // way where a loaded register is first read inside a loop, the one just issued included)
        const bool listed = !only_walk && ((nw == 63u) || __ballot(wcur != 0) != 0);
        asm volatile("" ::: "memory");
        // the next group's first word is asked for now: it arrives while this group is walked
        unsigned wnext = 0;
        if (q + 1 < NQ) {
          if (nw != 0 && nw != 63u) wnext = tw[(size_t)(q + 1) * nw * 64];
        } else {
          tin = (unsigned)__builtin_amdgcn_readfirstlane((int)tin);
          if ((tin & 63u) != 0 && (tin & 63u) != 63u) wnext = X.dense[(size_t)(tin >> 6) * 64 + lane];
        }
        if (listed && !(X.diag & 8)) {
#pragma unroll
          for (int r = 0; r < G; r++) {
            gh_row<C>(v, coef, s);
            if ((r % K) == 0 && !GH_NOSTORE) {
              double *dst = stage + (r / K) * WS + lane * C;
#pragma unroll
              for (int i = 0; i < C; i += 2) *reinterpret_cast<gh_double2 *>(dst + i) = gh_double2{v[i], v[i + 1]};
            }
          }
          const int nrow0 = 2 + b * R + q * G;  // the row the group's first step produces
          if (X.diag & 4) {
          } else if (nw != 63u) {
            // word after word, the next one asked for before this one is worked on; a group's words are filled from
            // the front: the first empty one ends it
            const unsigned *gw = tw + (size_t)q * nw * 64;
            unsigned wd = wcur;
            for (unsigned k = 0; k < nw; k++) {
              if (k > 0 && __ballot(wd != 0) == 0) break;  // (looked at before the next load is issued: see above)
              asm volatile("" ::: "memory");
              unsigned wn = 0;
              if (k + 1 < nw) wn = gw[(size_t)(k + 1) * 64];
              gh_lookup<C, K>(wd != 0, wd & ((1u << (GH_PB(C) + 5)) - 1u), wd >> (GH_PB(C) + 5), stage, se, lt, a, mE0, nrow0, one_hi, accK, accF);
              wd = wn;
            }
          } else {
            // (a count of 2^19 or more in the tile, or more words than the layout takes: the lists in CSR form)
            const size_t item = item_blk + q;
            const unsigned e0 = X.item_ptr[item], e1 = X.item_ptr[item + 1];
            for (unsigned kk = e0; kk < e1; kk += 64) {
              unsigned pos = 0, cnt = 0;
              if (kk + lane < e1) {
                pos = X.ent_pos[kk + lane];
                cnt = X.ent_cnt[kk + lane];
              }
              gh_lookup<C, K>(kk + lane < e1, pos, cnt, stage, se, lt, a, mE0, nrow0, one_hi, accK, accF);
            }
          }
        } else {
          // (a group none of whose cells occurs, or whose cells are a helper's, is only walked)
#pragma unroll
          for (int r = 0; r < G; r++) gh_row<C>(v, coef, s);
        }
        wcur = wnext;
      }
    };
    // the two sums of what has been looked up, over the wave in a fixed tree: the same bits on every run.  (The exponent
    // sum is an integer well below 2^53: exact in a double, whatever the order.)
    auto sums_out = [&](double *out, bool add) {
      double kd = (double)accK;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        kd += __shfl_xor(kd, o);
        accF += __shfl_xor(accF, o);
      }
      if (lane == 0) {
        if (add) {  // (what the earlier phases summed: this wave is the strip's only writer in this launch)
          kd += out[0];
          accF = out[1] + accF;
        }
        out[0] = kd;
        out[1] = accF;
      }
      accK = 0;
      accF = 0.0;
    };
    // ---- helper jobs: a tile (strip jj, block jb, table dd) whose own wave only walked it, leaving the state of the
    // wave before the block; tickets go through the jobs in the order the tiles become ready, table after table.
    // Returns false when there is none left (or the launch is being given up). ----
    bool jobs_left = n_jobs != 0 && b00 >= bB && !(X.diag & 32);
    unsigned long long q_off = 0;  // queues this wave has found exhausted, or too late for it
    int q_next = (int)((ticket * 4u + (unsigned)w) % GH_JQ);
    // A queue per strip, a strip's jobs by block; a wave starts at a queue of its own and moves on when one is exhausted.
    // (Jobs are taken only by waves whose own strip has ENDED.  Taken before its start -- by waves the diagonal has not
    // reached yet, from the strips to their left -- they cost more than they bring: a wave that is inside a job when its
    // first halo arrives starts late, and everything to its right with it.  MI355X, 32 discounts, kernel ms: no jobs 1.23;
    // jobs up to 4 blocks before the wave's own first one 1.19, 12: 1.09, 24: 1.04, 48: 0.98; after the strip only: 0.93.)
    auto run_job = [&]() -> bool {
      unsigned job = 0;
      int dd = 0;
      bool got = false;
      for (int tries = 0; tries < GH_JQ && !got; tries++) {
        const int q = q_next;
        // (only strips of workgroups with LOWER tickets than this one -- the strip-major tickets' rule: whatever a wave
        // waits for belongs to a workgroup that started before its own and is running or through)
        if (q / P >= j) q_off |= 1ull << q;
        if ((q_off >> q) & 1ull) {
          q_next = (q_next + 1 == GH_JQ) ? 0 : q_next + 1;
          continue;
        }
        const unsigned j0 = s_jq[q], nq = s_jq[q + 1] - j0;
        const unsigned long long all = (unsigned long long)nq * (unsigned long long)X.D;
        unsigned t = 0;
        if (nq != 0 && lane == 0) t = atomicAdd(X.jticket + 16 * q, 1u);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (nq == 0 || (unsigned long long)t >= all) {
          q_off |= 1ull << q;
          q_next = (q_next + 1 == GH_JQ) ? 0 : q_next + 1;
          continue;
        }
        job = j0 + t / (unsigned)X.D;
        dd = (int)(t % (unsigned)X.D);
        got = true;
      }
      if (!got) return false;
      const unsigned jd = X.jobs[GH_JBASE + job];
      const int jj = (int)(jd & 0xffffu), jb = (int)(jd >> 16);
      const size_t slot = (size_t)dd * n_jobs + job;
      const unsigned tile = X.tile_off[jj + 1] + (unsigned)(jb - gh_first_block(jj, UC, R));
      unsigned ti = X.tinfo[tile];
      // the record: written a block after the tile's own wave walked past it
      {
        unsigned long long t_begin = 0;
        bool timing = false;
        unsigned idle = 0;
        while (__hip_atomic_load(X.jflag + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(8);
          // (A timeout of 0 means "give up at once" in every wait of this kernel and of k_fill_hb -- what
          // STB_CHAIN_TIMEOUT_MS=0 is for, the tests of the fallback.  This wait never sees it: a launch with a timeout of 0
          // has no helper jobs, stb_launch_grid.  Written with the test in the loop head like the others, this loop took the
          // kernel from 124 to 145 registers and the 64-discount evaluation from 1.29 to 1.42 ms: found by building the
          // round's commits side by side, tools/ab_grid.py with STB_LIB_PATH.)
          if ((++idle & 31) != 0) continue;
          if (!timing) {
            timing = true;
            t_begin = wall_clock64();
          }
          const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (err != 0 || (X.timeout != 0 && (unsigned long long)wall_clock64() - t_begin >= X.timeout)) {
            if (err == 0 && lane == 0) {
              __hip_atomic_store(X.hdr + 2, (unsigned)(jj | (dd << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(X.hdr + 1, 0xA00u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return false;
          }
        }
      }
      asm volatile("" ::: "memory");
      {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(X.jrec_v) + (slot * 64 + lane) * C;
#pragma unroll
        for (int i = 0; i < C; i++) v[i] = __longlong_as_double((long long)__hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        ep = __hip_atomic_load(X.jrec_e + slot * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ti = (unsigned)__builtin_amdgcn_readfirstlane((int)ti);
      unsigned wcur = 0;
      if ((ti & 63u) != 0 && (ti & 63u) != 63u) wcur = X.dense[(size_t)(ti >> 6) * 64 + lane];
      a = X.a[dd];
      mE0 = 2 + (jj * U - HL) * C;
      m0 = mE0 + lane * C;
      accK = 0;
      accF = 0.0;
      block_rows(jb, ti, wcur, 0u, (size_t)tile * (size_t)NQ, false);
      sums_out(X.dotp + ((size_t)dd * (size_t)(X.JW + (int)n_jobs) + (size_t)X.JW + job) * 2, false);
      return true;
    };
    // ---- the strip's own walk ----
    a = X.a[d];
    mE0 = 2 + (jw * U - HL) * C;
    m0 = mE0 + lane * C;
    ep = 1 + PC_BIAS;
    if (b00 < bB) {
      // (the state the previous phase left: plain loads, another launch wrote it)
#pragma unroll
      for (int i = 0; i < C; i++) v[i] = X.state_v[(((size_t)d * X.JW + jw) * 64 + lane) * C + i];
      ep = X.state_e[((size_t)d * X.JW + jw) * 64 + lane];
    } else {
#pragma unroll
      for (int i = 0; i < C; i++) v[i] = 0.0;
      if (jw == 0 && lane == HL - 1) v[C - 1] = ldexp(1.0, -1 - PC_BIAS);  // row 1: S^1_1 = 1
    }
    accK = 0;
    accF = 0.0;
    // the strip's listed cells, group after group in (block, group of G rows) order: NW words per lane, the first of
    // them asked for a group ahead and the tile's entry a block ahead (no address depends on a load just made)
    const unsigned tile0 = X.tile_off[jw + 1];  // tile (jw, b00)
    const size_t item0 = (size_t)tile0 * (size_t)NQ;
    unsigned ti = (b0 < bE) ? X.tinfo[tile0 + (unsigned)(b0 - b00)] : 0u;
    unsigned tj = (n_jobs && b0 < bE) ? X.tjob[tile0 + (unsigned)(b0 - b00)] : 0xffffffffu;  // the tile's place among the jobs, if it is one
    ti = (unsigned)__builtin_amdgcn_readfirstlane((int)ti);
    tj = (unsigned)__builtin_amdgcn_readfirstlane((int)tj);
    unsigned wcur = 0;
    if ((ti & 63u) != 0 && (ti & 63u) != 63u) wcur = X.dense[(size_t)(ti >> 6) * 64 + lane];
    unsigned pend = 0xffffffffu;  // a job whose record has been stored and whose "written" word is still to be set
    // what a block begins with and needs no halo for: the hand-over to the right neighbour in the workgroup and, for the
    // next workgroup, the block's record -- both of the wave as it stands
    auto block_top = [&](const int b) {
      // ---- the rightmost HL lanes, for the right neighbour in this workgroup ----
      if (has_next) {
        if ((b & (SLH - 1)) == 0 || b == b0) wait_ge(&taken[w + 1], b - SLH, 0x400u);
        if (lane >= U) {
          double *dst = &xv[w][b & (SL - 1)][(lane - U) * C];
#pragma unroll
          for (int i = 0; i < C; i++) dst[i] = v[i];
          xe[w][b & (SL - 1)][lane - U] = ep;
        }
        lds_post(&posted[w], b + 1);
      }
      // ---- ... and for the next workgroup: the record of the block ----
      if (rec) {
        unsigned long long *dst = rec_v + (size_t)b * (size_t)(HL * C);
#pragma unroll
        for (int i = 0; i < C; i += 2)
          gh_store_wt16(dst + i, (unsigned long long)__double_as_longlong(v[i]) | GH_WRITTEN,
                        (unsigned long long)__double_as_longlong(v[i + 1]) | GH_WRITTEN);
        __hip_atomic_store(rec_e + (size_t)b * (size_t)HL, (unsigned)ep + GH_EOFF32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    // (the first block's top before the wave dozes for its first halo: what a strip loses at its start it never makes up
    // -- its neighbours walk at the same pace -- see k_fill_hb; a phase that continues a strip renormalises first)
    const bool top_done = b0 == b00 && b0 < bE;
    if (top_done) block_top(b0);
    while (!lds_peek(&s_awake) && !lds_peek(&s_abort)) __builtin_amdgcn_s_sleep(2);
    if (w > 0)
      while (lds_peek(left_cnt) < b0 + 1 && !lds_peek(&s_abort)) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_s_setprio(3);
    if (dbg) dbg[0] = wall_clock64();
    for (int b = b0; b < bE; b++) {
      if (b > b00) gh_renorm<C>(v, ep);
      if (b > b0 || !top_done) block_top(b);
      // ---- the halo: the left neighbour's rightmost HL lanes as they stand before the block ----
      if (jw > 0) {
        double hv[C];
        int he = 0;
        const double *src = left_v + (b & left_mask) * (MHL * C) + lane * C;
        const int *srce = left_e + (b & left_mask) * MHL + lane;
        const int seen = aborted ? 0x7fffffff : lds_peek(left_cnt);
        asm volatile("" ::: "memory");
        if (lane < HL) {
#pragma unroll
          for (int i = 0; i < C; i++) hv[i] = src[i];
          he = *srce;
        }
        if (seen < b + 1) {
          wait_ge(left_cnt, b + 1, 0x100u);
          if (lane < HL) {
#pragma unroll
            for (int i = 0; i < C; i++) hv[i] = src[i];
            he = *srce;
          }
        }
        if (lane < HL) {
#pragma unroll
          for (int i = 0; i < C; i++) v[i] = hv[i];
          ep = he;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_post(&taken[w], b + 1);
      }
      if (dbg) dbg[1 + b] = wall_clock64();
      // ---- a tile that is a job: the wave as it stands goes to the job's record.  Its "written" word follows a block
      // later, when the stores have long been through (waiting for them here would cost a round trip to memory). ----
      if (pend != 0xffffffffu) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(X.jflag + (size_t)d * n_jobs + pend, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend = 0xffffffffu;
      }
      const bool is_job = tj != 0xffffffffu;
      if (is_job) {
        const size_t slot = (size_t)d * n_jobs + tj;
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(X.jrec_v) + (slot * 64 + lane) * C;
#pragma unroll
        for (int i = 0; i < C; i += 2)
          gh_store_wt16(dst + i, (unsigned long long)__double_as_longlong(v[i]), (unsigned long long)__double_as_longlong(v[i + 1]));
        __hip_atomic_store(X.jrec_e + slot * 64 + lane, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend = tj;
      }
      unsigned tin = 0, tjn = 0xffffffffu;
      if (b + 1 < bE) {
        tin = X.tinfo[tile0 + (unsigned)(b + 1 - b00)];  // the next tile's entry
        if (n_jobs) tjn = X.tjob[tile0 + (unsigned)(b + 1 - b00)];
      }
      block_rows(b, ti, wcur, tin, item0 + (size_t)(b - b00) * NQ, is_job);
      ti = (unsigned)__builtin_amdgcn_readfirstlane((int)tin);
      tj = (unsigned)__builtin_amdgcn_readfirstlane((int)tjn);
    }
    if (pend != 0xffffffffu) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(X.jflag + (size_t)d * n_jobs + pend, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sums_out(X.dotp + strip * 2, b00 < bB);
    if (bE < NB) {
#pragma unroll
      for (int i = 0; i < C; i++) X.state_v[(((size_t)d * X.JW + jw) * 64 + lane) * C + i] = v[i];
      X.state_e[((size_t)d * X.JW + jw) * 64 + lane] = ep;
    }
    if (dbg) dbg[NB + 1] = wall_clock64();
    __builtin_amdgcn_s_setprio(0);
    // ---- ... and after it: the jobs that are left ----
    while (jobs_left && !lds_peek(&s_abort))
      if (!run_job()) jobs_left = false;
  } else if (wave == P && j > 0 && jw0 < X.JWa) {
    // ================= fetcher: the left workgroup's last strip's records -> LDS, for spine wave 0 =================
    const int gl = (HL <= 16) ? 16 : 32;
    const int grp = lane / gl, sub = lane % gl, ngrp = 64 / gl;
    const bool act = sub < HL;
    const size_t rec_left = ((size_t)d * X.B + (j - 1)) * (size_t)NB;
    int bb = max(gh_first_block(jw0, UC, R), bB);  // blocks below it are delivered
    bool woke = false;
    unsigned long long t_begin = 0;
    bool timing = false;
    unsigned idle = 0;
    while (bb < bE) {
      const int mb = bb + grp;
      const bool want = act && mb < bE;
      unsigned long long bv[C];
      unsigned be = 1;
#pragma unroll
      for (int i = 0; i < C; i++) bv[i] = 1;
      if (want) {
        const unsigned long long *src = X.ck_v + ((rec_left + mb) * HL + sub) * C;
#pragma unroll
        for (int i = 0; i < C; i++) bv[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        be = __hip_atomic_load(X.ck_e + (rec_left + mb) * HL + sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int tk = lds_peek(&taken[0]);
      bool have = be != 0;
#pragma unroll
      for (int i = 0; i < C; i++) have = have && bv[i] != 0;
      const unsigned long long miss = ~__ballot(have);
      int k = miss ? (int)(__builtin_ctzll(miss) / gl) : ngrp;
      k = min(k, bE - bb);
      k = min(k, tk + FSL - bb);
      if (k > 0) {
        if (want && grp < k) {
          double *dst = &fv[mb & (FSL - 1)][sub * C];
#pragma unroll
          for (int i = 0; i < C; i++) dst[i] = __longlong_as_double((long long)(bv[i] & ~GH_WRITTEN));
          fe[mb & (FSL - 1)][sub] = (int)(be - GH_EOFF32);
        }
        bb += k;
        lds_post(&fetched, bb);
        if (!woke) {
          woke = true;
          lds_post(&s_awake, 1);
        }
        timing = false;
        idle = 0;
        continue;
      }
      if (woke)  // (until the first record has come the looks follow each other as fast as they return)
        for (int i = 0; i < X.poll_nap; i++) __builtin_amdgcn_s_sleep(1);
      if ((++idle & 31) != 0 && X.timeout != 0) continue;
      if (!timing) {
        timing = true;
        t_begin = wall_clock64();
      }
      const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (err != 0 || lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin >= X.timeout) {
        if (lane == 0) {
          __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (err == 0) {
            __hip_atomic_store(X.hdr + 2, who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(X.hdr + 1, 0x900u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        lds_post(&fetched, 0x7fffffff);  // release the spine: it runs on with stale halos
        lds_post(&s_awake, 1);
        break;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side

unsigned stb_grid_job_cap(int C, int D, unsigned n_tiles, int phases);
int stb_grid_geometry(unsigned N, unsigned M, int D, grid_geom *out) {
  grid_geom g;
  memset(&g, 0, sizeof(g));
  *out = g;
  if (N < 3 || M < 2 || D < 1 || N >= (1u << 20)) return 1;
  const int cus = stb_cu_count();
  const unsigned cmax = (M < N - 1) ? M : N - 1;  // columns 2..cmax hold stored cells: elements 0 .. cmax - 2
  int Pc = stb_period_rows(N);
  const int Penv = stb_env_int("STB_FILL_P", 0);
  if (Penv > 0 && Penv < Pc) Pc = Penv;
  int R = stb_env_int("STB_GRID_ROWS", GH_MAXR);
  if (R > Pc) R = Pc;
  if (R > GH_MAXR) R = GH_MAXR;
  R = R / 8 * 8;
  if (R < 8) return 1;
  // 2 columns per lane -- the faster walk, 32 against 50 cycles a row -- while every strip of the batch gets a SIMD
  // to itself (one workgroup of 4 strips per compute unit); 4 beyond: 2.6 times fewer waves for the same columns
  {
    const int HL2 = R / 2, U2 = 64 - HL2;
    const uint64_t strips2 = (cmax - 1 + U2 * 2 - 1) / (U2 * 2);
    const uint64_t wgs2 = (strips2 + 3) / 4 * (uint64_t)D;
    // ... 8 columns per lane (round 5; a strip is 58 own lanes = 464 columns, a table of 10^4 columns 22 strips) only when
    // asked for, STB_GRID_C=8 or a threshold of 4-column workgroups in STB_GRID_C8_WGS.  Built on the review's advice for
    // 64 discounts -- 704 workgroups of 3 waves, all resident, where the 4-column form's 784 do not fit and the strips
    // right of column 6656 start when the first end -- and on tools/ubench/rowpace.hip (chip full: 0.104 against 0.197 ns
    // per own column and row).  Measured (MI355X, 64 discounts x 10^6 pairs, N = 10^4, kernel ms, profiles/r05_grid_c8.txt):
    // 1.57 against the 4-column form's 1.42; the bare walk 0.98 against 0.80.  Why: a table is a CHAIN of N row steps
    // whatever the strip width, and under this load (two walking waves a SIMD, clock ~2.0 GHz) a row of 8 columns per
    // lane takes 62 ns against 41 -- strip 0 alone is 0.65 ms against 0.41 -- so what the wider strip saves in
    // instructions and residency it loses on every table's critical path.  The form is kept (tested in every strip
    // shape) for batches whose chain is not what bounds them.
    const int HL4 = R / 4, U4 = 64 - HL4;
    const uint64_t strips4 = (cmax - 1 + U4 * 4 - 1) / (U4 * 4);
    const uint64_t wgs4 = (strips4 + 3) / 4 * (uint64_t)D;
    const int c8_from = stb_env_int("STB_GRID_C8_WGS", 0);  // 4-column workgroups from which the 8-column form takes over (0: never)
    g.C = stb_env_int("STB_GRID_C", wgs2 <= (uint64_t)cus ? 2 : ((c8_from > 0 && wgs4 > (uint64_t)c8_from && R % 8 == 0) ? 8 : 4));
    if (g.C != 2 && g.C != 4 && g.C != 8) g.C = 2;
    if (R % g.C != 0) return 1;
  }
  g.R = R;
  g.HL = R / g.C;
  g.U = 64 - g.HL;
  const int UC = g.U * g.C;
  g.JW = (int)((cmax - 1 + UC - 1) / UC);
  if (g.JW < 1) g.JW = 1;
  g.NB = (int)((N - 1 + R - 1) / R);
  if (g.JW >= 65535 || g.NB >= 65536) return 1;
  // strips per workgroup: 4, a walking wave per SIMD (MI355X, 64 discounts x 10^6 pairs, N = M = 10^4, groups of 12
  // rows, kernel ms: 3 strips 1.84, 4: 1.73, 5: 1.81, 6: 1.90, 7: 1.85)
  // (8 columns per lane: 3 -- a staged row is 4 KB; 512 workgroups of 4 waves for 64 tables of 10^4 columns, two a unit)
  g.P = stb_env_int("STB_GRID_P", g.C == 8 ? 3 : 4);
  if (g.P < 1 || g.P > GH_PMAX) g.P = g.C == 8 ? 3 : 4;
  g.B = (g.JW + g.P - 1) / g.P;
  // rows per group: a look-up pass costs the same for 1 or 64 listed cells (10^6 pairs over a 10^4 x 10^4 table: 1.6
  // a row in a strip of 80 columns, 4.2 in one of 208), and half of a group's rows are staged in LDS per walking wave
  // (1 KB a row with 2 columns per lane, 2 KB with 4) beside ~16 KB of rings: two workgroups must fit a compute unit's
  // 160 KB once there are more workgroups than units
  {
    // (4 columns per lane, 64 discounts as above: groups of 8 rows 1.85, 12: 1.73, 16: 1.97)
    const bool two_per_cu = (int64_t)g.B * D > cus;
    // (... with the dense lists: 12 rows 1.57, 16: 1.46, 24: 1.39)
    int Gd = (g.C == 2) ? 24 : (g.C == 8 ? 16 : ((g.P > 4 && two_per_cu) ? 8 : 24));
    g.K = (g.C >= 4) ? (stb_env_int("STB_GRID_K", 4) == 2 ? 2 : 4) : 2;  // every K-th row of a group is staged
    Gd = stb_env_int("STB_GRID_G", Gd);
    while (Gd > 2 && (R % Gd != 0 || (Gd != 8 && Gd != 12 && Gd != 16 && Gd != 24))) Gd -= 2;
    if (Gd != 8 && Gd != 12 && Gd != 16 && Gd != 24) Gd = 8;
    if (R % Gd != 0) return 1;  // (R is a multiple of 8: cannot happen)
    g.G = Gd;
    g.NQ = R / Gd;
  }
  uint64_t nt = 0;
  for (int j = 0; j < g.JW; j++) {
    const int b0 = gh_first_block(j, UC, R);
    if (b0 >= g.NB) return 1;  // (cannot happen: column 2 + j U C <= N - 1)
    nt += (uint64_t)(g.NB - b0);
  }
  if (nt * g.NQ >= (1ull << 31)) return 1;
  g.n_tiles = (unsigned)nt;
  // Phases.  One walking wave keeps a SIMD's vector issue busy and every strip of a table moves at the pace of the
  // slowest, while the strips right of the diagonal have nothing to do yet: with more strips than SIMDs the launch is
  // cut into phases of blocks, each launched with the strips the diagonal reaches before its end -- evenly spread
  // by the dispatcher because they all work for the whole phase.
  {
    // (Measured, 64 discounts as above: 1 launch 1.62, 2: 1.66, 4: 1.73, 6: 1.77 -- what the phases gain in balance
    // they lose at their starts, where every table's chain of strips builds up again: one launch unless asked.)
    int ph = stb_env_int("STB_GRID_PHASES", 1);
    if (ph < 1) ph = 1;
    if (ph > GH_MAXPH) ph = GH_MAXPH;
    if (ph > g.NB) ph = g.NB;
    g.phases = ph;
  }
  // Helper jobs (one launch only): tiles whose listed cells the strip's own wave leaves to waves that have nothing to
  // do yet.  A job's record is the whole wave before the block, 64 (4 + 8 C) bytes: at most GH_JOB_MB of them per launch.
  g.job_cap = stb_grid_job_cap(g.C, D, g.n_tiles, g.phases);
  size_t o = 256 + 64 * (GH_MAXPH + GH_JQ);  // header: error words, then a ticket word per phase -- and one per job queue -- on a line of its own
  o = stb_align_up(o, 256);
  g.off_cke = o;
  o += stb_align_up((size_t)D * g.B * g.NB * g.HL * sizeof(unsigned), 256);
  g.off_ckv = o;
  o += stb_align_up((size_t)D * g.B * g.NB * g.HL * g.C * 8, 256);
  g.off_jflag = o;
  o += stb_align_up((size_t)D * g.job_cap * sizeof(unsigned), 256);
  g.zero_bytes = o;
  g.off_state_e = o;
  o += stb_align_up((size_t)D * g.JW * 64 * sizeof(int), 256);
  g.off_state_v = o;
  o += stb_align_up((size_t)D * g.JW * 64 * g.C * 8, 256);
  g.off_jrec_e = o;
  o += stb_align_up((size_t)D * g.job_cap * 64 * sizeof(int), 256);
  g.off_jrec_v = o;
  o += stb_align_up((size_t)D * g.job_cap * 64 * g.C * 8, 256);
  g.bytes = o;
  g.ok = 1;
  *out = g;
  return 0;
}

// the strip shape the grid form takes for these sizes: columns per lane, rows per group, every K-th row staged (diagnostics:
// bench.py names the kernel instantiation with it); non-zero where the form does not apply
extern "C" int stb_grid_shape(unsigned N, unsigned M, int D, int *C_out, int *G_out, int *K_out) {
  grid_geom g;
  if (stb_grid_geometry(N, M, D, &g)) return 1;
  if (C_out) *C_out = g.C;
  if (G_out) *G_out = g.G;
  if (K_out) *K_out = g.K;
  return 0;
}

unsigned stb_grid_job_cap(int C, int D, unsigned n_tiles, int phases) {
  if (phases != 1 || D < 1 || !stb_env_int("STB_GRID_HELP", 1)) return 0;
  uint64_t rec = 64ull * (4 + 8 * C), room = ((uint64_t)stb_env_int("STB_GRID_JOB_MB", GH_JOB_MB) << 20) / ((uint64_t)D * rec);
  // (... and what is taken off every table's critical path comes back as a tail in which all waves work side by side: a
  // job is ~7 us of a wave, a table has ~50 of them)
  const uint64_t most = (uint64_t)stb_env_int("STB_GRID_JOBS", GH_JOBS);
  if (room > most) room = most;
  return (unsigned)(room < n_tiles ? room : n_tiles);
}

size_t stb_grid_workspace(unsigned N, unsigned M, int D) {
  // (the strip shape depends on the batch and on tunables: room for either number of columns per lane)
  size_t need = 0;
  grid_geom g;
  if (stb_grid_geometry(N, M, D, &g) == 0) need = g.bytes;
  const int cs[3] = {2, 4, 8};
  const int Pc = stb_period_rows(N);
  int R = GH_MAXR < Pc ? GH_MAXR : Pc;
  R = R / 8 * 8;
  if (R < 8 || N < 3 || M < 2) return need;
  const unsigned cmax = (M < N - 1) ? M : N - 1;
  for (int c : cs) {
    if (R % c != 0) continue;
    const int HL = R / c, U = 64 - HL, UC = U * c;
    const size_t JW = (cmax - 1 + UC - 1) / UC, NB = (N - 1 + R - 1) / R, B = (JW + (c == 8 ? 0 : 3)) / (c == 8 ? 1 : 4);
    const size_t jobs = stb_grid_job_cap(c, D, (unsigned)(JW * NB), 1);  // (records and "written" words of the helper jobs)
    const size_t b = 4096 + 64 * (GH_MAXPH + GH_JQ) + (size_t)D * B * NB * HL * (4 + 8 * c) + (size_t)D * JW * 64 * (4 + 8 * c) + 2048 +
                     (size_t)D * jobs * (64 * (4 + 8 * c) + 4) + 1024;
    if (b > need) need = b;
  }
  return need + 256;
}

// first tile of every strip (entry j + 1; a strip's tiles are its blocks from its first one on), [JW + 2] words
void stb_grid_tile_offsets(const grid_geom &g, std::vector<unsigned> &off) {
  off.assign((size_t)g.JW + 2, 0u);
  const int UC = g.U * g.C;
  unsigned o = 0;
  for (int j = 0; j < g.JW; j++) {
    off[j + 1] = o;
    o += (unsigned)(g.NB - gh_first_block(j, UC, g.R));
  }
  off[0] = o;  // (the total, for whoever wants to check)
}

template <int C, int K>
static int gh_launch(const gh_args &X, int G, unsigned grid, int P, hipStream_t st) {
  const size_t shm = (size_t)P * (size_t)(G / K) * 64 * C * sizeof(double);  // (every K-th row of a group is staged)
  size_t ask = shm;
  // While there are no more workgroups than compute units each should have a unit to itself (two walking waves on
  // one SIMD share its issue, and every strip moves at the pace of the slowest): ask for more than half of a unit's LDS.
  if ((int)grid <= stb_cu_count() && ask < 84 * 1024 && stb_env_int("STB_GRID_ALONE", 1)) ask = 84 * 1024;
  const dim3 block(64 * (P + 1));
  switch (G) {
    case 8: STB_LAUNCH_SHM((k_grid_hb<C, 8, K>), dim3(grid), block, ask, st, X); break;
    case 12: STB_LAUNCH_SHM((k_grid_hb<C, 12, K>), dim3(grid), block, ask, st, X); break;
    case 16: STB_LAUNCH_SHM((k_grid_hb<C, 16, K>), dim3(grid), block, ask, st, X); break;
    case 24: STB_LAUNCH_SHM((k_grid_hb<C, 24, K>), dim3(grid), block, ask, st, X); break;
    default: return stb_fail("stb_groups_aterms: no grid kernel for groups of %d rows", G);
  }
  return 0;
}

int stb_launch_grid(fill_args &A, int D, char *ws, size_t ws_left, const dot_request *dot, unsigned **hdr_out, hipStream_t st) {
  const unsigned N = A.N, M = A.M;
  grid_geom g;
  if (stb_grid_geometry(N, M, D, &g)) return stb_fail("stb_groups_aterms: the grid form does not take N=%u M=%u D=%d", N, M, D);
  if (g.bytes > ws_left) return stb_fail("stb_groups_aterms: workspace too small for the grid form (%zu > %zu)", g.bytes, ws_left);
  if (!dot || !dot->item_ptr || !dot->tile_off || !dot->dense || !dot->tinfo || dot->col0 != 4)
    return stb_fail("stb_groups_aterms: the grid form sums over cell lists built for its strips");
  if (dot->geom_C != g.C || dot->geom_R != g.R || dot->geom_G != g.G)
    return stb_fail("stb_groups_aterms: cell lists built for %d columns a lane, blocks of %d rows, groups of %d; the walk is %d, %d, %d",
                    dot->geom_C, dot->geom_R, dot->geom_G, g.C, g.R, g.G);
  gh_args X;
  memset(&X, 0, sizeof(X));
  X.a = A.a;
  X.lt = A.lt;
  X.hdr = (unsigned *)ws;
  X.ck_e = (unsigned *)(ws + g.off_cke);
  X.ck_v = (unsigned long long *)(ws + g.off_ckv);
  X.state_e = (int *)(ws + g.off_state_e);
  X.state_v = (double *)(ws + g.off_state_v);
  X.tile_off = dot->tile_off;
  X.dense = dot->dense;
  X.tinfo = dot->tinfo;
  X.item_ptr = dot->item_ptr;
  X.ent_pos = dot->ent_pos;
  X.ent_cnt = dot->ent_cnt;
  X.dotp = dot->dotp;
  X.N = N;
  X.M = M;
  X.D = D;
  X.B = g.B;
  X.JW = g.JW;
  X.NB = g.NB;
  X.P = g.P;
  X.R = g.R;
  X.HL = g.HL;
  X.U = g.U;
  X.timeout = (unsigned long long)stb_env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;  // wall_clock64: 100 MHz
  X.poll_nap = stb_env_int("STB_HB_POLL_NAP", 4);
  if (X.poll_nap < 1) X.poll_nap = 1;
  X.diag = stb_env_int("STB_GRID_DIAG", 0);
  if (dot->jobs && !dot->tjob) return stb_fail("stb_groups_aterms: a job list without the tiles' places in it");
  X.jobs = dot->jobs;
  X.tjob = dot->tjob;
  X.job_cap = g.job_cap;
  if (!g.job_cap || X.timeout == 0) X.jobs = nullptr;  // (a timeout of 0: every wait gives up at once -- and the wait for a job's record is not entered)
  X.jticket = X.hdr + 64 + 16 * GH_MAXPH;  // (GH_JQ lines)
  X.jflag = (unsigned *)(ws + g.off_jflag);
  X.jrec_e = (int *)(ws + g.off_jrec_e);
  X.jrec_v = (double *)(ws + g.off_jrec_v);
  // (partial sums: two per strip and per job; the number of jobs is read from the list by whoever sums them up)
  const_cast<dot_request *>(dot)->parts_per_table = g.JW;
  const_cast<dot_request *>(dot)->parts_extra_dev = X.jobs ? X.jobs + GH_JNUM : nullptr;
  if (dot->dotp_cap && (size_t)D * (size_t)(g.JW + (int)g.job_cap) * 2 > dot->dotp_cap) return stb_fail("stb_groups_aterms: partial-sum buffer too small");
  const char *tl_file = getenv("STB_HB_TIMELINE");
  const size_t dbg_words = (size_t)g.JW * (g.NB + 2);
  if (tl_file && *tl_file) {
    HIPCHK(hipMalloc((void **)&X.dbg, dbg_words * 8));
    HIPCHK(hipMemsetAsync(X.dbg, 0, dbg_words * 8, st));
  }
  if (stb_a_flush(A, st)) return 1;
  if (!dot || dot->ws_zero < g.zero_bytes) HIPCHK(hipMemsetAsync(ws, 0, g.zero_bytes, st));
  if (dot) const_cast<dot_request *>(dot)->zero_bytes = g.zero_bytes;
  *hdr_out = X.hdr;
  const int UC = g.U * g.C;
  for (int ph = 0; ph < g.phases; ph++) {
    X.b_begin = (int)((int64_t)g.NB * ph / g.phases);
    X.b_end = (int)((int64_t)g.NB * (ph + 1) / g.phases);
    if (X.b_end <= X.b_begin) continue;
    int jwa = 0;
    while (jwa < g.JW && gh_first_block(jwa, UC, g.R) < X.b_end) jwa++;
    X.JWa = jwa;
    X.ticket = X.hdr + 64 + 16 * ph;
    const unsigned grid = (unsigned)((jwa + g.P - 1) / g.P) * (unsigned)D;
    const int rc = (g.C == 2) ? gh_launch<2, 2>(X, g.G, grid, g.P, st)
                   : (g.C == 8) ? (g.K == 4 ? gh_launch<8, 4>(X, g.G, grid, g.P, st) : gh_launch<8, 2>(X, g.G, grid, g.P, st))
                                : (g.K == 4 ? gh_launch<4, 4>(X, g.G, grid, g.P, st) : gh_launch<4, 2>(X, g.G, grid, g.P, st));
    if (rc) return 1;
  }
  HIPCHK(hipGetLastError());
  if (X.dbg) {
    HIPCHK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(dbg_words);
    HIPCHK(hipMemcpy(h.data(), X.dbg, dbg_words * 8, hipMemcpyDeviceToHost));
    (void)hipFree(X.dbg);
    FILE *f = fopen(tl_file, "wb");
    if (f) {
      const int hd[8] = {g.JW, g.NB, 0, g.C, g.P, g.R, g.U, D};  // (the format of tools/timeline_hb.py, no tile part)
      fwrite(hd, sizeof(int), 8, f);
      fwrite(h.data(), 8, dbg_words, f);
      fclose(f);
    }
  }
  return 0;
}

// fill_chain.hip -- the chain forms: ONE launch per fill, column blocks keep their columns for all
// rows and hand their right edge to the next block through 8-byte granules in HBM.
//
//   k_fill_chain   S table of S_remake_part (reference lib/stable.c:321-388), also as the DOT kernel
//                  that sums count * log S for aterms (lib/samplea.c:68-80) without storing a table
//   k_fillv_chain  V table (lib/stable.c:451-482), bit-identical to the reference

#include <type_traits>

#include "fill_chain.h"

// ---- chain form: ONE launch per fill, column blocks hand their right edge to the next block ------
//
// The forms above advance every strip by R rows per launch and recompute an R-column halo so that
// strips never talk to each other.  Here a column block owns its 64*P columns for ALL rows: what a
// column needs from its left neighbour (one row up) travels wave to wave, so there is no halo, no
// launch per row block and no frontier round trip.  Nothing in the steady state is a barrier: the
// waves of a block run free and meet through counters in LDS.
//
//  * P producer waves, 64 columns each (one per lane, DPP shift inside the wave), carry the
//    recurrence and write raw significands into an LDS ring of CH_RD trips (a trip = CH_U rows).
//    Producer w reads the last column of producer w-1 from that ring, one trip behind it.
//  * NC consumer waves take (trip, slice) items round-robin, turn 8 rows x 64 columns into logs
//    (stage-major, as in k_fill_pc) and store them; they are off the producers' critical path.
//  * The publisher wave writes the block's last column to global memory as 8-byte granules: the raw
//    double per row (-0.0 for an exact zero, so that 0 = "not written yet") and the lane exponent
//    per trip, with write-through stores.  The fetcher wave of the next block reads 128 rows per
//    round trip with L1-bypassing loads, delivers the leading complete trips into an LDS ring and
//    re-reads the rest.  A granule is its own flag (one aligned 8-byte store), so the hand-off needs
//    no fence and no ordering.
//
// Block (d, j) starts at the trip in which the diagonal enters its first column and lags its left
// neighbour by the hand-off latency; a triangular table starts column block j at row 64*P*j anyway.
// Blocks take their (j, d) from an atomic ticket, j-major, so a block only ever waits for a block
// with a smaller ticket, i.e. one that is running or done: forward progress does not depend on
// dispatch order or on how many blocks are resident.  Every wait is bounded by wall-clock time; on
// expiry the block records an error, stops waiting and runs to its end (stb_fill_status).
// DOT: the logs are not stored; each is multiplied by the cell's occurrence count and summed (the
// whole of aterms' table part, lib/samplea.c:68-80, without a table in memory or a second pass).
//   DOT = 1: dense -- a count slab in the table's layout; every cell's log is computed.
//   DOT = 2: sparse -- per (trip, 64-column slice) item the list of cells that occur at all
//            (position in the 8 x 64 tile + count): only their logs are computed, empty items are
//            skipped without even waiting for the producer.
template <int P, int NC, int NF, int DOT>
__global__ __launch_bounds__(64 * (P + NC + 1 + NF)) void k_fill_chain(fill_args A, chain_args X) {
  constexpr int U = CH_U, RD = CH_RD, RE = CH_RE;
  constexpr int OW = 64 * P;  // columns of a block
  static_assert(P >= 1 && P <= NC && NF >= 1 && P + NC + 1 + NF <= 16, "block shape");
  __shared__ double2 lt[128];
  __shared__ __attribute__((aligned(16))) double vbuf[RD][U][OW];
  __shared__ int ebuf[4][OW];
  __shared__ int slot_p[RD][P];
  __shared__ __attribute__((aligned(16))) double edge_in[RE * U];
  __shared__ int edge_e[RE];
  __shared__ int prod_done[P], cons_cnt[NC], pub_done, edge_ready, s_abort;
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) s_ticket = atomicAdd(X.hdr, 1u);
  if (tid < 128) lt[tid] = A.lt[tid];
  for (int i = tid; i < RE * U; i += blockDim.x) edge_in[i] = 0.0;
  __syncthreads();
  const int j = (int)(s_ticket / (unsigned)X.D);
  const int d = (int)(s_ticket % (unsigned)X.D);
  if (j >= X.B) return;  // (never: the grid is exactly B*D blocks)

  const unsigned N = A.N, M = A.M;
  const int TP = X.TP, G = X.G;
  const int c0 = 1 + j * OW;  // first column of the block
  auto first_trip = [&](int w) {  // trip in which the diagonal reaches the first column of slice w
    const int c = c0 + 64 * w;
    return (c <= 3) ? 0 : (c - 3) / U;
  };
  const int g0b = first_trip(0);
  const bool has_left = j > 0, has_right = j < X.B - 1;
  double *table = A.tables + (uint64_t)d * A.tstride;
  if (tid < P) prod_done[tid] = first_trip(tid);
  if (tid < NC) cons_cnt[tid] = 0;
  if (tid == 0) {
    pub_done = has_right ? first_trip(P - 1) : 0x7fffffff;
    edge_ready = has_left ? g0b : 0x7fffffff;
    s_abort = 0;
  }
  __syncthreads();

  bool aborted = false;
  // wait until *cnt >= need; bounded: on expiry (or when another wave gave up) stop waiting for good
  auto wait_ge = [&](const int *cnt, int need, unsigned code) {
    if (aborted || lds_peek(cnt) >= need) return;
    const unsigned long long t_begin = wall_clock64();
    for (;;) {
      __builtin_amdgcn_s_sleep(1);
      if (lds_peek(cnt) >= need) {
        return;
      }
      if (lds_peek(&s_abort)) break;
      if ((unsigned long long)wall_clock64() - t_begin > X.timeout) {
        if (lane == 0) {
          __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (__hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __hip_atomic_store(X.hdr + 2, (unsigned)(j | (d << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(X.hdr + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        break;
      }
    }
    aborted = true;
  };

  if (wave < P) {
    // ================= producers =================
    __builtin_amdgcn_s_setprio(3);
    const int w = wave;
    const int g0w = first_trip(w);
    const int col = 64 * w + lane;  // my column inside the block
    const int c = c0 + col;
    const double a = A.a[d];
    // row 2 of the table: S^2_1 = 1 - a, S^2_2 = 1; everything else starts above the diagonal
    double v = (c == 1) ? ldexp(1.0 - a, -1 - PC_BIAS) : (c == 2) ? ldexp(1.0, -1 - PC_BIAS) : 0.0;
    double coef = (double)(2 + g0w * U) - (double)c * a;  // n - 1 - c a for the first row of trip g0w
    double s = 1.0;
    int ep = 1 + PC_BIAS;
    int p = g0w / TP, tin = g0w - p * TP;
    // the consumer item that last used the ring slot a trip is about to overwrite
    int chk_i = (g0w - RD - g0b) * P + w;
    int chk_c = (chk_i >= 0) ? chk_i % NC : w, chk_k = (chk_i >= 0) ? chk_i / NC : 0;  // (first i >= 0 is w)
    const int *left_cnt = (w == 0) ? &edge_ready : &prod_done[w - 1];
    const int *next_cnt = (w < P - 1) ? &prod_done[w + 1] : &pub_done;
    // What a trip needs from the other waves -- the left neighbour's progress, the consumers' and
    // the right neighbour's progress on the ring slot it overwrites -- and its U left inputs are
    // read one trip AHEAD, under the previous trip's arithmetic, so that no LDS round trip sits on
    // the row chain.  The inputs are speculative: they are valid if the counter read BEFORE them
    // (LDS is in order) already covered the trip; otherwise wait and read again.
    int n_left, n_cons, n_next;
    double ne[U];
    auto load_left = [&](double(&x)[U], int g) {
      if (w == 0) {
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = edge_in[(g & (RE - 1)) * U + u];
      } else {
        x[0] = vbuf[(g - 1) & (RD - 1)][U - 1][64 * w - 1];
#pragma unroll
        for (int u = 1; u < U; u++) x[u] = vbuf[g & (RD - 1)][u - 1][64 * w - 1];
      }
    };
    auto look_ahead = [&](int g) {
      n_left = lds_peek(left_cnt);
      n_cons = lds_peek(&cons_cnt[chk_c]);
      n_next = lds_peek(next_cnt);
      asm volatile("" ::: "memory");
      load_left(ne, g);
    };
    look_ahead(g0w);
    for (int g = g0w; g < G; g++) {
      double e[U];
#pragma unroll
      for (int u = 0; u < U; u++) e[u] = ne[u];
      const bool chk = chk_i >= 0;
      const int next_need = (w < P - 1) ? g - RD + 2 : g - RD + 1;
      if (n_left < g + 1 || (chk && (n_cons < chk_k + 1 || n_next < next_need))) {
        wait_ge(left_cnt, g + 1, 0x100u + (unsigned)g);
        if (chk) {
          wait_ge(&cons_cnt[chk_c], chk_k + 1, 0x300u + (unsigned)g);  // slot g % RD converted
          wait_ge(next_cnt, next_need, 0x400u + (unsigned)g);          // ... read by w+1 / published
        }
        asm volatile("" ::: "memory");
        load_left(e, g);
      }
      if (chk) {
        chk_c += P;
        if (chk_c >= NC) {
          chk_c -= NC;
          chk_k++;
        }
      }
      chk_i += P;
      if (g + 1 < G) look_ahead(g + 1);
      if (g == g0w || tin == 0) {
        // ---- period set-up ----
        if (g != g0w && v != 0.0) {  // renormalise: significand back to 2^-PC_BIAS * [0.5,1)
          const int k = __builtin_amdgcn_frexp_exp(v);
          v = ldexp(v, -k - PC_BIAS);
          ep += k + PC_BIAS;
        }
        // freeze the scale of the cross-lane input for the period (bounds: see k_fill_pc)
        int el = ep;
        if (w == 0) {
          if (has_left) el = edge_e[g & (RE - 1)];
        } else {
          el = ebuf[p & 3][64 * w - 1];
          // the row above the first row of a period was produced under the previous exponent
          if (tin == 0 && p >= 1 && lane == 0) e[0] = ldexp(e[0], ebuf[(p - 1) & 3][64 * w - 1] - el);
        }
        int dl = wave_shr1(ep, ep) - ep;
        if (lane == 0) dl = el - ep;
        s = ldexp(1.0, min(max(dl, -1100), 220));
        ebuf[p & 3][col] = ep;
      }
      if (lane == 0) slot_p[g & (RD - 1)][w] = p & 3;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double t0 = wave_shr1(v, e[u]) * s;
        v = fma(coef, v, t0);
        coef += 1.0;
        vbuf[g & (RD - 1)][u][col] = v;
      }
      lds_post(&prod_done[w], g + 1);
      if (++tin == TP) {
        tin = 0;
        p++;
      }
    }
  } else if (wave < P + NC) {
    // ================= consumers =================
    const int ci = wave - P;
    int w = ci % P, t = g0b + ci / P;
    int done = 0;
    double acc = 0.0;  // (DOT) this lane's share of the sum
    if (DOT == 2) {
      // item (t, w) of block j is slice sg = j P + w of the table: its cells are item_ptr[idx] ..
      // item_ptr[idx + 1] with idx = t * nsg + sg.  The range of the NEXT item is read while this
      // one is processed (a wave-uniform load each).
      auto item_range = [&](int tt, int ww, unsigned &b0, unsigned &b1) {
        const unsigned idx = (unsigned)tt * X.nsg + (unsigned)(j * P + ww);
        b0 = X.item_ptr[idx];
        b1 = X.item_ptr[idx + 1];
      };
      unsigned nb = 0, ne = 0;
      if (t < G) item_range(t, w, nb, ne);
      for (; t < G;) {
        const unsigned beg = nb, end = ne;
        int w2 = w + NC % P, t2 = t + NC / P;
        if (w2 >= P) {
          w2 -= P;
          t2++;
        }
        if (t2 < G) item_range(t2, w2, nb, ne);
        if (beg != end && t >= first_trip(w)) {
          // the first 64 entries can come in while we wait for the producer
          unsigned k = beg + lane;
          unsigned pos = (k < end) ? X.ent_pos[k] : 0u, c = (k < end) ? X.ent_cnt[k] : 0u;
          wait_ge(&prod_done[w], t + 1, 0x600u + (unsigned)t);
          const int slot = t & (RD - 1);
          const int pidx = ((TP == 1) ? t : (int)__umulhi((unsigned)t, X.tp_magic)) & 3;  // period of trip t
          for (;;) {
            const int col = 64 * w + (int)(pos & 63u);
            const double val = bfp_log(vbuf[slot][pos >> 6][col], ebuf[pidx][col], lt);
            acc += (c != 0) ? (double)c * val : 0.0;
            if (k - lane + 64 >= end) break;  // (wave-uniform)
            k += 64;
            pos = (k < end) ? X.ent_pos[k] : 0u;
            c = (k < end) ? X.ent_cnt[k] : 0u;
          }
        }
        done++;
        lds_post(&cons_cnt[ci], done);
        w = w2;
        t = t2;
      }
    }
    for (; t < G;) {
      if (t >= first_trip(w)) {
        wait_ge(&prod_done[w], t + 1, 0x600u + (unsigned)t);
        const int ridx = 64 * w + lane;
        const int cc = c0 + ridx;
        const int coff = cc - 2;  // offset in a table row (column 1: the slack before the row)
        const int slot = t & (RD - 1);
        const int myep = ebuf[slot_p[slot][w]][ridx];
        const int r0 = 3 + t * U;
        const unsigned pitch = stb_row_pitch((unsigned)r0, M);
        const bool fast = (unsigned)(r0 + U - 1) <= N && stb_row_pitch((unsigned)(r0 + U - 1), M) == pitch &&
                          !(j == 0 && t == 0 && w == 0);
        const uint64_t rowoff = stb_row_offset((unsigned)r0, M);
        double *rowbase = table + rowoff;
        const unsigned *cntbase = X.cnt + rowoff;
        if (fast) {
          double x[U], z[U], kf[U], r[U], pl[U];
          double2 tt[U];
          unsigned cn[U];
          if (DOT == 1) {
#pragma unroll
            for (int u = 0; u < U; u++) cn[u] = cntbase[(size_t)u * pitch + coff];
          }
#pragma unroll
          for (int u = 0; u < U; u++) x[u] = vbuf[slot][u][ridx];
#pragma unroll
          for (int u = 0; u < U; u++) tt[u] = lt[(__double2hiint(x[u]) >> 13) & 127];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int hi = __double2hiint(x[u]);
            z[u] = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(x[u]));
            kf[u] = (double)((int)((hi >> 20) & 0x7ff) - 1023 + myep);
          }
#pragma unroll
          for (int u = 0; u < U; u++) r[u] = fma(z[u], tt[u].x, -1.0);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], 0.2, -0.25);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], pl[u], 1.0 / 3.0);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], pl[u], -0.5);
#pragma unroll
          for (int u = 0; u < U; u++) pl[u] = fma(r[u], pl[u], 1.0);
#pragma unroll
          for (int u = 0; u < U; u++) {
            const double val = fma(kf[u], 0.693147180559945309417, fma(r[u], pl[u], tt[u].y));
            if (DOT) {
              // (cells outside the table proper -- the row slack -- have count 0 and may hold anything)
              acc += (cn[u] != 0) ? (double)cn[u] * val : 0.0;
            } else {
              rowbase[(size_t)u * pitch + coff] = val;
            }
          }
        } else {
          for (int u = 0; u < U; u++) {
            const int rr = r0 + u;
            if ((unsigned)rr <= N && cc >= 2) {
              const double val = bfp_log(vbuf[slot][u][ridx], myep, lt);
              if (DOT) {
                const unsigned c = cntbase[coff];
                acc += (c != 0) ? (double)c * val : 0.0;
              } else {
                rowbase[coff] = val;
              }
            }
            const unsigned pt = stb_row_pitch((unsigned)rr, M);
            rowbase += pt;
            cntbase += pt;
          }
        }
      }
      done++;
      lds_post(&cons_cnt[ci], done);
      w += NC % P;
      t += NC / P;
      if (w >= P) {
        w -= P;
        t++;
      }
    }
    if (DOT) {
      // fixed-shape tree over the wave: the same bits on every run
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) X.dotp[((uint64_t)d * X.B + j) * NC + ci] = acc;
    }
  } else if (wave == P + NC) {
    // ================= publisher =================
    if (has_right) {
      __builtin_amdgcn_s_setprio(2);
      unsigned long long *ev_out = X.edge_v + ((uint64_t)d * X.B + j) * X.EV;
      unsigned long long *ee_out = X.edge_e + ((uint64_t)d * X.B + j) * X.NP;
      const int t_first = first_trip(P - 1);
      int pp = t_first / TP, ptin = t_first - pp * TP;  // period of trip t, tracked without dividing
      for (int t = t_first; t < G; t++) {
        // (latency matters here, not issue slots: spin without sleeping)
        if (!aborted) {
          const unsigned long long t_begin = wall_clock64();
          while (lds_peek(&prod_done[P - 1]) < t + 1) {
            if (lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin > X.timeout) {
              wait_ge(&prod_done[P - 1], t + 1, 0x700u + (unsigned)t);  // (records the failure)
              break;
            }
          }
        }
        const int slot = t & (RD - 1);
        if (lane < U) {
          unsigned long long b = (unsigned long long)__double_as_longlong(vbuf[slot][lane][OW - 1]);
          if ((b << 1) == 0) b = CH_NEGZERO;
          __hip_atomic_store(ev_out + 3 + t * U + lane, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == U) {
          const long long ex = (long long)ebuf[pp & 3][OW - 1] + (long long)CH_EOFF;
          __hip_atomic_store(ee_out + t, (unsigned long long)ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_post(&pub_done, t + 1);
        if (++ptin == TP) {
          ptin = 0;
          pp++;
        }
      }
    }
  } else {
    // ================= fetchers (waves P+NC+1 ..) =================
    if (has_left) {
      const unsigned long long *ev_in = X.edge_v + ((uint64_t)d * X.B + (j - 1)) * X.EV;
      const unsigned long long *ee_in = X.edge_e + ((uint64_t)d * X.B + (j - 1)) * X.NP;
      unsigned long long t_begin = 0;
      bool timing = false;
      for (int t = g0b; t < G;) {
        // (NF fetcher waves run this same loop out of step: each delivers what its own load found
        // complete beyond what has been delivered already, so the polling period divides by NF)
        t = max(t, lds_peek(&edge_ready));
        if (t >= G) break;
        // trips t .. t+nt-1 may be written: their ring slots were read by the first producer
        int lim = lds_peek(&prod_done[0]) + RE;
        if (lim > G) lim = G;
        if (lim <= t) {
          wait_ge(&prod_done[0], t - RE + 1, 0x800u + (unsigned)t);
          if (aborted) break;
          continue;
        }
        const int nt = min(16, lim - t);
        // 128 rows (the values one row above the rows they feed) and 17 trip exponents, one round trip
        const int row0 = 2 + t * U;
        const int ra = row0 + lane, rb = row0 + 64 + lane;
        const bool need_a = lane < 8 * nt, need_b = 64 + lane < 8 * nt;
        const bool need_e = lane <= nt;
        unsigned long long va = 0, vb = 0, ve = 0;
        if (need_a) va = __hip_atomic_load(ev_in + ra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need_b) vb = __hip_atomic_load(ev_in + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need_e) ve = __hip_atomic_load(ee_in + t - 1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ma = __ballot(!need_a || va != 0);
        const unsigned long long mb = __ballot(!need_b || vb != 0);
        const unsigned long long me = __ballot(!need_e || ve != 0);
        int nr = 0;  // leading trips with all 8 rows, their exponent and the one before it present
        for (; nr < nt; nr++) {
          const unsigned long long rows = (nr < 8) ? (ma >> (8 * nr)) : (mb >> (8 * (nr - 8)));
          if ((rows & 0xffull) != 0xffull || ((me >> nr) & 3ull) != 3ull) break;
        }
        if (nr == 0) {
          if (!timing) {
            timing = true;
            t_begin = wall_clock64();
          }
          const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (err != 0 || lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin > X.timeout) {
            if (lane == 0) {
              __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (err == 0) {
                __hip_atomic_store(X.hdr + 2, (unsigned)(j | (d << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(X.hdr + 1, 0x900u + (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
            lds_post(&edge_ready, 0x7fffffff);  // release the producer: it runs on with stale edges
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        timing = false;
        // exponent of each row's trip (lane q of ve holds trip t-1+q) and of the trip before it
        const int ex = (int)(long long)(ve - CH_EOFF);
        const int ka = lane >> 3, kb = 8 + (lane >> 3);
        const int ea = __shfl(ex, ka + 1), ea1 = __shfl(ex, ka);
        const int eb = __shfl(ex, kb + 1), eb1 = __shfl(ex, kb);
        double xa = __longlong_as_double((long long)va), xb = __longlong_as_double((long long)vb);
        // the first row of a trip comes from the trip before: bring it to this trip's exponent
        if ((lane & 7) == 0) {
          xa = ldexp(xa, ea1 - ea);
          xb = ldexp(xb, eb1 - eb);
        }
        const int cur = lds_peek(&edge_ready);  // trips below it were delivered by another fetcher
        if (ka < nr && t + ka >= cur) edge_in[((t + ka) & (RE - 1)) * U + (lane & 7)] = xa;
        if (kb < nr && t + kb >= cur) edge_in[((t + kb) & (RE - 1)) * U + (lane & 7)] = xb;
        if (lane >= 1 && lane <= nr && t - 1 + lane >= cur) edge_e[(t - 1 + lane) & (RE - 1)] = ex;
        t += nr;
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_max(&edge_ready, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
      }
    }
  }
}

// ---- chain form of the V-table fill (lib/stable.c:451-482) ------------------------
//
// V^n_m needs V^{n-1}_m and V^{n-1}_{m-1}: the same stencil as the S table, with plain doubles (the
// ratios stay O(1): no exponents, no logs).  So a block is only the chain: P producer waves of one
// column per lane that store their row segment straight into the table, a publisher and a fetcher
// exactly as in k_fill_chain (8-byte granules; -0.0 stands for an exact zero).  The cell update is
// the reference's own expression with contraction off, so the table is bit-identical to it.
template <int P>
__global__ __launch_bounds__(64 * (P + 2)) void k_fillv_chain(fill_args A, chain_args X) {
  constexpr int U = CH_U, RD = 16, RE = CH_RE;
  constexpr int OW = 64 * P;
  __shared__ __attribute__((aligned(16))) double xedge[P][RD][U];  // last column of each slice
  __shared__ __attribute__((aligned(16))) double edge_in[RE * U];
  __shared__ int prod_done[P], pub_done, edge_ready, s_abort;
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_ticket = atomicAdd(X.hdr, 1u);
  for (int i = tid; i < RE * U; i += blockDim.x) edge_in[i] = 0.0;
  __syncthreads();
  const int j = (int)(s_ticket / (unsigned)X.D);
  const int d = (int)(s_ticket % (unsigned)X.D);
  if (j >= X.B) return;
  const unsigned N = A.N, M = A.M;
  const int G = X.G;          // trips: rows 2 + U g .. 9 + U g
  const int c0 = 1 + j * OW;  // first column of the block
  auto first_trip = [&](int w) {  // trip in which the diagonal reaches the first column of slice w
    const int c = c0 + 64 * w;
    return (c <= 2) ? 0 : (c - 2) / U;
  };
  const int g0b = first_trip(0);
  const bool has_left = j > 0, has_right = j < X.B - 1;
  if (tid < P) prod_done[tid] = first_trip(tid);
  if (tid == 0) {
    pub_done = first_trip(P - 1);
    edge_ready = has_left ? g0b : 0x7fffffff;
    s_abort = 0;
  }
  __syncthreads();
  double *table = A.tables + (uint64_t)d * A.tstride;
  bool aborted = false;
  auto wait_ge = [&](const int *cnt, int need, unsigned code) {
    if (aborted || lds_peek(cnt) >= need) return;
    if (!chain_wait_slow(cnt, need, &s_abort, X.hdr, X.timeout, code, (unsigned)(j | (d << 16)), wave < P ? 1 : 2))
      aborted = true;
  };

  if (wave < P) {
    // ================= producers =================
    __builtin_amdgcn_s_setprio(3);
    const int w = wave;
    const int g0w = first_trip(w);
    const int c = c0 + 64 * w + lane;  // my column
    const double a = A.a[d];
    const double ca = (double)c * a, cb = (double)(c - 1) * a;
    // row 1 (and every row above the diagonal): V_1 = +inf by convention, everything else 0
    double v = (c == 1) ? HUGE_VAL : 0.0;
    // columns 2..M have a slot; column 1 and the columns past M go to the dump
    const bool ok = c >= 2 && (unsigned)c <= M;
    double *dump = reinterpret_cast<double *>(X.edge_e) + lane;  // (64 words the V fill does not use)
    double *pc = ok ? table + stb_vrow_offset((unsigned)(2 + g0w * U), M) + (c - 2) : dump;
    const int *left_cnt = (w == 0) ? &edge_ready : &prod_done[w - 1];
    const int *next_cnt = (w < P - 1) ? &prod_done[w + 1] : &pub_done;
    int n_left, n_next;
    double ne[U];
    auto load_left = [&](double(&x)[U], int g) {
      if (w == 0) {
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = edge_in[(g & (RE - 1)) * U + u];
      } else {
        x[0] = xedge[w - 1][(g - 1) & (RD - 1)][U - 1];
#pragma unroll
        for (int u = 1; u < U; u++) x[u] = xedge[w - 1][g & (RD - 1)][u - 1];
      }
    };
    auto look_ahead = [&](int g) {
      n_left = lds_peek(left_cnt);
      n_next = lds_peek(next_cnt);
      asm volatile("" ::: "memory");
      load_left(ne, g);
    };
    look_ahead(g0w);
    auto trip = [&](int g, auto partial_tag) {
      constexpr bool partial = decltype(partial_tag)::value;
      double e[U];
#pragma unroll
      for (int u = 0; u < U; u++) e[u] = ne[u];
      const int next_need = (w < P - 1) ? g - RD + 2 : g - RD + 1;
      if (__builtin_expect(n_left < g + 1 || n_next < next_need, 0)) {
        wait_ge(left_cnt, g + 1, 0x100u + (unsigned)g);
        wait_ge(next_cnt, next_need, 0x400u + (unsigned)g);
        asm volatile("" ::: "memory");
        load_left(e, g);
      }
      if (g + 1 < G) look_ahead(g + 1);
      const int r0 = 2 + g * U;
      // (rows 2+8g .. 9+8g have lengths 1+8g .. 8+8g: one pitch per trip, as for the S table)
      const size_t inc = ok ? stb_vrow_pitch((unsigned)r0, M) : 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int n = r0 + u;
        const double left = wave_shr1(v, e[u]);
        // lib/stable.c:475-480 (see v_cell), with this lane's constants hoisted
        double y;
        {
#pragma clang fp contract(off)
          const double nm1 = (double)(n - 1);
          const double num = 1.0 + ((c < n) ? ((nm1 - ca) * v) : 0.0);
          const double den = 1.0 / left + (nm1 - cb);
          y = num / den;
        }
        v = (c == 1) ? HUGE_VAL : (c > n) ? 0.0 : y;
        if (lane == 63) xedge[w][g & (RD - 1)][u] = v;
        if (partial)
          *(((unsigned)n <= N) ? pc : dump) = v;
        else
          *pc = v;
        pc += inc;
      }
      lds_post(&prod_done[w], g + 1);
    };
    const int Gfull = ((int)N >= 1 + U) ? ((int)N - 1) / U : 0;  // trips whose rows all exist
    int g = g0w;
    for (; g < Gfull; g++) trip(g, std::false_type{});
    for (; g < G; g++) trip(g, std::true_type{});
  } else if (wave == P) {
    // ================= publisher =================
    if (has_right) {
      unsigned long long *ev_out = X.edge_v + ((uint64_t)d * X.B + j) * X.EV;
      for (int t = first_trip(P - 1); t < G; t++) {
        wait_ge(&prod_done[P - 1], t + 1, 0x700u + (unsigned)t);
        if (lane < U) {
          unsigned long long b = (unsigned long long)__double_as_longlong(xedge[P - 1][t & (RD - 1)][lane]);
          if ((b << 1) == 0) b = CH_NEGZERO;
          __hip_atomic_store(ev_out + 2 + t * U + lane, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_post(&pub_done, t + 1);
      }
    } else {
      // nobody to publish to: only release the ring slots
      for (int t = first_trip(P - 1); t < G; t++) {
        wait_ge(&prod_done[P - 1], t + 1, 0x700u + (unsigned)t);
        lds_post(&pub_done, t + 1);
      }
    }
  } else {
    // ================= fetcher =================
    if (has_left) {
      const unsigned long long *ev_in = X.edge_v + ((uint64_t)d * X.B + (j - 1)) * X.EV;
      unsigned long long t_begin = 0;
      bool timing = false;
      for (int t = g0b; t < G;) {
        int lim = lds_peek(&prod_done[0]) + RE;
        if (lim > G) lim = G;
        if (lim <= t) {
          wait_ge(&prod_done[0], t - RE + 1, 0x800u + (unsigned)t);
          if (aborted) break;
          continue;
        }
        const int nt = min(16, lim - t);
        const int row0 = 1 + t * U;  // the rows one above the rows of trip t
        const bool need_a = lane < 8 * nt, need_b = 64 + lane < 8 * nt;
        unsigned long long va = 0, vb = 0;
        if (need_a) va = __hip_atomic_load(ev_in + row0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need_b) vb = __hip_atomic_load(ev_in + row0 + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ma = __ballot(!need_a || va != 0);
        const unsigned long long mb = __ballot(!need_b || vb != 0);
        int nr = 0;
        for (; nr < nt; nr++) {
          const unsigned long long rows = (nr < 8) ? (ma >> (8 * nr)) : (mb >> (8 * (nr - 8)));
          if ((rows & 0xffull) != 0xffull) break;
        }
        if (nr == 0) {
          if (!timing) {
            timing = true;
            t_begin = wall_clock64();
          }
          const unsigned err = __hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (err != 0 || lds_peek(&s_abort) || (unsigned long long)wall_clock64() - t_begin > X.timeout) {
            if (lane == 0) {
              __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (err == 0) {
                __hip_atomic_store(X.hdr + 2, (unsigned)(j | (d << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(X.hdr + 1, 0x900u + (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
            lds_post(&edge_ready, 0x7fffffff);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        timing = false;
        const int ka = lane >> 3, kb = 8 + (lane >> 3);
        // (-0.0 is the granule of an exact zero: the cells it feeds are forced to 0 anyway)
        const double xa = (va == CH_NEGZERO) ? 0.0 : __longlong_as_double((long long)va);
        const double xb = (vb == CH_NEGZERO) ? 0.0 : __longlong_as_double((long long)vb);
        if (ka < nr) edge_in[((t + ka) & (RE - 1)) * U + (lane & 7)] = xa;
        if (kb < nr) edge_in[((t + kb) & (RE - 1)) * U + (lane & 7)] = xb;
        t += nr;
        lds_post(&edge_ready, t);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side: geometry, workspace, launches

// geometry of k_fill_chain: column blocks per table, trips, edge stream lengths
struct chain_geom {
  int P, NC, NF, B, G;
  uint64_t EV, NP;
  size_t bytes;  // header + edge streams for D tables
};
static chain_geom chain_geometry(unsigned N, unsigned M, int D) {
  chain_geom g;
  // block shape: up to ~5 tables of 10^4 columns narrower blocks on more compute units win (the
  // fill is a latency chain), with three fetcher waves to shorten the hand-off; for more tables in
  // flight wider blocks with fewer hand-offs do
  const bool narrow = (uint64_t)D * M <= 50000;
  g.P = stb_env_int("STB_CHAIN_P", narrow ? 2 : 4);
  if (g.P != 1 && g.P != 2 && g.P != 4) g.P = narrow ? 2 : 4;
  g.NC = stb_env_int("STB_CHAIN_NC", g.P == 4 ? 10 : g.P == 2 ? 6 : 3);
  if (g.NC < g.P) g.NC = g.P;
  g.NF = stb_env_int("STB_CHAIN_NF", narrow ? 3 : 1);
  if (g.NF < 1) g.NF = 1;
  if (g.NF > 3) g.NF = 3;
  if (g.NF == 2) g.NF = 3;  // (compiled shapes have one or three fetchers)
  if (g.P + g.NC + 1 + g.NF > 16) g.NF = 1;
  if (g.P + g.NC + 1 + g.NF > 16) g.NC = 15 - g.NF - g.P;
  const unsigned cols = (M < N - 1) ? M : N - 1;  // columns 1..min(M, N-1) hold stored cells
  g.B = (int)((cols + 64 * g.P - 1) / (64 * g.P));
  if (g.B < 1) g.B = 1;
  g.G = (N > 2) ? (int)((N - 2 + CH_U - 1) / CH_U) : 0;
  g.EV = (uint64_t)3 + (uint64_t)g.G * CH_U + 136;  // the fetcher reads 128 rows at a time
  g.NP = (uint64_t)g.G + 24;                        // ... and 17 trip exponents
  g.bytes = 256 + (size_t)D * g.B * (g.EV + g.NP) * sizeof(unsigned long long);
  return g;
}

// geometry of k_fillv_chain
struct vchain_geom {
  int P, B, G;
  uint64_t EV;
  size_t bytes;
};
static vchain_geom vchain_geometry(unsigned N, unsigned M, int D) {
  vchain_geom g;
  g.P = stb_env_int("STB_FILLV_P", 4);
  if (g.P != 1 && g.P != 2 && g.P != 4) g.P = 4;
  const unsigned cols = (M < N) ? M : N;  // columns 1..min(M, N) (row n stores m <= n)
  g.G = (int)((N - 1 + CH_U - 1) / CH_U);  // rows 2..N
  g.B = (int)((cols + 64 * g.P - 1) / (64 * g.P));
  if (g.B < 1) g.B = 1;
  g.EV = (uint64_t)2 + (uint64_t)g.G * CH_U + 136;
  g.bytes = 256 + 512 + (size_t)D * g.B * g.EV * sizeof(unsigned long long);
  return g;
}

// bytes of workspace the chain forms need for D tables (S or V)
size_t stb_chain_workspace(unsigned N, unsigned M, int D) {
  if (N < 2 || M < 2 || D < 1) return 0;
  const size_t s = (N >= 3) ? chain_geometry(N, M, D).bytes : 0;
  const size_t v = vchain_geometry(N, M, D).bytes;
  return (s > v ? s : v) + 256;
}

int stb_chain_tuning(unsigned N, unsigned M, int D, int *P_out) {
  const chain_geom g = chain_geometry(N, M, D);
  if (P_out) *P_out = g.P;
  return 0;
}

static unsigned long long chain_timeout_ticks() {
  return (unsigned long long)stb_env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;  // wall_clock64: 100 MHz
}

// S tables.  ws: 256-byte aligned scratch of at least stb_chain_workspace bytes, zeroed here.
int stb_launch_chain(fill_args &A, int D, char *ws, size_t ws_left, const dot_request *dot, unsigned **hdr_out,
                     hipStream_t st) {
  const unsigned N = A.N, M = A.M;
  const chain_geom cg = chain_geometry(N, M, D);
  int Pc = stb_period_rows(N);
  const int Penv = stb_env_int("STB_FILL_P", 0);
  if (Penv > 0 && Penv < Pc) Pc = Penv;
  chain_args X;
  X.TP = Pc / CH_U;
  if (X.TP < 1) return stb_fail("stb_fill_S: renormalisation period %d shorter than a trip", Pc);
  X.tp_magic = (unsigned)((0x100000000ull + (unsigned)X.TP - 1) / (unsigned)X.TP);
  X.G = cg.G;
  X.D = D;
  X.B = cg.B;
  X.EV = cg.EV;
  X.NP = cg.NP;
  if (cg.bytes > ws_left) return stb_fail("stb_fill_S: workspace too small for the chain form");
  X.hdr = (unsigned *)ws;
  X.edge_e = (unsigned long long *)(ws + 256);
  X.edge_v = X.edge_e + (size_t)D * cg.B * X.NP;
  X.timeout = chain_timeout_ticks();
  HIPCHK(hipMemsetAsync(ws, 0, stb_align_up(cg.bytes, 16), st));
  *hdr_out = X.hdr;
  stb_launch_s1(A, D, st);
  const dim3 grid((unsigned)cg.B * (unsigned)D);
  X.cnt = dot ? dot->cnt : nullptr;
  X.item_ptr = dot ? dot->item_ptr : nullptr;
  X.ent_pos = dot ? dot->ent_pos : nullptr;
  X.ent_cnt = dot ? dot->ent_cnt : nullptr;
  X.nsg = dot ? dot->nsg : 0;
  X.dotp = dot ? dot->dotp : nullptr;
  if (dot) const_cast<dot_request *>(dot)->parts_per_table = cg.B * cg.NC;
  const int dk = !dot ? 0 : (dot->item_ptr ? 2 : 1);
#define CHAIN1(PP, NN, FF, DD) STB_LAUNCH((k_fill_chain<PP, NN, FF, DD>), grid, dim3(64 * (PP + NN + 1 + FF)), st, A, X)
#define CHAIN(PP, NN, FF)                 \
  do {                                    \
    if (dk == 2) CHAIN1(PP, NN, FF, 2);   \
    else if (dk == 1) CHAIN1(PP, NN, FF, 1); \
    else CHAIN1(PP, NN, FF, 0);           \
  } while (0)
  const int shape = cg.P * 1000 + cg.NC * 10 + cg.NF;
  // (the summing variants are compiled for the two default block shapes only)
  if (dk != 0 && shape != 2063 && shape != 4101)
    return stb_fail("stb_fill_S: the fused evaluation needs a default block shape (unset STB_CHAIN_P / _NC / _NF)");
  switch (shape) {
    case 2063: CHAIN(2, 6, 3); break;
    case 4101: CHAIN(4, 10, 1); break;
    case 1031: CHAIN1(1, 3, 1, 0); break;
    case 1033: CHAIN1(1, 3, 3, 0); break;
    case 1061: CHAIN1(1, 6, 1, 0); break;
    case 1063: CHAIN1(1, 6, 3, 0); break;
    case 2041: CHAIN1(2, 4, 1, 0); break;
    case 2043: CHAIN1(2, 4, 3, 0); break;
    case 2061: CHAIN1(2, 6, 1, 0); break;
    case 2081: CHAIN1(2, 8, 1, 0); break;
    case 2083: CHAIN1(2, 8, 3, 0); break;
    case 4061: CHAIN1(4, 6, 1, 0); break;
    case 4063: CHAIN1(4, 6, 3, 0); break;
    case 4081: CHAIN1(4, 8, 1, 0); break;
    case 4083: CHAIN1(4, 8, 3, 0); break;
    default:
      return stb_fail("stb_fill_S: no chain kernel for %d producers / %d consumers / %d fetchers", cg.P, cg.NC, cg.NF);
  }
#undef CHAIN
#undef CHAIN1
  HIPCHK(hipGetLastError());
  return 0;
}

// V tables: P producer waves per block, no consumers (nothing to convert)
int stb_launch_vchain(fill_args &A, int D, char *ws, size_t ws_left, unsigned **hdr_out, hipStream_t st) {
  const vchain_geom vg = vchain_geometry(A.N, A.M, D);
  chain_args X;
  memset(&X, 0, sizeof(X));
  X.TP = 1;
  X.G = vg.G;
  X.D = D;
  X.B = vg.B;
  X.EV = vg.EV;
  if (vg.bytes > ws_left) return stb_fail("stb_fill_V: workspace too small for the chain form");
  X.hdr = (unsigned *)ws;
  X.edge_e = (unsigned long long *)(ws + 256);  // (here: the dump for columns without a slot)
  X.edge_v = (unsigned long long *)(ws + 256 + 512);
  X.timeout = chain_timeout_ticks();
  HIPCHK(hipMemsetAsync(ws, 0, stb_align_up(vg.bytes, 16), st));
  *hdr_out = X.hdr;
  const dim3 grid((unsigned)vg.B * (unsigned)D);
  if (vg.P == 1) STB_LAUNCH((k_fillv_chain<1>), grid, dim3(64 * 3), st, A, X);
  else if (vg.P == 2) STB_LAUNCH((k_fillv_chain<2>), grid, dim3(64 * 4), st, A, X);
  else STB_LAUNCH((k_fillv_chain<4>), grid, dim3(64 * 6), st, A, X);
  HIPCHK(hipGetLastError());
  return 0;
}

// fill_chain.hip -- the chain forms: ONE launch per fill, column strips keep their columns for all
// rows and hand their right edge to the next strip through 8-byte granules in HBM.
//
//   k_fill_chain   S table of S_remake_part's double-S branch (reference lib/stable.c:321-388); as a
//                  DOT kernel it sums count * log S for aterms (lib/samplea.c:68-80) without storing
//                  a table
//   k_fillv_chain  V table (lib/stable.c:451-482), bit-identical to the reference
//
// k_fill_chain.  A workgroup owns a strip of W = 64*C*P columns for ALL rows of one table:
//   * P producer waves carry the recurrence S^n_m = (n-1-m a) S^{n-1}_m + S^{n-1}_{m-1} in the linear
//     domain, C adjacent columns per lane sharing one exponent (block-floating cells, frozen for a
//     period of rows, renormalised at its end); the left neighbour inside the wave comes through
//     one DPP wave shift per row, the one across producer waves through the LDS ring, one trip
//     behind.  Per row: 2 DPP moves, 1 multiply, C fma, C adds and the store of the C raw
//     significands into an LDS ring of RD trips (a trip = 8 rows).
//   * NC = MG*C*P consumer waves turn (trip, 64-column slice) items into logs -- exponent field + 7
//     mantissa bits index a 128-entry table, degree-5 polynomial -- stage-major over the 8 rows, and
//     store them: 512 contiguous bytes per row and wave.  A consumer hands its ring slot back as soon
//     as the 8 x 64 significands are in its registers.
//   * the consumer of the strip's LAST slice also publishes the strip's right edge: the raw double
//     per row (8-byte granules, -0.0 for an exact zero, so that 0 means "not written yet") and the
//     lane exponent per trip, write-through stores; a granule is its own flag.
//   * NF fetcher waves poll the left neighbour strip's granules (128 rows per round trip,
//     L1-bypassing loads) and deliver complete trips into an LDS ring for the first producer.
// Waves meet through counters in LDS only (LDS executes a wave's instructions in order); a producer
// reads everything a trip needs -- counters and its 8 left inputs -- one trip ahead, under the
// previous trip's arithmetic, and takes one unlikely branch when a counter was short.
//
// Strips take (j, d) from an atomic ticket, j-major: a strip waits only for a strip with a smaller
// ticket, i.e. one that is running or done, so progress does not depend on dispatch order or
// residency.  Every wait is bounded; on expiry the strip records an error and runs to its end
// (stb_fill_status, which then repeats the fill with k_fill_pc).
//
// DOT = 1 / 2: the logs are not stored but multiplied by the cell's occurrence count among the
// (n,t) pairs and summed -- aterms' table part (lib/samplea.c:68-80) without a table in memory.

#include <type_traits>

#include "fill_chain.h"

#define ST_U CH_U    // rows per trip
#define ST_RE CH_RE  // trips in the edge ring
#ifndef ST_LOOK
#define ST_LOOK -1   // the row of a trip after which the next trip's counters and left inputs are read (-1: before row 0)
#endif
#ifndef ST_FK
#define ST_FK 2      // polls a fetcher keeps in flight
#endif

// which wave does what: waves 0 .. P-1 produce; fetchers take the first later waves that share a
// SIMD with a producer (waves go to SIMDs round-robin, and fetchers mostly sleep); the rest convert
template <int WT, int P, int NF>
struct chain_roles {
  int role[16];  // 0 producer, 1 consumer, 2 fetcher
  int idx[16];   // ordinal among the waves of the same role
  constexpr chain_roles() : role{}, idx{} {
    for (int w = 0; w < 16; w++) {
      role[w] = 1;
      idx[w] = 0;
    }
    for (int w = 0; w < P; w++) {
      role[w] = 0;
      idx[w] = w;
    }
    int nf = 0;
    for (int w = P; w < WT && nf < NF; w++)
      if ((w & 3) < P) {
        role[w] = 2;
        idx[w] = nf++;
      }
    for (int w = WT - 1; w >= P && nf < NF; w--)
      if (role[w] == 1) {
        role[w] = 2;
        idx[w] = nf++;
      }
    int nc = 0;
    for (int w = P; w < WT; w++)
      if (role[w] == 1) idx[w] = nc++;
  }
};

// waves per SIMD the register allocation must leave room for: as many workgroups per compute unit
// (up to four) as fit in LDS while a wave keeps at least 80 registers.  A workgroup's waves go to the
// four SIMDs round-robin, ceil(WT/4) on the fullest, and the dispatcher only places a workgroup
// whose waves all fit (placement census, tools/census_chain.py: with 96 registers a second
// 10-wave workgroup was never co-resident; the strips that do not fit on the chip at once wait for a
// free unit, and those are the short strips at the right end of the tables, which then start when the
// long ones end).
constexpr int chain_min_waves(int C, int P, int MG, int NF, int RD) {
  const int WT = P + MG * C * P + NF;
  const int lds = RD * ST_U * 64 * C * P * 8 + 8192;
#ifdef STB_NO_MINW
  return 1;
#endif
#ifdef STB_STAMPS
  // (the diagnostic build carries clocks and counters in registers: with the production budget it
  // spills and its timeline shows consumers twice as slow as they are; it runs few tables anyway)
  const int nb_max = 2;
#else
  const int nb_max = 4;
#endif
  for (int nb = nb_max; nb >= 2; nb--)
    if (nb * lds <= 160 * 1024 && nb * ((WT + 3) / 4) <= 6) return nb * ((WT + 3) / 4);
  return 1;
}

template <int C, int P, int MG, int NF, int DOT, int RD>
__device__ __forceinline__ void chain_body(const fill_args &A, const chain_args &X) {
  constexpr int U = ST_U, RE = ST_RE;
  constexpr int S = C * P;      // 64-column slices per strip
  constexpr int NC = MG * S;    // consumer waves
  constexpr int WP = 64 * C;    // columns of a producer wave
  constexpr int W = WP * P;     // columns of a strip
  constexpr int WT = P + NC + NF;
  static_assert(C == 1 || C == 2 || C == 4, "columns per lane");
  static_assert(S <= 4 && WT <= 16 && NF >= 1 && MG >= 1 && (RD & (RD - 1)) == 0 && RD >= 2, "block shape");
  __shared__ double2 lt[128];
  // (the ring is dynamic LDS -- RD*U*W*8 bytes at launch -- so that the compiler does not count it: it
  // would relax `waves per SIMD` to the occupancy LDS allows with a workgroup's waves spread evenly over
  // the SIMDs, 5 for two 10-wave workgroups, one short of what the fullest SIMD must hold)
  extern __shared__ __attribute__((aligned(16))) double chain_ring[];
  double(*vbuf)[U][W] = reinterpret_cast<double(*)[U][W]>(chain_ring);
  __shared__ int ebuf[8][64 * P];  // lane exponent per period (ring of 8 periods), per producer lane
  __shared__ __attribute__((aligned(16))) double edge_in[RE * U];
  __shared__ int edge_e[RE];
  __shared__ __attribute__((aligned(16))) unsigned long long cons_cnt64[MG][2];
  int(*cons_cnt)[4] = reinterpret_cast<int(*)[4]>(cons_cnt64);  // items done, per consumer [group][slice]
  __shared__ int prod_done[P], edge_ready, s_abort, s_awake;
  __shared__ int post_pad[64];  // where lanes 1..63 of a producer's progress post go (see the producers)
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_ticket = atomicAdd(X.hdr, 1u);
  if (tid < 128) lt[tid] = A.lt[tid];
  for (int i = tid; i < RE * U; i += blockDim.x) edge_in[i] = 0.0;
  __syncthreads();
  const int j = (int)(s_ticket / (unsigned)X.D);
  const int d = (int)(s_ticket % (unsigned)X.D);
  if (j >= X.B) return;  // (never: the grid is exactly B*D blocks)

  const unsigned N = A.N, M = A.M;
  const int TP = X.TP, G = X.G;
  const int c0 = 1 + j * W;  // first column of the strip
  auto first_trip = [&](int sl) {  // trip in which the diagonal reaches the first column of slice sl
    const int c = c0 + 64 * sl;
    return (c <= 3) ? 0 : (c - 3) / U;
  };
  const int g0b = first_trip(0);
  const bool has_left = j > 0, has_right = j < X.B - 1;
  double *table = A.tables + (uint64_t)d * A.tstride;
  if (tid < MG * 4) (&cons_cnt[0][0])[tid] = 0;
  if (tid < P) prod_done[tid] = first_trip(C * tid);
  if (tid == 0) {
    edge_ready = has_left ? g0b : 0x7fffffff;
    s_abort = 0;
    s_awake = has_left ? 0 : 1;
  }
  __syncthreads();

  constexpr chain_roles<WT, P, NF> ROLES{};
  int role = 1, ridx = 0;
#pragma unroll
  for (int w = 0; w < WT; w++)
    if (wave == w) {
      role = ROLES.role[w];
      ridx = ROLES.idx[w];
    }
  const unsigned who = (unsigned)(j | (d << 16));
  bool aborted = false;
  auto wait_ge = [&](const int *cnt, int need, unsigned code, int nap) {
    // (what the counter guards is read -- or written -- after it: the compiler may not move LDS accesses across)
    if (aborted || lds_peek(cnt) >= need) {
      asm volatile("" ::: "memory");
      return;
    }
    if (!chain_wait_slow(cnt, need, &s_abort, X.hdr, X.timeout, code, who, nap)) aborted = true;
    asm volatile("" ::: "memory");
  };
  auto period_of = [&](int t) { return (TP == 1) ? t : (int)__umulhi((unsigned)t, X.tp_magic); };
  // A strip has nothing to do until the diagonal reaches it, and a workgroup that spins meanwhile slows
  // the one it shares a compute unit with (8 tables: 1.55 ms against 1.35 with the last strips not even
  // resident).  So everybody dozes -- one LDS look per ~1000 cycles -- until the fetcher has seen the
  // left strip publish a trip shortly before the first one needed here.
  auto doze = [&]() {
    while (!lds_peek(&s_awake) && !lds_peek(&s_abort)) __builtin_amdgcn_s_sleep(16);
  };

  if (role == 0) {
    // ================= producers =================
    // A lone wave issues an instruction every ~5 cycles (fp64: ~9) whatever its kind, so this loop is
    // counted in instructions: what a trip needs from other waves -- three kinds of counters and its
    // 8 left inputs -- is read while the previous trip is computed, and the rare cases sit behind one
    // unlikely branch each.
    doze();
    __builtin_amdgcn_s_setprio(3);
    // (one copy of the loop per producer wave of the block, each with its own index w folded in)
    auto produce = [&](auto wc) {
      constexpr int w = decltype(wc)::value;
#ifdef STB_STAMPS
      // placement census: where the producer of every strip ran (compute unit, SIMD, XCD) and when
      unsigned long long *census = nullptr;
      unsigned long long census_c0 = 0;
      if (X.dbg && w == 0 && s_ticket < 1024u) {
        census = X.dbg + (size_t)160 * 1280 * 4 + 160 * 8 * 4 + 160 * 16 * 4 + (size_t)s_ticket * 4;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0) {
          census[0] = (unsigned long long)j | ((unsigned long long)d << 16) | (1ull << 40);
          census[1] = (unsigned long long)hw | ((unsigned long long)(xcc & 15u) << 32);
          census[2] = wall_clock64();
          census_c0 = clock64();
        }
      }
#endif
      const int g0w = first_trip(C * w);
      const double a = A.a[d];
      const int colw = WP * w + lane * C;  // first of my C columns inside the strip
      const int cl = c0 + colw;
      double v[C], coef[C];
#pragma unroll
      for (int i = 0; i < C; i++) {
        const int c = cl + i;
        // row 2 of the table: S^2_1 = 1 - a, S^2_2 = 1; everything else starts above the diagonal
        v[i] = (c == 1) ? ldexp(1.0 - a, -1 - PC_BIAS) : (c == 2) ? ldexp(1.0, -1 - PC_BIAS) : 0.0;
        coef[i] = (double)(2 + g0w * U) - (double)c * a;  // n - 1 - c a for the first row of trip g0w
      }
      double s = 1.0;
      int ep = 1 + PC_BIAS;
      int p = g0w / TP, tin = g0w - p * TP;
      // the consumers that last read the ring slot a trip overwrites: trip g - RD, i.e. item kp = g - RD - g0b
      // of the strip, which group q = kp % MG took as its (kp / MG)-th
      int kp = g0w - RD - g0b, q = 0, qi = 0;
      if (kp > 0) {
        q = kp % MG;
        qi = kp / MG;
      }
      const int *left_cnt = (w == 0) ? &edge_ready : &prod_done[w > 0 ? w - 1 : 0];
      const int *next_cnt = &prod_done[(w < P - 1) ? w + 1 : w];
      int n_left, n_next = 0;
      int nc[C];
#ifdef STB_STAMPS
      unsigned long long st_relook = 0, st_wait = 0, st_left = 0, st_ticks = 0;
#endif
      // the progress post as ONE store of the whole wave: lane 0 writes the counter, the other lanes a
      // scratch word each (no exec masking: two scalar instructions and a branch less per trip)
      int *post_addr = (lane == 0) ? &prod_done[w] : &post_pad[lane];
      auto peek_counters = [&](int qq) {
        n_left = lds_peek(left_cnt);
        if (w < P - 1) n_next = lds_peek(next_cnt);
        // (the C counters of a group are adjacent: 8-byte LDS reads)
        if (C >= 2) {
#pragma unroll
          for (int i = 0; i < C; i += 2) {
            const unsigned long long c2 =
                __hip_atomic_load(&cons_cnt64[qq][(C * w + i) / 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nc[i] = (int)(unsigned)c2;
            nc[(i + 1) % C] = (int)(unsigned)(c2 >> 32);
          }
        } else {
          nc[0] = lds_peek(&cons_cnt[qq][w]);
        }
        asm volatile("" ::: "memory");
      };
      auto load_left = [&](double(&x)[U], int g) {
        if (w == 0) {
          // (separate 8-byte reads: each lands in the register pair the row's DPP shift then overwrites,
          // which a 16-byte read's register tuple does not allow without two copies per row)
          const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&edge_in[(g & (RE - 1)) * U]);
#pragma unroll
          for (int u = 0; u < U; u++)
            x[u] = __longlong_as_double((long long)__hip_atomic_load(src + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        } else {
          // the last column of the producer to my left: row U-1 of trip g-1, rows 0..U-2 of trip g
          const int ecol = (w > 0) ? WP * w - 1 : 0;
          const double *b1 = &vbuf[g & (RD - 1)][0][ecol];
          x[0] = vbuf[(g - 1) & (RD - 1)][U - 1][ecol];
#pragma unroll
          for (int u = 1; u < U; u++) x[u] = b1[(u - 1) * W];
        }
      };
      auto counters_ok = [&](int g) {
        bool ok = n_left >= g + 1;
        if (kp >= 0) {
#pragma unroll
          for (int i = 0; i < C; i++) ok = ok && nc[i] >= qi + 1;
          // ... and the next producer has read its left inputs from it (rows of trip g - RD feed its
          // trips g - RD and g - RD + 1)
          if (w < P - 1) ok = ok && n_next >= g - RD + 2;
        }
        return ok;
      };
      // one trip: `e` holds its left inputs (read speculatively during the previous trip), `en`
      // receives the next trip's
      auto trip = [&](int g, double(&e)[U], double(&en)[U]) {
        if (__builtin_expect(!counters_ok(g), 0)) {
          // the counters were read a trip ago: look again (one LDS round trip) before settling down to wait
#ifdef STB_STAMPS
          const unsigned long long tw0 = wall_clock64();
          st_relook++;
#endif
          peek_counters(q);
          load_left(e, g);
          if (!counters_ok(g)) {
#ifdef STB_STAMPS
            st_wait++;
            if (n_left < g + 1) st_left++;
#endif
            wait_ge(left_cnt, g + 1, 0x100u, 1);
            if (kp >= 0) {
#pragma unroll
              for (int i = 0; i < C; i++) wait_ge(&cons_cnt[q][C * w + i], qi + 1, 0x300u, 1);  // slot g % RD converted
              if (w < P - 1) wait_ge(next_cnt, g - RD + 2, 0x400u, 1);
            }
            asm volatile("" ::: "memory");
            load_left(e, g);
          }
#ifdef STB_STAMPS
          st_ticks += wall_clock64() - tw0;
#endif
        }
#ifdef STB_STAMPS
        if (X.dbg && d == 0 && lane == 0 && w == 0 && j < 160 && g < 1280) X.dbg[((size_t)j * 1280 + g) * 4 + 3] = wall_clock64();
#endif
        // the trip after this one: its slot was last read by item kp + 1
        kp++;
        if (kp > 0 && ++q == MG) {
          q = 0;
          qi++;
        }
        // (unconditionally: past the last trip this reads ring slots that exist and uses nothing, and a
        // load under a condition costs two register copies per row in the trip that consumes it)
        auto look_ahead = [&]() {
          peek_counters(q);
          load_left(en, g + 1);
        };
        if (ST_LOOK < 0) look_ahead();
        if (__builtin_expect(g == g0w || tin == 0, 0)) {
          // ---- period set-up ----
          if (g != g0w) {  // renormalise: the lane's largest significand back to 2^-PC_BIAS * [0.5,1)
            int kmax = -4000;
#pragma unroll
            for (int i = 0; i < C; i++)
              if (v[i] != 0.0) kmax = max(kmax, __builtin_amdgcn_frexp_exp(v[i]));
            if (kmax > -4000) {
#pragma unroll
              for (int i = 0; i < C; i++) v[i] = ldexp(v[i], -kmax - PC_BIAS);
              ep += kmax + PC_BIAS;
            }
          }
          // freeze the scale of the cross-lane input for the period (bounds: see k_fill_pc)
          int el = ep;
          if (w == 0) {
            if (has_left) el = edge_e[g & (RE - 1)];
          } else {
            const int elane = (w > 0) ? 64 * w - 1 : 0;  // lane 63 of the producer to my left
            el = ebuf[p & 7][elane];
            // the row above the first row of a period was produced under the previous exponent
            if (tin == 0 && p >= 1 && lane == 0) e[0] = ldexp(e[0], ebuf[(p - 1) & 7][elane] - el);
          }
          int dl = wave_shr1(ep, ep) - ep;
          if (lane == 0) dl = el - ep;
          s = ldexp(1.0, min(max(dl, -1100), 220));
          ebuf[p & 7][64 * w + lane] = ep;
        }
        double *slot = &vbuf[g & (RD - 1)][0][colw];
        // The row stores are written out as ds_write2_b64: left to itself the compiler joins a lane's
        // columns into ds_write_b128, whose source must be four adjacent registers, and pays for that
        // with two to four register copies a row -- in a loop that is counted in instructions.  The
        // two 8-bit offsets of ds_write2 (units of 8 bytes) reach two rows when the strip is 128
        // columns wide, one row otherwise: one address register per `RB` rows.
        constexpr int RB = (W + C - 1 <= 255) ? 2 : 1;
        unsigned sbase[U / RB];
        if (C >= 2) {
          const unsigned sb = lds_addr_of(slot);
#pragma unroll
          for (int k = 0; k < U / RB; k++) sbase[k] = sb + (unsigned)(k * RB * W * 8);
        }
        auto row = [&](auto uc) {
          constexpr int u = decltype(uc)::value;
          const double t0 = wave_shr1(v[C - 1], e[u]) * s;
#pragma unroll
          for (int i = C - 1; i >= 1; i--) v[i] = fma(coef[i], v[i], v[i - 1]);
          v[0] = fma(coef[0], v[0], t0);
#pragma unroll
          for (int i = 0; i < C; i++) coef[i] += 1.0;
#ifdef STB_STAMPS
          if (X.mode == 3) return;
#endif
          if constexpr (C >= 2) {
            constexpr int off = (u % RB) * W;
            lds_store2<off>(sbase[u / RB], v[0], v[1]);
            if constexpr (C == 4) lds_store2<off + 2>(sbase[u / RB], v[2], v[3]);
          } else {
            slot[u * W] = v[0];
          }
        };
        static_assert(U == 8, "rows of a trip");
        row(std::integral_constant<int, 0>{});
        if (ST_LOOK == 0) look_ahead();
        row(std::integral_constant<int, 1>{});
        if (ST_LOOK == 1) look_ahead();
        row(std::integral_constant<int, 2>{});
        if (ST_LOOK == 2) look_ahead();
        row(std::integral_constant<int, 3>{});
        if (ST_LOOK == 3) look_ahead();
        row(std::integral_constant<int, 4>{});
        if (ST_LOOK == 4) look_ahead();
        row(std::integral_constant<int, 5>{});
        if (ST_LOOK == 5) look_ahead();
        row(std::integral_constant<int, 6>{});
        if (ST_LOOK == 6) look_ahead();
        row(std::integral_constant<int, 7>{});
        if (ST_LOOK == 7) look_ahead();
#ifdef STB_STAMPS
        if (X.dbg && d == 0 && lane == 0 && w == P - 1 && j < 160 && g < 1280) X.dbg[((size_t)j * 1280 + g) * 4 + 0] = wall_clock64();
#endif
        asm volatile("" ::: "memory");
        __hip_atomic_store(post_addr, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        if (++tin == TP) {
          tin = 0;
          p++;
        }
      };
      double ea[U], eb[U];
      peek_counters(q);
      load_left(ea, g0w);
      int g = g0w;
      for (; g + 1 < G; g += 2) {
        trip(g, ea, eb);
        trip(g + 1, eb, ea);
      }
      if (g < G) trip(g, ea, eb);
#ifdef STB_STAMPS
      if (census && lane == 0) {
        census[3] = wall_clock64();
        census[1] |= ((unsigned long long)(clock64() - census_c0) >> 10) << 36;  // shader cycles / 1024 of the producer's life
      }
      if (X.dbg && d == 0 && lane == 0 && j < 160 && w < 8) {  // looked again, waited, waited for the left input, ticks
        unsigned long long *o = X.dbg + (size_t)160 * 1280 * 4 + ((size_t)j * 8 + w) * 4;
        o[0] = st_relook;
        o[1] = st_wait;
        o[2] = st_left;
        o[3] = st_ticks;
      }
#endif
    };
    if (P == 1 || ridx == 0) produce(std::integral_constant<int, 0>{});
    else if (ridx == 1) produce(std::integral_constant<int, (P >= 2) ? 1 : 0>{});
    else if (ridx == 2) produce(std::integral_constant<int, (P >= 4) ? 2 : 0>{});
    else produce(std::integral_constant<int, (P >= 4) ? 3 : 0>{});
  } else if (role == 1) {
    // ================= consumers =================
    doze();
    const int ci = ridx;
    const int sl = ci % S, qg = ci / S;
    const int wv = sl / C;           // the producer wave that owns the slice
    const int col = 64 * sl + lane;  // my column inside the strip
    const int cc = c0 + col;
    const int coff = cc - 2;         // offset in a table row (column 1: the slack before the row)
    const unsigned coff8 = (unsigned)(coff + 1) * 8u;
    int one_hi = 0x3ff00000;
    asm volatile("" : "+v"(one_hi));  // (the bit pattern of 1.0's high word, kept in a vector register for v_bfi_b32)
    const int ft = first_trip(sl);
    const bool publisher = (sl == S - 1) && has_right;
    unsigned long long *ev_out = X.edge_v + ((uint64_t)d * X.B + j) * X.EV;
    unsigned long long *ee_out = X.edge_e + ((uint64_t)d * X.B + j) * X.NP;
    double acc = 0.0;  // (DOT) this lane's share of the sum
    int k = 0;
#ifdef STB_STAMPS
    unsigned long long c_wait = 0, c_items = 0;
    const unsigned long long c_begin = wall_clock64();
#endif
    // (DOT = 2) the cells of item (t, sl) are item_ptr[idx] .. item_ptr[idx + 1], idx = t * nsg + slice of the table
    unsigned nb = 0, ne = 0;
    auto item_range = [&](int tt, unsigned &b0, unsigned &b1) {
      const unsigned idx = (unsigned)tt * X.nsg + (unsigned)(j * S + sl);
      b0 = X.item_ptr[idx];
      b1 = X.item_ptr[idx + 1];
    };
    if (DOT == 2 && g0b + qg < G) item_range(g0b + qg, nb, ne);
    for (int t = g0b + qg; t < G; t += MG, k++) {
      unsigned beg = 0, end = 0;
      if (DOT == 2) {
        beg = nb;
        end = ne;
        if (t + MG < G) item_range(t + MG, nb, ne);
      }
#ifdef STB_STAMPS
      const bool live = t >= ft && (DOT != 2 || beg != end) && X.mode != 1;
#else
      const bool live = t >= ft && (DOT != 2 || beg != end);
#endif
      if (live || (publisher && t >= ft)) {
        unsigned pos = 0, cnt2 = 0;
        if (DOT == 2 && live) {  // the first 64 entries can come in while we wait for the producer
          const unsigned kk = beg + lane;
          pos = (kk < end) ? X.ent_pos[kk] : 0u;
          cnt2 = (kk < end) ? X.ent_cnt[kk] : 0u;
        }
#ifdef STB_STAMPS
        const unsigned long long tc0 = wall_clock64();
#endif
        wait_ge(&prod_done[wv], t + 1, 0x600u, 1);
#ifdef STB_STAMPS
        const unsigned long long tc1 = wall_clock64();
        c_wait += tc1 - tc0;
        c_items++;
#endif
        const int slot = t & (RD - 1);
        const int pidx = period_of(t) & 7;
        if (publisher) {
          // the strip's last column, rows of trip t (lanes 0..7), and its lane exponent (lane 8)
          if (lane <= U) {
            unsigned long long b;
            unsigned long long *dst;
            if (lane < U) {
              b = (unsigned long long)__double_as_longlong(vbuf[slot][lane][W - 1]);
              if ((b << 1) == 0) b = CH_NEGZERO;
              dst = ev_out + 3 + t * U + lane;
            } else {
              b = (unsigned long long)((long long)ebuf[pidx][64 * P - 1] + (long long)CH_EOFF);
              dst = ee_out + t;
            }
            __hip_atomic_store(dst, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#ifdef STB_STAMPS
          if (X.dbg && d == 0 && lane == 0 && j < 160 && t < 1280) X.dbg[((size_t)j * 1280 + t) * 4 + 1] = wall_clock64();
#endif
        }
        if (live && DOT == 2) {
          unsigned kk = beg + lane;
          for (;;) {
            const int cx = 64 * sl + (int)(pos & 63u);
            const double val = bfp_log(vbuf[slot][pos >> 6][cx], ebuf[pidx][cx / C], lt);
            acc += (cnt2 != 0) ? (double)cnt2 * val : 0.0;
            if (kk - lane + 64 >= end) break;  // (wave-uniform)
            kk += 64;
            pos = (kk < end) ? X.ent_pos[kk] : 0u;
            cnt2 = (kk < end) ? X.ent_cnt[kk] : 0u;
          }
          lds_post(&cons_cnt[qg][sl], k + 1);
        } else if (live) {
          const int myep = ebuf[pidx][col / C];
          const int r0 = 3 + t * U;
          const unsigned pitch = stb_row_pitch((unsigned)r0, M);
          const bool fast = (unsigned)(r0 + U - 1) <= N && stb_row_pitch((unsigned)(r0 + U - 1), M) == pitch &&
                            !(j == 0 && t == 0 && sl == 0);
          const uint64_t rowoff = stb_row_offset((unsigned)r0, M);
          double *rowbase = table + rowoff;
          const unsigned *cntbase = X.cnt + rowoff;
          double x[U];
#pragma unroll
          for (int u = 0; u < U; u++) x[u] = vbuf[slot][u][col];
          unsigned cn[U];
          if (DOT == 1 && fast) {
#pragma unroll
            for (int u = 0; u < U; u++) cn[u] = cntbase[(size_t)u * pitch + coff];
          }
          // the significands are in registers: the ring slot may be overwritten
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          lds_post(&cons_cnt[qg][sl], k + 1);
          if (fast) {
            double z[U], kf[U], r[U], pl[U];
            double2 tt[U];
#ifdef STB_STAMPS
            if (X.mode == 2) {
#pragma unroll
              for (int u = 0; u < U; u++) tt[u] = make_double2(0.75 + 1e-3 * u, 0.3 + 1e-3 * lane);
            } else
#endif
            {
#pragma unroll
              for (int u = 0; u < U; u++) tt[u] = lt[(__double2hiint(x[u]) >> 13) & 127];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
              const int hi = __double2hiint(x[u]);
              z[u] = __hiloint2double(mantissa_of_one(hi, one_hi), __double2loint(x[u]));
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
                // (row address from scalar registers, my column as a byte offset: no vector address
                // arithmetic; coff >= -1, hence the base one element down.  With several producer
                // waves the compiler does not see that the row address is the same in every lane.)
                if constexpr (P == 1) store_sbase(rowbase - 1 + (size_t)u * pitch, coff8, val);
                else rowbase[(size_t)u * pitch + coff] = val;
              }
            }
          } else {
#pragma unroll 1
            for (int u = 0; u < U; u++) {
              const int rr = r0 + u;
              if ((unsigned)rr <= N && cc >= 2) {
                const double val = bfp_log(x[u], myep, lt);
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
        } else {
          lds_post(&cons_cnt[qg][sl], k + 1);  // (only published)
        }
      } else {
        lds_post(&cons_cnt[qg][sl], k + 1);  // nothing to do for this item
      }
    }
#ifdef STB_STAMPS
    if (X.dbg && d == 0 && lane == 0 && j < 160) {  // per consumer wave: items, ticks waited for the producer, ticks in all
      unsigned long long *o = X.dbg + (size_t)160 * 1280 * 4 + (size_t)160 * 8 * 4 + ((size_t)j * 16 + ci) * 4;
      o[0] = c_items;
      o[1] = c_wait;
      o[2] = wall_clock64() - c_begin;
      o[3] = 1;
    }
#endif
    if (DOT) {
      // fixed-shape tree over the wave: the same bits on every run
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) X.dotp[((uint64_t)d * X.B + j) * NC + ci] = acc;
    }
  } else if (role == 2) {
    // ================= fetchers =================
    if (has_left) {
      const unsigned long long *ev_in = X.edge_v + ((uint64_t)d * X.B + (j - 1)) * X.EV;
      const unsigned long long *ee_in = X.edge_e + ((uint64_t)d * X.B + (j - 1)) * X.NP;
      // A poll reads a window of 8 trips from the left strip's edge -- per lane one row's value (the
      // row above the one it feeds), its trip's exponent and that of the trip before -- and delivers
      // the leading trips it finds complete.  Store to visible load takes 0.4-0.5 us, the length of a
      // producer's trip: FK polls are issued a fraction of that apart and then settled in order, so a
      // poll may start below what an earlier one -- or another fetcher wave -- has delivered since; it
      // delivers what lies beyond.  Everything a round needs from LDS is asked for a nap ahead, and
      // nothing is exchanged between lanes: the wave shares its SIMD with the producer.
      struct edge_poll {
        unsigned long long va, e1, e0;
        int tb, nt;
      };
      const int ka = lane >> 3;
      int t = g0b;  // trips below it are delivered
      {
        // the left strip publishes from its trip g0b - 8 on (the first trip of its last 64 columns): wait,
        // dozing, for the exponent granule of trip g0b - 6, then wake the workgroup
        const unsigned long long *probe = ee_in + max(g0b - 6, 0);
        unsigned long long t_begin = 0;
        unsigned spins = 0;
        while (__hip_atomic_load(probe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(16);
          if ((++spins & 255u) != 0 && X.timeout != 0) continue;
          if (t_begin == 0) t_begin = wall_clock64();
          if (__hip_atomic_load(X.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || lds_peek(&s_abort) ||
              (unsigned long long)wall_clock64() - t_begin >= X.timeout)
            break;  // (the main loop below gives up properly)
        }
        lds_post(&s_awake, 1);
      }
      int pd = lds_peek(&prod_done[0]);
      auto issue = [&](edge_poll &p) {
        // trips tb .. tb+nt-1 may be written: their ring slots were read by the producer
        int lim = pd + RE - 1;
        if (lim > G) lim = G;
        p.tb = t;
        p.nt = max(0, min(8, lim - t));
        // (every lane loads, t < G, and the arrays have room for the window past the last trip: a load
        // under a condition would turn the wait for THIS poll's data into a wait for all loads in flight)
        p.va = __hip_atomic_load(ev_in + 2 + t * U + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.e1 = __hip_atomic_load(ee_in + t + ka, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.e0 = __hip_atomic_load(ee_in + t + ka - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto settle = [&](const edge_poll &p) {
        // leading trips with all 8 rows, their exponent and the one before it present
        const unsigned long long miss = ~__ballot(p.va != 0 && p.e1 != 0 && p.e0 != 0);
        const int nr = min(p.nt, miss ? (int)(__builtin_ctzll(miss) >> 3) : 8);
        int cur = t;
        if (NF > 1) cur = max(cur, lds_peek(&edge_ready));
        t = cur;
        if (p.tb + nr <= cur) return false;
        const int ex = (int)(long long)(p.e1 - CH_EOFF);
        double xa = __longlong_as_double((long long)p.va);
        if (ka < nr && p.tb + ka >= cur) {
          if ((lane & 7) == 0) {
            // the first row of a trip comes from the trip before: bring it to this trip's exponent
            xa = ldexp(xa, (int)(long long)(p.e0 - CH_EOFF) - ex);
            edge_e[(p.tb + ka) & (RE - 1)] = ex;
          }
          edge_in[((p.tb + ka) & (RE - 1)) * U + (lane & 7)] = xa;
#ifdef STB_STAMPS
          if (X.dbg && d == 0 && (lane & 7) == 0 && j < 160 && p.tb + ka < 1280)
            X.dbg[((size_t)j * 1280 + p.tb + ka) * 4 + 2] = wall_clock64();
#endif
        }
        t = p.tb + nr;
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_max(&edge_ready, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        return true;
      };
      auto pause = [&]() {
        // (a spinning wave takes issue slots from the producer on its SIMD)
        if (X.poll_nap >= 4) __builtin_amdgcn_s_sleep(4);
        else if (X.poll_nap == 3) __builtin_amdgcn_s_sleep(3);
        else if (X.poll_nap == 2) __builtin_amdgcn_s_sleep(2);
        else __builtin_amdgcn_s_sleep(1);
      };
      constexpr int FK = ST_FK;
      edge_poll q[FK];
      unsigned long long t_begin = 0;
      bool timing = false;
      unsigned idle = 0;
      while (t < G) {
#pragma unroll
        for (int i = 0; i < FK; i++) {
          issue(q[i]);
          if (i == FK - 1) pd = lds_peek(&prod_done[0]);  // (for the next round)
          pause();
        }
        bool any = false;
#pragma unroll
        for (int i = 0; i < FK; i++) any = settle(q[i]) || any;
        if (any) {
          timing = false;
          idle = 0;
          continue;
        }
        // nothing new: the clock, the error word and the abort flag (a scalar-cache and a memory round
        // trip) are looked at once per 32 fruitless rounds only
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
          lds_post(&edge_ready, 0x7fffffff);  // release the producer: it runs on with stale edges
          break;
        }
      }
    }
  }
}


template <int C, int P, int MG, int NF, int DOT, int RD>
__global__ __launch_bounds__(64 * (P + MG * C * P + NF), chain_min_waves(C, P, MG, NF, RD)) void k_fill_chain(fill_args A,
                                                                                                   chain_args X) {
  chain_body<C, P, MG, NF, DOT, RD>(A, X);
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
    if (aborted || lds_peek(cnt) >= need) {
      asm volatile("" ::: "memory");
      return;
    }
    if (!chain_wait_slow(cnt, need, &s_abort, X.hdr, X.timeout, code, (unsigned)(j | (d << 16)), wave < P ? 1 : 2))
      aborted = true;
    asm volatile("" ::: "memory");
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
// host side

struct chain_geom {
  int C, P, MG, NF, RD, B, G;
  uint64_t EV, NP;
  size_t bytes;  // header + edge streams for D tables
};

static chain_geom chain_geometry(unsigned N, unsigned M, int D, bool summing = false) {
  chain_geom g;
  // strip shape by batch: 128-column strips on many compute units while the fill is a latency chain
  // (few tables: time = rows x row time of a strip + strips x hand-off, and a producer wave's row time
  // grows with its columns), 256-column strips -- half the hand-offs, fewer producer instructions per
  // cell -- beyond.  Measured on MI355X, N = M = 10^4: one table 0.82 ms with 128 columns against 1.0
  // with 256; eight tables 1.25 ms with 256 against 1.5 with 128.
  const uint64_t cols = (uint64_t)D * M;
  // (the summing kernels have little consumer work per strip: they are bound by the producers, and
  // narrow strips put more producer waves on a compute unit -- 64 discounts x 10^6 pairs at N = 10^4:
  // 2.5 ms with 128-column strips against 4.4 with 256)
  const bool few = summing || cols <= (uint64_t)stb_env_int("STB_CHAIN_NARROW_COLS", 50000);
  g.C = stb_env_int("STB_CHAIN_C", few ? 2 : 4);
  if (g.C != 1 && g.C != 2 && g.C != 4) g.C = few ? 2 : 4;
  g.P = stb_env_int("STB_CHAIN_P", 1);
  if (g.P != 1 && g.P != 2 && g.P != 4) g.P = 1;
  if (g.C * g.P > 4) g.P = 4 / g.C;
  g.MG = stb_env_int("STB_CHAIN_MG", 3);
  if (g.MG < 1) g.MG = 1;
  if (g.MG > 3) g.MG = 3;
  if (g.P + g.MG * g.C * g.P + 1 > 16) g.MG = 2;
  g.NF = stb_env_int("STB_CHAIN_NF", 1);
  if (g.NF < 1) g.NF = 1;
  if (g.NF > 3) g.NF = 3;
  g.RD = stb_env_int("STB_CHAIN_RD", g.P > 1 ? 8 : 4);
  if (g.RD != 2 && g.RD != 4 && g.RD != 8) g.RD = 4;
  const unsigned c = (M < N - 1) ? M : N - 1;  // columns 1..min(M, N-1) hold stored cells
  const unsigned W = 64u * g.C * g.P;
  g.B = (int)((c + W - 1) / W);
  if (g.B < 1) g.B = 1;
  g.G = (N > 2) ? (int)((N - 2 + ST_U - 1) / ST_U) : 0;
  g.EV = (uint64_t)3 + (uint64_t)g.G * ST_U + 136;  // the fetcher reads 128 rows at a time
  g.NP = (uint64_t)g.G + 24;                        // ... and 17 trip exponents
  g.bytes = 256 + (size_t)D * g.B * (g.EV + g.NP) * sizeof(unsigned long long);
  return g;
}

static size_t chain_s_workspace(unsigned N, unsigned M, int D) {
  if (N < 3 || M < 2 || D < 1) return 0;
  // (the strip width is a tunable: size for the narrowest, which has the most strips)
  chain_geom g = chain_geometry(N, M, D);
  const unsigned c = (M < N - 1) ? M : N - 1;
  const size_t Bmax = (c + 63) / 64;
  return 256 + (size_t)D * Bmax * (g.EV + g.NP) * sizeof(unsigned long long) + 256;
}

int stb_chain_tuning(unsigned N, unsigned M, int D, int *W_out) {
  const chain_geom g = chain_geometry(N, M, D);
  if (W_out) *W_out = 64 * g.C * g.P;
  return 0;
}

#ifdef STB_STAMPS
static unsigned long long *g_chain_dbg = nullptr;
#endif

int stb_launch_chain(fill_args &A, int D, char *ws, size_t ws_left, const dot_request *dot, unsigned **hdr_out,
                     hipStream_t st) {
  const unsigned N = A.N, M = A.M;
  const chain_geom sg = chain_geometry(N, M, D, dot != nullptr);
  int Pc = stb_period_rows(N);
  const int Penv = stb_env_int("STB_FILL_P", 0);
  if (Penv > 0 && Penv < Pc) Pc = Penv;
  chain_args X;
  memset(&X, 0, sizeof(X));
  X.TP = Pc / ST_U;
  if (X.TP < 1) return stb_fail("stb_fill_S: renormalisation period %d shorter than a trip", Pc);
  if (sg.RD > 2 * X.TP + 2) return stb_fail("stb_fill_S: ring of %d trips too deep for periods of %d trips", sg.RD, X.TP);
  X.tp_magic = (unsigned)((0x100000000ull + (unsigned)X.TP - 1) / (unsigned)X.TP);
  X.G = sg.G;
  X.D = D;
  X.B = sg.B;
  X.EV = sg.EV;
  X.NP = sg.NP;
  if (sg.bytes > ws_left) return stb_fail("stb_fill_S: workspace too small for the chain form");
  X.hdr = (unsigned *)ws;
  X.edge_e = (unsigned long long *)(ws + 256);
  X.edge_v = X.edge_e + (size_t)D * sg.B * X.NP;
  X.timeout = (unsigned long long)stb_env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;  // wall_clock64: 100 MHz
  X.poll_nap = stb_env_int("STB_CHAIN_POLL_NAP", 2);
  HIPCHK(hipMemsetAsync(ws, 0, stb_align_up(sg.bytes, 16), st));
  *hdr_out = X.hdr;
  if (stb_launch_s1(A, D, st)) return 1;
  // (cell lists are built for one form's tiling: col0 says which -- the chain form's is 1)
  if (dot && dot->item_ptr && dot->col0 != 1)
    return stb_fail("stb_groups_aterms: cell lists built for another form (layout %d) handed to the chain form", dot->col0);
  X.cnt = dot ? dot->cnt : nullptr;
  X.item_ptr = dot ? dot->item_ptr : nullptr;
  X.ent_pos = dot ? dot->ent_pos : nullptr;
  X.ent_cnt = dot ? dot->ent_cnt : nullptr;
  X.nsg = dot ? dot->nsg : 0;
  X.dotp = dot ? dot->dotp : nullptr;
  if (dot) const_cast<dot_request *>(dot)->parts_per_table = sg.B * sg.MG * sg.C * sg.P;
  if (dot && dot->dotp_cap && (size_t)D * (size_t)dot->parts_per_table > dot->dotp_cap) return stb_fail("stb_groups_aterms: partial-sum buffer too small");
  const int dk = !dot ? 0 : (dot->item_ptr ? 2 : 1);
#ifdef STB_STAMPS
  X.dbg = nullptr;
  X.mode = stb_env_int("STB_CHAIN_DEBUG", 0);
  if (getenv("STB_TIMELINE_FILE")) {
    if (!g_chain_dbg) HIPCHK(hipMalloc(&g_chain_dbg, sizeof(unsigned long long) * (160 * 1280 * 4 + 160 * 8 * 4 + 160 * 16 * 4 + 1024 * 4)));
    HIPCHK(hipMemsetAsync(g_chain_dbg, 0, sizeof(unsigned long long) * (160 * 1280 * 4 + 160 * 8 * 4 + 160 * 16 * 4 + 1024 * 4), st));
    X.dbg = g_chain_dbg;
  }
#endif
  const dim3 grid((unsigned)sg.B * (unsigned)D);
#define STRIP1(CC, PP, MM, FF, DD, RR) \
  STB_LAUNCH_SHM((k_fill_chain<CC, PP, MM, FF, DD, RR>), grid, dim3(64 * (PP + MM * CC * PP + FF)), \
                 (size_t)RR * ST_U * 64 * CC * PP * sizeof(double), st, A, X)
#define STRIP(CC, PP, MM, FF, RR)                   \
  do {                                              \
    if (dk == 2) STRIP1(CC, PP, MM, FF, 2, RR);     \
    else if (dk == 1) STRIP1(CC, PP, MM, FF, 1, RR); \
    else STRIP1(CC, PP, MM, FF, 0, RR);             \
  } while (0)
  const int shape = sg.C * 10000 + sg.P * 1000 + sg.MG * 100 + sg.NF * 10 + sg.RD;
  // (the summing variants are compiled for the two default strip shapes only)
  if (dk != 0 && shape != 21314 && shape != 41314 && shape != 41214 && shape != 41114)
    return stb_fail("stb_fill_S: the fused evaluation needs a default strip shape (unset STB_CHAIN_C / _P / _MG / _NF / _RD)");
  switch (shape) {
    case 21314: STRIP(2, 1, 3, 1, 4); break;  // few tables
    case 41314: STRIP(4, 1, 3, 1, 4); break;  // several
    case 41214: STRIP(4, 1, 2, 1, 4); break;
    case 41114: STRIP(4, 1, 1, 1, 4); break;
    case 21324: STRIP1(2, 1, 3, 2, 0, 4); break;
    case 21318: STRIP1(2, 1, 3, 1, 0, 8); break;
    case 21214: STRIP1(2, 1, 2, 1, 0, 4); break;
    case 11314: STRIP1(1, 1, 3, 1, 0, 4); break;
    case 12328: STRIP1(1, 2, 3, 2, 0, 8); break;
    case 12318: STRIP1(1, 2, 3, 1, 0, 8); break;
    case 12314: STRIP1(1, 2, 3, 1, 0, 4); break;
    case 14228: STRIP1(1, 4, 2, 2, 0, 8); break;
    case 22218: STRIP1(2, 2, 2, 1, 0, 8); break;
    default:
      return stb_fail("stb_fill_S: no chain kernel for C=%d P=%d MG=%d NF=%d RD=%d", sg.C, sg.P, sg.MG, sg.NF, sg.RD);
  }
#undef STRIP
#undef STRIP1
  HIPCHK(hipGetLastError());
#ifdef STB_STAMPS
  if (X.dbg) {
    HIPCHK(hipStreamSynchronize(st));
    const size_t cnt = (size_t)160 * 1280 * 4 + 160 * 8 * 4 + 160 * 16 * 4 + 1024 * 4;
    unsigned long long *h = (unsigned long long *)malloc(cnt * sizeof(*h));
    HIPCHK(hipMemcpy(h, X.dbg, cnt * sizeof(*h), hipMemcpyDeviceToHost));
    FILE *f = fopen(getenv("STB_TIMELINE_FILE"), "w");
    for (int jj = 0; jj < 160; jj++)
      for (int t = 0; t < 1280; t++) {
        unsigned long long *q = h + ((size_t)jj * 1280 + t) * 4;
        if (q[0] | q[1] | q[2] | q[3]) fprintf(f, "%d %d %llu %llu %llu %llu\n", jj, t, q[0], q[1], q[2], q[3]);
      }
    for (int jj = 0; jj < 160; jj++)
      for (int ww = 0; ww < 8; ww++) {  // per producer wave: re-looks, waits, waits on the left input, ticks waited
        unsigned long long *q = h + (size_t)160 * 1280 * 4 + ((size_t)jj * 8 + ww) * 4;
        if (q[0] | q[1] | q[2] | q[3]) fprintf(f, "%d %d %llu %llu %llu %llu\n", jj, 100000 + ww, q[0], q[1], q[2], q[3]);
      }
    for (int jj = 0; jj < 160; jj++)
      for (int ww = 0; ww < 16; ww++) {
        unsigned long long *q = h + (size_t)160 * 1280 * 4 + (size_t)160 * 8 * 4 + ((size_t)jj * 16 + ww) * 4;
        if (q[3]) fprintf(f, "%d %d %llu %llu %llu %llu\n", jj, 200000 + ww, q[0], q[1], q[2], q[3]);
      }
    fclose(f);
    if (getenv("STB_CENSUS_FILE")) {  // ticket, strip, table, HW_ID, XCC_ID, first and last wall clock of the producer, its shader cycles
      f = fopen(getenv("STB_CENSUS_FILE"), "w");
      for (int k = 0; k < 1024; k++) {
        unsigned long long *q = h + (size_t)160 * 1280 * 4 + 160 * 8 * 4 + 160 * 16 * 4 + (size_t)k * 4;
        if (q[0])
          fprintf(f, "%d %llu %llu %llu %llu %llu %llu %llu\n", k, q[0] & 0xffffull, (q[0] >> 16) & 0xffffull, q[1] & 0xffffffffull,
                  (q[1] >> 32) & 15ull, q[2], q[3], (q[1] >> 36) << 10);
      }
      fclose(f);
    }
    free(h);
  }
#endif
  return 0;
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
  const size_t s = chain_s_workspace(N, M, D);
  const size_t v = vchain_geometry(N, M, D).bytes;
  return (s > v ? s : v) + 256;
}

static unsigned long long chain_timeout_ticks() {
  return (unsigned long long)stb_env_int("STB_CHAIN_TIMEOUT_MS", 2000) * 100000ull;  // wall_clock64: 100 MHz
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

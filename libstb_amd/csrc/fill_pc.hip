// fill_pc.hip -- k_fill_pc, the producer/consumer fill launched per 128 rows, and k_s1.
//
// Replaces the table part of S_remake_part (reference lib/stable.c:321-388) when many tables are
// in flight, and is the fallback when a chain-form fill reports that a block gave up waiting: this
// form has no waits between workgroups at all (the kernel boundary publishes the frontier).

#include "stb_common.h"

// ---- producer/consumer form of the block-floating fill ---------------------------------------
//
// One workgroup = one column block of 256 columns = 1 producer wave + NCW consumer waves.
//  * The producer (wave 0) carries ONLY the recurrence, four columns per lane, and hands the raw
//    significands of its owned columns (the right-most 64*NCW; the rest is halo) to LDS.
//  * Each consumer wave turns 64 of them per row into logs and stores them -- 512 contiguous bytes
//    per row and wave -- so the log work (17 of ~25 instructions per cell) is spread over NCW other
//    SIMDs of the same CU, costs nothing on halo columns, and the table still moves 8 B per cell.
// Producer and consumers run PC_U rows apart through a two-slot LDS ring, one barrier per PC_U rows;
// the PC_U logs a consumer lane owns per trip are evaluated stage-major for ILP.
// log S^n_1 (the S1 vector) is not produced here: k_s1 evaluates lgamma(n-a) - lgamma(1-a).
#define PC_U 8
#ifndef PC_ST
#define PC_ST 4   // rows a consumer lane converts at once
#endif
#ifndef PC_MINW
#define PC_MINW 6  // waves per SIMD the register allocation leaves room for
#endif

template <int NCW>
// (three consumer waves: 90 registers, five waves a SIMD -- asking for six only made the compiler say it could not)
__global__ __launch_bounds__(64 * (1 + NCW), NCW == 3 ? PC_MINW - 1 : PC_MINW) void k_fill_pc(fill_args A, int k, int P) {
  constexpr int C = 4;
  constexpr int OW = 64 * NCW;      // owned (stored) columns per block
  constexpr int H = 256 - OW;       // halo columns recomputed per block
  __shared__ double2 lt[128];
  __shared__ __attribute__((aligned(32))) double vbuf[2][PC_U][OW];
  __shared__ int ebuf[2][OW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: the role branches are scalar branches)
  if (tid < 128) lt[tid] = A.lt[tid];

  const int j = blockIdx.x;
  const int d = blockIdx.y;
  const unsigned N = A.N, M = A.M;
  const int n0 = 2 + k * A.R;
  const int n1 = min((int)N, n0 + A.R - 1);
  const int nf = n0 - 1;
  const int cmin = 2 + j * OW - H;  // first column of the block (halo included)
  double *table = A.tables + (uint64_t)d * A.tstride;

  // ---- producer state: four columns per lane sharing ONE exponent, so that inside a lane the
  // left-neighbour term needs no rescaling (value of cell i = v[i] * 2^ep) ----
  double v[C], ca[C], s = 1.0;
  int ep = 1 + PC_BIAS;
  const int c0 = cmin + lane * C;
  const bool owned = lane * C >= H;
  if (wave == 0) {
    const double a = A.a[d];
    const uint64_t fbase = ((uint64_t)d * 2) * A.W;
    const double *fm_in = A.fm + fbase + (uint64_t)(k & 1) * A.W;
    const int *fe_in = A.fe + fbase + (uint64_t)(k & 1) * A.W;
    const int cmax_f = min(nf, (int)M);
    double m[C];
    int e[C], E = STB_EZ;
#pragma unroll
    for (int i = 0; i < C; i++) {
      const int c = c0 + i;
      m[i] = 0.0;
      e[i] = STB_EZ;
      if (k == 0) {
        if (c == 1) {
          m[i] = 0.5;
          e[i] = 1;
        }
      } else if (c >= 1 && c <= cmax_f) {
        m[i] = fm_in[c];
        e[i] = fe_in[c];
      }
      if (m[i] != 0.0) E = max(E, e[i]);
      ca[i] = (double)c * a;
    }
    if (E == STB_EZ) E = 1;  // an all-zero lane: exponent of the 1 the diagonal will bring
#pragma unroll
    for (int i = 0; i < C; i++) v[i] = (m[i] != 0.0) ? ldexp(m[i], max(e[i] - E, -1000) - PC_BIAS) : 0.0;
    ep = E + PC_BIAS;
  }
  // ---- consumer state ----
  const int ridx = (wave - 1) * 64 + lane;     // my slot in vbuf / ebuf (consumers only)
  const int cc = 2 + j * OW + ridx;            // my column

  int pidx = 0;
  for (int nb = n0; nb <= n1; nb += P, pidx++) {
    const int ne = min(n1, nb + P - 1);
    const int ns = max(nb, max(cmin, 3));  // rows above the block's first column are all zero
    if (wave == 0) {
      // period set-up: freeze the scale of the cross-lane input, s = 2^(ep_left - ep).  Adjacent
      // lanes (4 columns apart) differ by at most (N^2)^4, i.e. 8 log2 N <= 216 bits for N < 2^27,
      // and v_left <= 2^(-700 + 1450), so v_left * s stays below 2^970: no exponent adoption needed.
      const int epl = wave_shr1(ep, ep);
      s = ldexp(1.0, min(max(epl - ep, -1100), 220));
      if (owned) {
#pragma unroll
        for (int i = 0; i < C; i++) ebuf[pidx & 1][lane * C - H + i] = ep;
      }
      if (nb == 2 && cmin < 3) {  // row 2 (nothing is stored for it)
        const double t0 = wave_shr1_zero(v[3]) * s;
        v[3] = fma(1.0 - ca[3], v[3], v[2]);
        v[2] = fma(1.0 - ca[2], v[2], v[1]);
        v[1] = fma(1.0 - ca[1], v[1], v[0]);
        v[0] = fma(1.0 - ca[0], v[0], t0);
      }
    }
    __syncthreads();  // exponents (and, first time, the log table) visible to the consumers
    int myep = 0;
    if (wave > 0) myep = ebuf[pidx & 1][ridx];

    if (ns <= ne) {
      double coef[C];
      if (wave == 0) {
#pragma unroll
        for (int i = 0; i < C; i++) coef[i] = (double)(ns - 1) - ca[i];
      }
      // consumers address row r as rowbase(r)[coff]: a wave-uniform row base advanced by the row
      // pitch, plus a per-lane constant column offset
      double *rowbase = table + stb_row_offset((unsigned)ns, M);
      const int coff = cc - 2;
      const unsigned coff8 = (unsigned)coff * 8u;  // (cc >= 2)
      int one_hi = 0x3ff00000;
      asm volatile("" : "+v"(one_hi));  // (keep the pattern in a vector register)
      const int trips = (ne - ns + 1 + PC_U - 1) / PC_U;
      // trip q: the producer computes rows ns+U*q.., the consumers emit the rows of trip q-1
      for (int q = 0; q <= trips; q++) {
        if (wave == 0) {
          if (q < trips) {
            const int r0 = ns + q * PC_U;
            const int cnt = min(PC_U, ne - r0 + 1);
            if (cnt == PC_U) {
              // a full trip, unrolled; ds_write2_b64 takes any two register pairs where a 16-byte store
              // wants four adjacent registers, i.e. eight register copies a row
              unsigned dst = lds_addr_of(&vbuf[q & 1][0][owned ? lane * C - H : 0]);
              const unsigned step = (unsigned)(OW * sizeof(double));
#pragma unroll
              for (int u = 0; u < PC_U; u++) {
                const double t0 = wave_shr1_zero(v[3]) * s;
                v[3] = fma(coef[3], v[3], v[2]);
                v[2] = fma(coef[2], v[2], v[1]);
                v[1] = fma(coef[1], v[1], v[0]);
                v[0] = fma(coef[0], v[0], t0);
#pragma unroll
                for (int i = 0; i < C; i++) coef[i] += 1.0;
                if (owned) {
                  lds_store2<0>(dst, v[0], v[1]);
                  lds_store2<2>(dst, v[2], v[3]);
                }
                dst += step;
              }
            } else
            for (int u = 0; u < cnt; u++) {
              const double t0 = wave_shr1_zero(v[3]) * s;
              v[3] = fma(coef[3], v[3], v[2]);
              v[2] = fma(coef[2], v[2], v[1]);
              v[1] = fma(coef[1], v[1], v[0]);
              v[0] = fma(coef[0], v[0], t0);
#pragma unroll
              for (int i = 0; i < C; i++) coef[i] += 1.0;
              if (owned) {
                // (ds_write2_b64 takes any two register pairs; a 16-byte store wants four adjacent
                // registers and the compiler pays for them with eight register copies a row)
                const unsigned dst = lds_addr_of(&vbuf[q & 1][u][lane * C - H]);
                lds_store2<0>(dst, v[0], v[1]);
                lds_store2<2>(dst, v[2], v[3]);
              }
            }
          }
        } else if (q > 0) {
          const int r0 = ns + (q - 1) * PC_U;
          const int cnt = min(PC_U, ne - r0 + 1);
          if (cnt == PC_U) {
            // U rows of my column, stage-major PC_ST rows at a time (8 at once need 116 registers: 4
            // waves per SIMD; 4 at once fit 80: 6 waves, i.e. 8 workgroups per compute unit)
            const unsigned pitch = stb_row_pitch((unsigned)r0, M);
            const bool same_pitch = stb_row_pitch((unsigned)(r0 + PC_U - 1), M) == pitch;
#pragma unroll
            for (int h = 0; h < PC_U; h += PC_ST) {
              double x[PC_ST], z[PC_ST], kf[PC_ST], r[PC_ST], pl[PC_ST];
              double2 t[PC_ST];
#pragma unroll
              for (int u = 0; u < PC_ST; u++) x[u] = vbuf[(q - 1) & 1][h + u][ridx];
#pragma unroll
              for (int u = 0; u < PC_ST; u++) t[u] = lt[(__double2hiint(x[u]) >> 13) & 127];
#pragma unroll
              for (int u = 0; u < PC_ST; u++) {
                const int hi = __double2hiint(x[u]);
                z[u] = __hiloint2double(mantissa_of_one(hi, one_hi), __double2loint(x[u]));
                kf[u] = (double)((int)((hi >> 20) & 0x7ff) - 1023 + myep);
              }
#pragma unroll
              for (int u = 0; u < PC_ST; u++) r[u] = fma(z[u], t[u].x, -1.0);
#pragma unroll
              for (int u = 0; u < PC_ST; u++) pl[u] = fma(r[u], 0.2, -0.25);
#pragma unroll
              for (int u = 0; u < PC_ST; u++) pl[u] = fma(r[u], pl[u], 1.0 / 3.0);
#pragma unroll
              for (int u = 0; u < PC_ST; u++) pl[u] = fma(r[u], pl[u], -0.5);
#pragma unroll
              for (int u = 0; u < PC_ST; u++) pl[u] = fma(r[u], pl[u], 1.0);
              if (same_pitch) {
#pragma unroll
                for (int u = 0; u < PC_ST; u++)
                  store_sbase(rowbase + (size_t)u * pitch, coff8, fma(kf[u], 0.693147180559945309417, fma(r[u], pl[u], t[u].y)));
                rowbase += (size_t)PC_ST * pitch;
              } else {
#pragma unroll
                for (int u = 0; u < PC_ST; u++) {
                  store_sbase(rowbase, coff8, fma(kf[u], 0.693147180559945309417, fma(r[u], pl[u], t[u].y)));
                  rowbase += stb_row_pitch((unsigned)(r0 + h + u), M);
                }
              }
            }
          } else {
            for (int u = 0; u < cnt; u++) {
              rowbase[coff] = bfp_log(vbuf[(q - 1) & 1][u][ridx], myep, lt);
              rowbase += stb_row_pitch((unsigned)(r0 + u), M);
            }
          }
        }
        lds_barrier();
      }
    }
    if (wave == 0) {
      // renormalise the lane: the largest of the four significands back to 2^-PC_BIAS * [0.5,1)
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
  }

  if (wave == 0 && n1 < (int)N) {
    const uint64_t fbase = ((uint64_t)d * 2) * A.W;
    double *fm_out = A.fm + fbase + (uint64_t)((k + 1) & 1) * A.W;
    int *fe_out = A.fe + fbase + (uint64_t)((k + 1) & 1) * A.W;
#pragma unroll
    for (int i = 0; i < C; i++) {
      const int c = c0 + i;
      const bool mine = owned || (j == 0 && c == 1);
      if (mine && c >= 1 && c <= (int)M) {
        fm_out[c] = __builtin_amdgcn_frexp_mant(v[i]);
        fe_out[c] = (v[i] != 0.0) ? ep + __builtin_amdgcn_frexp_exp(v[i]) : STB_EZ;
      }
    }
  }
}

// S1[n-1] = log S^n_1 = log Gamma(n-a)/Gamma(1-a) for n = 1..N, all tables
__global__ __launch_bounds__(256) void k_s1(const double *a, double *S1, uint64_t s1stride, unsigned N) {
  const int d = blockIdx.y;
  const double ad = a[d];
  const double lg1 = lgamma(1.0 - ad);
  for (unsigned n = 1 + blockIdx.x * blockDim.x + threadIdx.x; n <= N; n += gridDim.x * blockDim.x)
    S1[(uint64_t)d * s1stride + n - 1] = (n == 1) ? 0.0 : lgamma((double)n - ad) - lg1;
}

// ... and, in the same launch, a workspace zeroed (the halo-block form's records are their own flags: 0 = not written yet).
// One launch instead of a memset and k_s1: a launch and the gap before the next are ~10 us of a 0.3 ms step.
// the first nb * D blocks are k_s1's; the nz blocks after them zero n16 x 16 bytes from `ws` (16-byte aligned).
// (av, Da: the discounts themselves when they are still on their way to `a` -- written there for the fill that follows)
__global__ __launch_bounds__(256) void k_prep(double *a, double *S1, uint64_t s1stride, unsigned N, unsigned nz, unsigned nb,
                                              uint4 *ws, uint64_t n16, stb_a64 av, int Da, unsigned ns) {
  // (the ns = nb * D blocks of S1 come first: a chain of lgamma evaluations is latency, the zeroing bandwidth -- they overlap)
  if (blockIdx.x >= ns) {
    const uint4 z = {0u, 0u, 0u, 0u};
    for (uint64_t i = (uint64_t)(blockIdx.x - ns) * 256 + threadIdx.x; i < n16; i += (uint64_t)nz * 256) ws[i] = z;
    return;
  }
  const unsigned bi = blockIdx.x, d = bi / nb, bx = bi % nb;
  const double ad = Da > 0 ? av.v[d] : a[d];
  if (Da > 0 && bx == 0 && threadIdx.x == 0) a[d] = ad;
  const double lg1 = lgamma(1.0 - ad);
  for (unsigned n = 1 + bx * 256 + threadIdx.x; n <= N; n += nb * 256)
    S1[(uint64_t)d * s1stride + n - 1] = (n == 1) ? 0.0 : lgamma((double)n - ad) - lg1;
}

int stb_launch_prep(const fill_args &A, int D, void *ws, size_t zero_bytes, hipStream_t st) {
  if ((zero_bytes & 15) || ((uintptr_t)ws & 15)) return stb_fail("stb_fill: workspace of %zu bytes at %p is not 16-byte aligned", zero_bytes, ws);
  const unsigned nb = (A.N + 255) / 256 < 64 ? (A.N + 255) / 256 : 64;
  const uint64_t n16 = zero_bytes / 16;
  unsigned nz = (unsigned)((n16 + 1023) / 1024);  // four stores a thread at least
  if (nz > 2048) nz = 2048;
  if (nz < 1) nz = 1;
  stb_a64 av;
  int Da = 0;
  if (!stb_a_take(&av, &Da)) Da = 0;
  if (Da != 0 && Da != D) return stb_fail("stb_fill: %d discounts on their way, %d tables", Da, D);
  hipLaunchKernelGGL(k_prep, dim3(nz + nb * (unsigned)D), dim3(256), 0, st, const_cast<double *>(A.a), A.S1, A.s1stride, A.N, nz, nb, (uint4 *)ws, n16,
                     av, Da, nb * (unsigned)D);
  return 0;
}

int stb_launch_s1(const fill_args &A, int D, hipStream_t st) {
  const unsigned nb = (A.N + 255) / 256 < 64 ? (A.N + 255) / 256 : 64;
  hipLaunchKernelGGL(k_s1, dim3(nb, D), dim3(256), 0, st, A.a, A.S1, A.s1stride, A.N);
  return 0;
}

// rows 2..N in launches of A.R rows; A.H = 256 - 64 * consumers; A.R <= A.H
//
// Every launch ends with compute units running dry while its last workgroups finish (a workgroup
// takes ~20 us, a launch of 64 tables 30-100 us), and the next launch cannot start before: the kernel
// boundary is what publishes the frontier.  Tables do not depend on each other, so a large batch is
// cut into STB_PC_STREAMS (default 2 from 24 tables on) sub-batches whose launches go to streams of
// their own, forked from and joined to the caller's stream by events: one sub-batch's tail is filled
// by the other's workgroups.
struct pc_side_streams {
  hipStream_t s[3] = {nullptr, nullptr, nullptr};
  hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
};
static thread_local pc_side_streams g_side[16];

static fill_args pc_slice(const fill_args &A, int d0) {
  fill_args B = A;
  B.a = A.a + d0;
  B.tables = A.tables + (uint64_t)d0 * A.tstride;
  B.S1 = A.S1 + (uint64_t)d0 * A.s1stride;
  B.fm = A.fm + (uint64_t)d0 * 2 * A.W;
  B.fe = A.fe + (uint64_t)d0 * 2 * A.W;
  return B;
}

int stb_launch_pc(fill_args &A, int D, hipStream_t st) {
  const int N = (int)A.N, M = (int)A.M, R = A.R;
  const int ncw = (256 - A.H) / 64;
  const int OW = 64 * ncw;
  const int P = [&] {  // equal-length renormalisation periods inside a launch
    int p = stb_period_rows(A.N);
    const int penv = stb_env_int("STB_FILL_P", 0);
    if (penv > 0 && penv < p) p = penv;
    if (p >= R) return R;
    const int per = (R + p - 1) / p;
    return (R + per - 1) / per;
  }();
  int ns = stb_env_int("STB_PC_STREAMS", D >= 24 ? 2 : 1);
  if (ns < 1) ns = 1;
  if (ns > 4) ns = 4;
  if (ns > D) ns = D;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  pc_side_streams &S = g_side[dev & 15];
  hipStream_t str[4] = {st, nullptr, nullptr, nullptr};
  if (ns > 1) {
    if (!S.fork) HIPCHK(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming));
    HIPCHK(hipEventRecord(S.fork, st));
    for (int i = 1; i < ns; i++) {
      if (!S.s[i - 1]) {
        HIPCHK(hipStreamCreateWithFlags(&S.s[i - 1], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&S.join[i - 1], hipEventDisableTiming));
      }
      str[i] = S.s[i - 1];
      HIPCHK(hipStreamWaitEvent(str[i], S.fork, 0));
    }
  }
  fill_args sub[4];
  int dn[4];
  for (int i = 0; i < ns; i++) {
    const int d0 = (int)((int64_t)D * i / ns), d1 = (int)((int64_t)D * (i + 1) / ns);
    sub[i] = pc_slice(A, d0);
    dn[i] = d1 - d0;
    stb_launch_s1(sub[i], dn[i], str[i]);
  }
  const int nlaunch = (N - 1 + R - 1) / R;
  for (int k = 0; k < nlaunch; k++) {
    int n1 = 2 + (k + 1) * R - 1;
    if (n1 > N) n1 = N;
    int ncols = (n1 < M ? n1 : M) - 1;
    if (ncols < 1) ncols = 1;
    for (int i = 0; i < ns; i++) {
      const dim3 grid((ncols + OW - 1) / OW, dn[i]);
      if (ncw == 3)
        STB_LAUNCH((k_fill_pc<3>), grid, dim3(256), str[i], sub[i], k, P);
      else
        STB_LAUNCH((k_fill_pc<2>), grid, dim3(192), str[i], sub[i], k, P);
    }
  }
  HIPCHK(hipGetLastError());
  for (int i = 1; i < ns; i++) {
    HIPCHK(hipEventRecord(S.join[i - 1], str[i]));
    HIPCHK(hipStreamWaitEvent(st, S.join[i - 1], 0));
  }
  return 0;
}

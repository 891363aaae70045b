// fill.hip -- stb_fill_S / stb_fill_V: the device form of S_make / S_remake's table fill
// (reference lib/stable.c:321-388 for S, :451-482 for V).  Chooses the kernel form, lays out the
// workspace, keeps the status of the last chain-form fill and the optional per-launch timing.
//
// Forms (kernels in their own translation units):
//   hb      k_fill_hb                      one launch: a spine that walks blocks of rows alone behind a halo + tile
//                                          workers (default wherever the row chain decides: tables of >= 512 rows
//                                          while the spine fits, e.g. 1-24 tables of 10^4 columns; the fused aterms
//                                          of 2-24 discounts)
//   chain   k_fill_chain / k_fillv_chain   one launch per fill (small tables, many short ones, the V table, the fused
//                                          aterms of more than 24 discounts)
//   ck      k_fill_ck                      (ablation build only) one launch: recurrence-only spine + tile workers
//   pc      k_fill_pc                      launched per 128 rows (many tables; fallback of the one-launch forms)
//   rows    k_fill_rows                    the reference's own operation order (STB_FILL_LOGDOMAIN)
//   + the superseded forms of tools/ablation/ablation.hip in the library `make -C tools/ablation` builds

#include "stb_common.h"

static unsigned frontier_pitch(unsigned M) { return (unsigned)stb_align_up((size_t)M + 2, 64); }

// A cell grows per row by U^n_m = n - m a + S^n_{m-1}/S^n_m, and the last term reaches n(n-1)/2
// next to the diagonal, so the bound is N^2 per row, not N.  Significands start a period at 2^-700
// and may use 1450 bits.
int stb_period_rows(unsigned N) {
  int bits = 1;
  while ((1ull << bits) < (unsigned long long)N) bits++;
  const int p = 1450 / (2 * bits + 1);
  return p < 1 ? 1 : p;
}

// ------------------------------------------------------------------------------------------------
// per-launch timing: when armed, every fill kernel is launched with a begin/end event pair
// (hipExtLaunchKernelGGL stamps them with the dispatch's own start/stop, i.e. what a kernel trace
// reports), so a caller can obtain the kernel-only time of a fill without a profiler attached.
struct fill_prof {
  bool armed = false;
  int used = 0;
  hipEvent_t ev[2 * 4096];
  int made = 0;
  double span_ms = 0.0;
};
static thread_local fill_prof g_prof;

void stb_prof_events(hipEvent_t *e0, hipEvent_t *e1) {
  *e0 = *e1 = nullptr;
  if (!g_prof.armed || g_prof.used + 2 > 2 * 4096) return;
  while (g_prof.made < g_prof.used + 2) {
    if (hipEventCreate(&g_prof.ev[g_prof.made]) != hipSuccess) return;
    g_prof.made++;
  }
  *e0 = g_prof.ev[g_prof.used];
  *e1 = g_prof.ev[g_prof.used + 1];
  g_prof.used += 2;
}

extern "C" void stb_fill_profile_begin(void) {
  g_prof.armed = true;
  g_prof.used = 0;
}

extern "C" int stb_fill_profile_end(double *kernel_ms_total, int *launches) {
  STB_ENTRY;
  // caller must have synchronised the stream(s) the fills ran on
  g_prof.armed = false;
  double tot = 0.0;
  const int n = g_prof.used / 2;
  for (int i = 0; i < n; i++) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
    tot += ms;
  }
  // device time from the first launch's start to the last end: what the launches took together when
  // some of them ran side by side on forked streams (the sum above counts such time twice)
  double span = 0.0;
  for (int i = 0; i < n; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof.ev[0], g_prof.ev[2 * i + 1]) == hipSuccess && ms > span) span = ms;
  }
  g_prof.span_ms = span;
  if (kernel_ms_total) *kernel_ms_total = tot;
  if (launches) *launches = n;
  g_prof.used = 0;
  return 0;
}

extern "C" double stb_fill_profile_span(void) { return g_prof.span_ms; }

// ------------------------------------------------------------------------------------------------
// workspace

static size_t ws_head(unsigned M, int D) {  // discounts + frontier
  const size_t W = frontier_pitch(M);
  return stb_align_up((size_t)D * sizeof(double), 256) + stb_align_up((size_t)D * 2 * W * (sizeof(double) + sizeof(int)), 256);
}

static size_t fill_workspace_need(unsigned N, unsigned M, int D) {
  size_t form = stb_chain_workspace(N, M, D);
  const size_t ck = stb_launch_ck ? stb_ck_workspace(N, M, D) : 0;
  if (ck > form) form = ck;
  const size_t hb = stb_hb_workspace(N, M, D);
  if (hb > form) form = hb;
  const size_t gr = stb_grid_workspace(N, M, D);
  if (gr > form) form = gr;
  if (stb_ablation_workspace) {
    const size_t ab = stb_ablation_workspace(N, M, D);
    if (ab > form) form = ab;
  }
  return ws_head(M, D) + form + 512;
}

// enough for D tables and for any smaller batch run in the same workspace
extern "C" size_t stb_fill_workspace_bytes(unsigned N, unsigned M, int D) {
  size_t need = fill_workspace_need(N, M, D);
  for (int d2 = 1; d2 < D && d2 <= 4096; d2++) {  // (the block shape, hence the edge streams, depends on the batch)
    const size_t n2 = fill_workspace_need(N, M, d2);
    if (n2 > need) need = n2;
  }
  return need;
}

// ------------------------------------------------------------------------------------------------
// choice of form

extern "C" int stb_default_variant(void) {
  const int v = stb_env_int("STB_FILL_VARIANT", STB_FILL_SCALED);
  return (v == STB_FILL_LOGDOMAIN || v == STB_FILL_SCALED_STEP || v == STB_FILL_SPLIT || v == STB_FILL_FUSED ||
          v == STB_FILL_PC || v == STB_FILL_CHAIN || v == STB_FILL_CHAINX || v == STB_FILL_CK || v == STB_FILL_HB)
             ? v
             : STB_FILL_SCALED;
}

extern "C" int stb_has_ablation(void) { return stb_ablation_fill != nullptr; }

// STB_FILL_SCALED picks the form by how many table columns are in flight: the chain form (one
// launch, no halo) up to STB_CHAIN_MAX_COLS columns over all tables, the producer/consumer form
// beyond.  (MI355X, tools/ab_chain.py: 16 tables of 10^4 columns 2.01 ms pc against 2.11 chain, 12
// tables 1.78 against 1.68; 32 tables of 4000 columns 0.78 against 0.70.)
static bool g_dot_req_active();  // (a fused aterms evaluation is pending on this thread: that is the chain form's)
static bool chain_wins(unsigned N, unsigned M, int D) {
  const uint64_t cap = (uint64_t)stb_env_int("STB_CHAIN_MAX_COLS", 150000);
  return (uint64_t)D * M <= cap && N >= 3 && N < (1u << 27);
}

// ... and between the two, by cells in all: the checkpointed form (spine + tile workers) from ~4 x 10^7 cells
// (one table of 10^4 columns, 8 of 4000) to ~10^9 (20 tables of 10^4), where the table traffic of the
// producer/consumer form's many launches catches up.  (MI355X, tools/sweep_forms.sh / tools/ab_ck.py,
// N = M = 10^4: 1 table 0.65 ms against 0.68 chain, 4 tables 0.75-0.79 against 1.03, 8 tables 0.90-0.98
// against 1.16-1.36, 16 tables 1.68-1.73 against 2.02 pc, 24 tables 2.48 against 2.40 pc; N = M = 4000:
// 8 tables 0.33 = chain, 16 tables 0.36 against 0.47, 64 tables 1.04 = pc.)
// STB_CK=0 / 1 switches it off / on wherever it is eligible.
static bool ck_wins(unsigned N, unsigned M, int D) {
  // (since the halo-block form took over the batches this one was made for, it is chosen on request only: STB_CK=1,
  // or the variant STB_FILL_CK; what the halo-block form leaves -- very many mid-sized tables -- is the
  // producer/consumer form's: 64 tables of 4000 columns 1.04 against 1.17 ms, 128 of 2000 0.56 against 0.68)
  const int force = stb_env_int("STB_CK", 0);
  if (force == 0 || g_dot_req_active() || !stb_launch_ck || !stb_ck_eligible(N, M, D)) return false;
  if (force > 0) return true;
  const uint64_t cells = (uint64_t)D * stb_table_cells(N, M);
  return cells >= (uint64_t)stb_env_int("STB_CK_MIN_MCELLS", 40) * 1000000ull &&
         cells <= (uint64_t)stb_env_int("STB_CK_MAX_MCELLS", 1000) * 1000000ull;
}

// The halo-block form (a spine that walks blocks of rows alone behind a halo + tile workers) is the fast one
// wherever the ROW CHAIN, not the chip's throughput, decides.  Chosen for tables of >= 512 rows while the batch's
// spine workgroups leave the tile workers room (25/32 of the compute units at most, 200 of an MI355X's 256: 24 tables of 10^4 columns, 64 of 4000) and the batch
// stays below 1.25 x 10^9 cells.  (MI355X, tools/ab_ck.py, ms: N = M = 10^4: 1 table 0.32-0.33 against 0.70 chain /
// 0.71 checkpointed, 2 tables 0.365 against 0.72, 4 tables 0.48-0.50 against 0.76, 8 tables 0.73-0.75 against
// 0.90-0.93, 16 tables 1.34-1.50 against 1.51-1.65, 24 tables 2.13 against 2.37 pc, 32 tables 4.3 against 2.9 pc;
// N = M = 4000: 1 table 0.15-0.16 against 0.28 chain, 3 tables 0.158 against 0.289, 8 tables 0.20 against 0.34,
// 64 tables 0.99 against 1.02 pc; N = M = 2000: 1 / 3 tables 0.095 / 0.10 against 0.15, 64 tables 0.32 against
// 0.37 chain, 128 tables 0.61 against 0.55 pc; N = M = 1000: 1 / 8 / 16 / 32 / 64 tables 0.06 / 0.067 / 0.075 /
// 0.087 / 0.122 against 0.08 / 0.086 / 0.089 / 0.090 / 0.118; N = M = 512: 0.05 = chain; N = M = 300: 0.05
// against 0.04; N = 3000, M = 200, 100 tables 0.23 against 0.25; N = 50000, M = 100, 4 tables 1.18 against 2.32;
// N = M = 20000: 1 table 0.75 against 1.44.)
// STB_HB=0 / 1 switches it off / on wherever it is eligible.
static bool hb_wins(unsigned N, unsigned M, int D) {
  const int force = stb_env_int("STB_HB", -1);
  if (force == 0 || g_dot_req_active() || !stb_hb_eligible(N, M, D)) return false;
  if (force > 0) return true;
  const uint64_t cells = (uint64_t)D * stb_table_cells(N, M);
  if (N < (unsigned)stb_env_int("STB_HB_MIN_N", 512) || cells > (uint64_t)stb_env_int("STB_HB_MAX_MCELLS", 1250) * 1000000ull) return false;
  return stb_hb_spine(N, M, D) <= (unsigned)stb_env_int("STB_HB_MAX_SPINE", stb_cu_count() * 25 / 32);
}

enum { FORM_ROWS_LOG, FORM_PC, FORM_CHAIN, FORM_ABLATION, FORM_CK, FORM_HB };

static int pick_form(int variant, unsigned N, unsigned M, int D) {
  switch (variant) {
    case STB_FILL_LOGDOMAIN: return FORM_ROWS_LOG;
    case STB_FILL_PC: return (N < (1u << 27)) ? FORM_PC : FORM_ROWS_LOG;
    case STB_FILL_CHAIN: return (N >= 3 && N < (1u << 27)) ? FORM_CHAIN : (N < 3 ? FORM_PC : FORM_ROWS_LOG);
    case STB_FILL_CK: return (stb_launch_ck && stb_ck_eligible(N, M, D)) ? FORM_CK : pick_form(STB_FILL_CHAIN, N, M, D);
    case STB_FILL_HB: return stb_hb_eligible(N, M, D) ? FORM_HB : pick_form(STB_FILL_CHAIN, N, M, D);
    case STB_FILL_CHAINX:  // (its converter blocks need a compute unit per 64-column chunk)
      if (D > 2) return pick_form(STB_FILL_CHAIN, N, M, D);
      return (N >= 3 && N < (1u << 27)) ? FORM_ABLATION : pick_form(STB_FILL_CHAIN, N, M, D);
    case STB_FILL_SCALED_STEP:
    case STB_FILL_SPLIT:
    case STB_FILL_FUSED: return FORM_ABLATION;
    default:
      // the block-floating forms bound the scale between lanes for N < 2^27 (see k_fill_pc)
      if (N >= (1u << 27)) return FORM_ROWS_LOG;
      // (a GPU shared with other processes: the form without waits between workgroups, see abi.hip)
      if (stb_shared_gpu() && !g_dot_req_active()) return FORM_PC;
      if (hb_wins(N, M, D)) return FORM_HB;
      if (ck_wins(N, M, D)) return FORM_CK;
      return chain_wins(N, M, D) ? FORM_CHAIN : FORM_PC;
  }
}

extern "C" int stb_fill_tuning(unsigned N, unsigned M, int D, int *C_out, int *R_out, int *launches) {
  const int form = pick_form(stb_default_variant(), N, M, D);
  if (form == FORM_CHAIN) {
    int P = 0;
    stb_chain_tuning(N, M, D, &P);
    if (C_out) *C_out = P;
    if (R_out) *R_out = (int)N;
    if (launches) *launches = 1;
    return 3;
  }
  if (form == FORM_CK) {
    int W = 0, rows = 0;
    stb_ck_tuning(N, M, D, &W, &rows);
    if (C_out) *C_out = W;     // columns of a wave strip
    if (R_out) *R_out = rows;  // rows of a tile
    if (launches) *launches = 1;
    return 4;
  }
  if (form == FORM_HB) {
    int W = 0, rows = 0;
    stb_hb_tuning(N, M, D, &W, &rows);
    if (C_out) *C_out = W;     // own columns of a strip
    if (R_out) *R_out = rows;  // rows of a block
    if (launches) *launches = 1;
    return 6;
  }
  int C = (form == FORM_PC) ? 4 : stb_env_int("STB_FILL_C", 2);
  int R = stb_env_int("STB_FILL_R", form == FORM_PC ? 128 : 64);
  if (form == FORM_PC && R > 128) R = 128;
  if (R < 1) R = 1;
  if (C_out) *C_out = C;
  if (R_out) *R_out = R;
  if (launches) *launches = ((int)N - 1 + R - 1) / R;
  return form == FORM_PC ? 2 : 5; /* 2 producer/consumer (k_fill_pc), 3 chain, 4 checkpointed, 5 another form */
}

// ------------------------------------------------------------------------------------------------
// the fill

// what the last fill of this thread was, so that stb_fill_status can wait for it, report a chain
// form that gave up, and repeat the fill with the producer/consumer form (struct in stb_common.h: an
// object that queues a fill and checks it later, possibly after other fills, keeps a copy)
static thread_local last_fill g_last;
static thread_local unsigned g_fallbacks = 0;
static thread_local const dot_request *g_dot_req = nullptr;

void stb_set_dot_request(const dot_request *r) { g_dot_req = r; }
static bool g_dot_req_active() { return g_dot_req != nullptr; }

extern "C" unsigned stb_fill_fallbacks(void) { return g_fallbacks; }

static void pc_geometry(fill_args &A) {
  const int ncw = stb_env_int("STB_PC_CONSUMERS", 2) == 3 ? 3 : 2;
  A.H = 256 - 64 * ncw;
  A.R = stb_env_int("STB_FILL_R", A.H);
  if (A.R > A.H) A.R = A.H;
  if (A.R < 1) A.R = 1;
  A.Wv = 256 - A.H;
}

void stb_fill_last(last_fill *out) { *out = g_last; }

// Waits for the fill on ITS stream (a copy on the null stream does not wait for a non-blocking stream,
// and torch's side streams are non-blocking), reads the header, and on a give-up repeats the fill with
// the form that has no waits between workgroups.
int stb_fill_status_of(last_fill *lf) {
  if (!lf->hdr) return 0;
  unsigned h[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(h, lf->hdr, lf->stamped ? sizeof(h) : 4 * sizeof(unsigned), hipMemcpyDeviceToHost, lf->st));
  HIPCHK(hipStreamSynchronize(lf->st));
  if (lf->stamped) {
    // (the halo-block form stamps its first workgroup's start and its spine waves' ends, 100 MHz ticks)
    const unsigned long long t0 = (unsigned long long)h[STB_HDR_T0] | ((unsigned long long)h[STB_HDR_T0 + 1] << 32);
    const unsigned long long t1 = (unsigned long long)h[STB_HDR_T1] | ((unsigned long long)h[STB_HDR_T1 + 1] << 32);
    if (t0 && t1 > t0)
      stb_note_span((double)(t1 - t0) * 1e-5, 1e-3 * (100.0 + 0.05 * (double)lf->A.N * (1.0 + (double)lf->D / 8.0)), "a table fill (stb_fill_S)");
  }
  if (h[1] == 0) return 0;
  stb_fail("%s: the fill gave up waiting for a neighbour block (code 0x%x, block %u of table %u)",
           lf->s_table ? "stb_fill_S" : "stb_fill_V", h[1], h[2] & 0xffffu, h[2] >> 16);
  if (!lf->can_fall_back || stb_env_int("STB_CHAIN_NO_FALLBACK", 0)) {
    lf->hdr = nullptr;
    return 1;
  }
  lf->hdr = nullptr;
  lf->fell_back = true;
  g_fallbacks++;
  fill_args A = lf->A;
  pc_geometry(A);
  if (stb_launch_pc(A, lf->D, lf->st)) return 1;
  HIPCHK(hipStreamSynchronize(lf->st));
  return 0;
}

extern "C" int stb_fill_status(void) {
  STB_ENTRY;
  return stb_fill_status_of(&g_last);
}

// The halo-block form also stores floats (S_FLOAT) and V ratios: its tile workers narrow or divide what they hold in
// double before the store -- the only form that does.  Taken for kinds 1-3 wherever it is taken for log S in double.
static bool hb_takes_kind(unsigned N, unsigned M, int D, int kind) {
  if (stb_env_int("STB_HB", -1) == 0 || stb_shared_gpu() || !stb_hb_eligible_out(N, M, D, kind)) return false;
  const uint64_t cells = (uint64_t)D * stb_table_cells(N, M);
  if (N < (unsigned)stb_env_int("STB_HB_MIN_N", 512) || cells > (uint64_t)stb_env_int("STB_HB_MAX_MCELLS", 1250) * 1000000ull) return false;
  return stb_hb_spine(N, M, D) <= (unsigned)stb_env_int("STB_HB_MAX_SPINE", stb_cu_count() * 25 / 32);
}

extern "C" int stb_fill_takes_kind(unsigned N, unsigned M, int D, int kind) { return hb_takes_kind(N, M, D, kind) ? 1 : 0; }

static thread_local stb_a64 g_a_vals;
static thread_local int g_a_D = 0;  // > 0: that many discounts of this thread's fill are not on the device yet
__global__ void k_set_a(stb_a64 av, double *a, int D) {
  if ((int)threadIdx.x < D) a[threadIdx.x] = av.v[threadIdx.x];
}
void stb_a_defer(const double *a_host, int D) {
  for (int d = 0; d < D; d++) g_a_vals.v[d] = a_host[d];
  g_a_D = D;
}
bool stb_a_take(stb_a64 *out, int *D_out) {
  if (g_a_D <= 0) return false;
  *out = g_a_vals;
  *D_out = g_a_D;
  g_a_D = 0;
  return true;
}
int stb_a_flush(const fill_args &A, hipStream_t st) {
  stb_a64 av;
  int D = 0;
  if (!stb_a_take(&av, &D)) return 0;
  hipLaunchKernelGGL(k_set_a, dim3(1), dim3(64), 0, st, av, const_cast<double *>(A.a), D);
  HIPCHK(hipGetLastError());
  return 0;
}

static thread_local bool g_fillv_exact = false;  // stb_fill_V_exact: this call walks the reference's own V recurrence

// kind: 0 log S (double), 1 log S (float), 2 V (double), 3 V (float); d_tables is the slab of that type
static int fill_common(const double *a_host, int D, unsigned N, unsigned M, double *d_tables, uint64_t table_stride,
                       double *d_S1, uint64_t s1_stride, void *d_ws, size_t ws_bytes, int variant, int kind,
                       hipStream_t st) {
  const bool vtable = (kind & 2) != 0;
  const char *who = vtable ? "stb_fill_V" : "stb_fill_S";
  g_last.hdr = nullptr;  // whatever this thread filled before is no longer "the last fill"
  g_last.fell_back = false;
  g_last.stamped = false;
  struct a_guard {  // (discounts still on their way when this call ends -- an error before the launch -- go nowhere)
    a_guard() { g_a_D = 0; }
    ~a_guard() { g_a_D = 0; }
  } a_guard_;
  if (D < 1) return stb_fail("%s: D=%d", who, D);
  if (N < 2 || M < 2) return stb_fail("%s: bounds N=%u M=%u too small", who, N, M);
  if (!a_host || !d_tables || !d_ws || (!vtable && !d_S1)) return stb_fail("%s: null pointer", who);
  if (ws_bytes < fill_workspace_need(N, M, D))
    return stb_fail("%s: workspace %zu < %zu", who, ws_bytes, fill_workspace_need(N, M, D));
  const uint64_t need = vtable ? stb_vtable_elems(N, M) : stb_table_elems(N, M);
  if (D > 1 && (table_stride < need || (!vtable && s1_stride < N))) return stb_fail("%s: strides too small", who);
  if (D > 1 && (table_stride & 1)) return stb_fail("%s: table stride must be even", who);
  for (int d = 0; d < D; d++)
    if (!(a_host[d] >= 0.0 && a_host[d] < 1.0)) return stb_fail("%s: discount %g outside [0,1)", who, a_host[d]);

  fill_args A;
  memset(&A, 0, sizeof(A));
  char *ws = (char *)d_ws;
  A.a = (const double *)ws;
  ws += stb_align_up((size_t)D * sizeof(double), 256);
  A.W = frontier_pitch(M);
  A.fm = (double *)ws;
  A.fe = (int *)(ws + (size_t)D * 2 * A.W * sizeof(double));
  ws = (char *)d_ws + ws_head(M, D);
  const size_t ws_left = ws_bytes - ws_head(M, D);
  A.tables = d_tables;
  A.tstride = table_stride;
  A.S1 = d_S1;
  A.s1stride = s1_stride;
  A.N = N;
  A.M = M;
  if (stb_logtab(&A.lt)) return 1;
  if (D <= 64 && stb_env_int("STB_A_BY_VALUE", 1)) {
    stb_a_defer(a_host, D);  // (the launcher's first kernel writes them, or stb_a_flush below)
  } else {
    HIPCHK(hipMemcpyAsync((void *)A.a, a_host, (size_t)D * sizeof(double), hipMemcpyHostToDevice, st));
  }

  if (kind != 0 && !(kind == 2 && (g_fillv_exact || stb_env_int("STB_FILLV_EXACT", 0) || !hb_takes_kind(N, M, D, kind)))) {
    // floats, and the V table from the S recurrence's own cells (one division per cell, off the serial path: 1e-10 of
    // the reference; STB_FILLV_EXACT=1 keeps the V table on the reference's own recurrence, bit for bit)
    if (!hb_takes_kind(N, M, D, kind))
      return stb_fail("%s: no kernel stores %s for N=%u M=%u D=%d (the caller narrows a double table instead)", who,
                      kind == 1 ? "floats" : (kind == 2 ? "V ratios" : "float V ratios"), N, M, D);
    unsigned *hdr = nullptr;
    if (stb_launch_hb(A, D, ws, ws_left, nullptr, &hdr, st, kind)) return 1;
    g_last.hdr = hdr;
    g_last.stamped = true;
    g_last.A = A;
    g_last.D = D;
    g_last.st = st;
    g_last.s_table = !vtable;
    g_last.can_fall_back = false;  // (k_fill_pc stores log S in double only: the caller repeats in another way)
    return 0;
  }
  if (vtable) {
    if (stb_a_flush(A, st)) return 1;
    if (stb_env_int("STB_FILLV_CHAIN", 1) && !stb_shared_gpu()) {
      unsigned *hdr = nullptr;
      if (stb_launch_vchain(A, D, ws, ws_left, &hdr, st)) return 1;
      g_last.hdr = hdr;
      g_last.st = st;
      g_last.s_table = false;
      g_last.can_fall_back = false;
      return 0;
    }
    const int C = 2;
    A.R = stb_env_int("STB_FILL_R", 48);
    if (A.R < 1) A.R = 1;
    if (A.R > 126) A.R = 126;
    A.H = (A.R + C - 1) / C * C;
    A.Wv = 64 * C - A.H;
    return stb_launch_rows(A, D, C, STB_ROWS_VRATIO, st);
  }

  const int form = pick_form(variant, N, M, D);
  if (form != FORM_HB && stb_a_flush(A, st)) return 1;  // (stb_launch_hb hands them to its k_prep)
  if (g_dot_req && form != FORM_CHAIN && form != FORM_CK && form != FORM_HB)
    return stb_fail("%s: the fused evaluation needs the chain or the checkpointed form", who);
  switch (form) {
    case FORM_CHAIN: {
      unsigned *hdr = nullptr;
      if (stb_launch_chain(A, D, ws, ws_left, g_dot_req, &hdr, st)) return 1;
      g_last.hdr = hdr;
      g_last.A = A;
      g_last.D = D;
      g_last.st = st;
      g_last.s_table = true;
      g_last.can_fall_back = (g_dot_req == nullptr);
      return 0;
    }
    case FORM_CK: {
      unsigned *hdr = nullptr;
      if (stb_launch_ck(A, D, ws, ws_left, g_dot_req, &hdr, st)) return 1;
      g_last.hdr = hdr;
      g_last.A = A;
      g_last.D = D;
      g_last.st = st;
      g_last.s_table = true;
      g_last.can_fall_back = (g_dot_req == nullptr);
      return 0;
    }
    case FORM_HB: {
      unsigned *hdr = nullptr;
      if (g_dot_req && g_dot_req->col0 == 4) {  // (the walking waves sum their strips' listed cells themselves: grid_hb.hip)
        if (stb_launch_grid(A, D, ws, ws_left, g_dot_req, &hdr, st)) return 1;
      } else if (stb_launch_hb(A, D, ws, ws_left, g_dot_req, &hdr, st)) return 1;
      g_last.hdr = hdr;
      g_last.stamped = !(g_dot_req && g_dot_req->col0 == 4);
      g_last.A = A;
      g_last.D = D;
      g_last.st = st;
      g_last.s_table = true;
      g_last.can_fall_back = (g_dot_req == nullptr);
      return 0;
    }
    case FORM_PC:
      pc_geometry(A);
      return stb_launch_pc(A, D, st);
    case FORM_ROWS_LOG: {
      const bool few = (uint64_t)D * M < 40000;
      int C = stb_env_int("STB_FILL_C", few ? 1 : 2);
      if (C != 1 && C != 2 && C != 4) return stb_fail("STB_FILL_C must be 1, 2 or 4");
      A.R = stb_env_int("STB_FILL_R", few ? 48 : 64);
      if (A.R < 1) A.R = 1;
      A.H = (A.R + C - 1) / C * C;
      if (A.H > 64 * C - C) return stb_fail("STB_FILL_R=%d too large for C=%d", A.R, C);
      A.Wv = 64 * C - A.H;
      return stb_launch_rows(A, D, C, STB_ROWS_LOGDOM, st);
    }
    default: {
      if (!stb_ablation_fill)
        return stb_fail("%s: fill variant %d is one of the superseded forms, which this build does not carry "
                        "(make -C tools/ablation builds a library that does)", who, variant);
      unsigned *hdr = nullptr;
      if (stb_ablation_fill(A, D, variant, ws, ws_left, &hdr, st)) return 1;
      g_last.hdr = hdr;
      g_last.st = st;
      g_last.s_table = true;
      g_last.can_fall_back = false;
      return 0;
    }
  }
}

extern "C" int stb_fill_S(const double *a_host, int D, unsigned N, unsigned M, double *d_tables, uint64_t table_stride,
                          double *d_S1, uint64_t s1_stride, void *d_ws, size_t ws_bytes, int variant, void *stream) {
  STB_ENTRY;
  return fill_common(a_host, D, N, M, d_tables, table_stride, d_S1, s1_stride, d_ws, ws_bytes, variant, 0,
                     (hipStream_t)stream);
}

// S_FLOAT storage written once, as floats (lib/stable.c:389-449: the recurrence in double, the stored value a float):
// D float slabs with the double slab's element offsets.  Fails (and says so) where only the double forms apply:
// stb_fill_takes_kind(N, M, D, 1) tells beforehand.
extern "C" int stb_fill_Sf(const double *a_host, int D, unsigned N, unsigned M, float *d_tables, uint64_t table_stride,
                           double *d_S1, uint64_t s1_stride, void *d_ws, size_t ws_bytes, void *stream) {
  STB_ENTRY;
  return fill_common(a_host, D, N, M, reinterpret_cast<double *>(d_tables), table_stride, d_S1, s1_stride, d_ws, ws_bytes,
                     STB_FILL_SCALED, 1, (hipStream_t)stream);
}

extern "C" int stb_fill_Vf(const double *a_host, int D, unsigned N, unsigned M, float *d_vtables, uint64_t vtable_stride,
                           void *d_ws, size_t ws_bytes, void *stream) {
  STB_ENTRY;
  return fill_common(a_host, D, N, M, reinterpret_cast<double *>(d_vtables), vtable_stride, nullptr, 0, d_ws, ws_bytes,
                     STB_FILL_SCALED, 3, (hipStream_t)stream);
}

extern "C" int stb_fill_V(const double *a_host, int D, unsigned N, unsigned M, double *d_vtables, uint64_t vtable_stride,
                          void *d_ws, size_t ws_bytes, void *stream) {
  STB_ENTRY;
  return fill_common(a_host, D, N, M, d_vtables, vtable_stride, nullptr, 0, d_ws, ws_bytes, STB_FILL_SCALED, 2,
                     (hipStream_t)stream);
}

// the V table on the reference's own recurrence (k_fillv_chain: two dependent divisions a row, bit for bit the
// reference's table), whatever the size: what stb_fill_V does by itself below 512 rows and under STB_FILLV_EXACT=1
extern "C" int stb_fill_V_exact(const double *a_host, int D, unsigned N, unsigned M, double *d_vtables, uint64_t vtable_stride,
                                void *d_ws, size_t ws_bytes, void *stream) {
  STB_ENTRY;
  g_fillv_exact = true;
  const int rc = fill_common(a_host, D, N, M, d_vtables, vtable_stride, nullptr, 0, d_ws, ws_bytes, STB_FILL_SCALED, 2,
                             (hipStream_t)stream);
  g_fillv_exact = false;
  return rc;
}

/*
 * stb_hip.h -- additive C ABI of libstb_amd for device-resident / batched use of the hot path.
 *
 * The reference (wbuntine/libstb) has no such interface: its only API is the scalar host one in
 * stable.h / psample.h, which this library also exports unchanged (see those headers).  The entry
 * points below expose the same arithmetic -- the table fill of S_remake_part
 * (reference lib/stable.c:321-388), the S_S gather-sum and restaurant terms of aterms
 * (lib/samplea.c:46-83) and the lgamma sum of bterms (lib/sampleb.c:33-41) -- for callers that keep
 * tables and (n,t) groups resident in HBM and evaluate many discounts at once (SURVEY 8b "new,
 * additive C ABI").  Plain pointers and sizes only; every device pointer is a hipMalloc'd (or
 * torch-allocated) address on the current device; `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  All functions return 0 on success, non-zero on failure with the
 * message available from stb_last_error(); nothing here falls back to the CPU.
 */
#ifndef STB_HIP_H
#define STB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnostics ---- */
const char *stb_last_error(void);
int stb_device_count(void);                 /* 0 when no usable GPU */
int stb_device_name(char *buf, int len);    /* name + gcnArch of the current device */

/* ---- which GPU.  The reference is a host library and has no such notion; SURVEY 5 asks for a
 * device-id knob.  A thread picks the GPU for the objects it creates (S_make, stb_groups_create) with
 * stb_set_device(k); without it STB_DEVICE=k in the environment decides; without either the HIP
 * runtime's current device is used.  Tables and group sets remember their device: every later call
 * on them (S_remake, growth inside S_S / S_V, S_free, stb_groups_aterms ...) switches to it and puts
 * the caller's current device back.  stb_device_enter / stb_device_leave are that switch, for
 * callers of the raw-pointer entry points below (which run on the current device). ---- */
int stb_set_device(int dev);                /* non-zero if dev is not a usable device */
int stb_get_device(void);                   /* the device the next S_make / stb_groups_create will use */
int stb_device_enter(int dev);              /* make dev current; returns what to pass to stb_device_leave */
void stb_device_leave(int prev);

/* ---- memory and stream helpers, so that C / FFI callers need no HIP headers ---- */
void *stb_device_malloc(size_t bytes);                 /* hipMalloc; NULL on failure */
void stb_device_free(void *p);
void *stb_host_malloc(size_t bytes);                   /* pinned (hipHostMalloc); NULL on failure */
void stb_host_free(void *p);
void *stb_host_malloc_huge(size_t bytes);              /* pinned AND on 2 MB pages (aligned_alloc + madvise + hipHostRegister); NULL on failure */
void stb_host_free_huge(void *p);
int stb_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int stb_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int stb_stream_sync(void *stream);
void *stb_stream_create(void);                         /* a stream that does not synchronise with the null stream; NULL on failure */
void stb_stream_destroy(void *stream);
void *stb_event_create(void);
void stb_event_destroy(void *ev);
int stb_event_record(void *ev, void *stream);
int stb_event_wait(void *ev);                          /* blocks the calling thread */
int stb_event_done(void *ev);                          /* 1 done, 0 not yet, -1 error */

/* ---- layout of one table slab (see libstb_amd/csrc/stb_layout.h) ---- */
uint64_t stb_cells(unsigned N, unsigned M);        /* stored values: sum_{n=3..N} min(n-2,M-1) */
uint64_t stb_elems(unsigned N, unsigned M);        /* doubles to allocate (rows start 16B aligned) */
uint64_t stb_rowoff(unsigned n, unsigned M);       /* element offset of row n; value (n,m) at +m-2 */
uint64_t stb_vcells(unsigned N, unsigned M);       /* same three for the V (ratio) table */
uint64_t stb_velems(unsigned N, unsigned M);
uint64_t stb_vrowoff(unsigned n, unsigned M);

/* growth policy of the table object (reference lib/stable.c:564-630, S_extend): given current and
 * maximum bounds and the (n+1, m+1) an accessor asks for, the bounds the table grows to.  Pure
 * integer function; S_S / S_V use it internally, exported so callers can pre-size. */
void stb_extend_policy(unsigned usedN, unsigned usedM, unsigned maxN, unsigned maxM, int N, int M,
                       unsigned *newN, unsigned *newM);

/* ---- K1/K2: fill D log-Stirling tables (D=1: S_remake; D>1: the batched-discount mode) ----
 * a_host[D]            discounts (host memory), 0 <= a < 1 (a = 0: the unsigned Stirling numbers of the first kind)
 * d_tables             D slabs of stb_elems(N,M) doubles, slab d at d_tables + d*table_stride
 * d_S1                 D vectors of N doubles (S1[n-1] = log S^n_{1,a}), vector d at + d*s1_stride
 * d_ws / ws_bytes      scratch of at least stb_fill_workspace_bytes(N,M,D)
 * variant              STB_FILL_SCALED (default), STB_FILL_LOGDOMAIN or STB_FILL_SCALED_STEP
 */
#define STB_FILL_SCALED 0      /* linear-domain recurrence, block-floating cells, table log; picks HB, CHAIN or PC by size */
#define STB_FILL_LOGDOMAIN 1   /* logadd(log(.)+., .) per cell, operation order of lib/stable.c:380-388 */
#define STB_FILL_SCALED_STEP 2 /* (ablation build) linear-domain, renormalised every row, libm-grade log */
#define STB_FILL_SPLIT 3       /* (ablation build) recurrence kernel + in-place log conversion on auxiliary streams */
#define STB_FILL_FUSED 4       /* (ablation build) recurrence and log in one kernel, launched per row block */
#define STB_FILL_PC 5          /* one producer wave (recurrence) + consumer waves (logs) per column block, via LDS */
#define STB_FILL_CHAIN 6       /* one launch: column blocks keep their columns for all rows, edges handed on in HBM */
#define STB_FILL_CHAINX 7      /* (ablation build) the chain alone in its blocks; logs by converter blocks (D <= 2) */
#define STB_FILL_CK 8          /* one launch: a recurrence-only spine publishes edges and checkpoints, tile workers convert and store */
#define STB_FILL_HB 9          /* one launch: a spine that walks blocks of rows behind a halo, alone; tile workers convert and store */
size_t stb_fill_workspace_bytes(unsigned N, unsigned M, int D);
int stb_default_variant(void); /* STB_FILL_SCALED unless the environment says STB_FILL_VARIANT=1 */
int stb_fill_S(const double *a_host, int D, unsigned N, unsigned M, double *d_tables,
               uint64_t table_stride, double *d_S1, uint64_t s1_stride, void *d_ws, size_t ws_bytes,
               int variant, void *stream);
/* what stb_fill_S will use for these sizes: columns per lane, rows per launch, kernel launches;
 * returns the form: 2 producer/consumer, 3 chain, 4 checkpointed (spine + tile workers), 5 another,
 * 6 halo blocks (spine that walks blocks of rows alone + tile workers) */
int stb_fill_tuning(unsigned N, unsigned M, int D, int *C_out, int *R_out, int *launches);
/* Completion status of the last one-launch fill (halo-block, chain, checkpointed) issued by THIS thread (waits
 * for it): 0, or non-zero with stb_last_error() set when a workgroup gave up waiting for its neighbour (the
 * fill's polls are bounded; STB_CHAIN_TIMEOUT_MS, default 2000).  The other forms cannot fail on the device. */
int stb_fill_status(void);
/* A one-launch stb_fill_S whose wait expired is repeated by stb_fill_status with the
 * producer/consumer form (no waits between workgroups) before it returns 0; this counts how often
 * that happened on this thread.  STB_CHAIN_NO_FALLBACK=1 turns the repeat off (status then fails). */
unsigned stb_fill_fallbacks(void);
/* A GPU shared with other processes.  The one-launch forms have workgroups that wait for each other; when other
 * processes hold the compute units those waits burn scheduler quanta (a 1.4 ms evaluation was measured at 725 ms with
 * four processes on one GPU).  Every such launch is timed on the device; stb_slow_launches() counts those that took more
 * than 20 x what their geometry should (the first one is reported through yaps_message).  After two of them -- or from
 * the start with STB_SHARED_GPU=1 in the environment / stb_set_shared_gpu(1) -- fills take the producer/consumer form and
 * evaluations go through stored tables: no waits between workgroups, results within 1e-10 as ever.  STB_SHARED_GPU=0 /
 * stb_set_shared_gpu(0) never switches; stb_set_shared_gpu(-1) is the automatic rule again. */
unsigned stb_slow_launches(void);
int stb_shared_gpu_mode(void);           /* 1: the forms without waits are being taken */
void stb_set_shared_gpu(int mode);
void stb_note_launch_span(double span_ms, double expect_ms);   /* (what the launches report; for tests of the rule) */
/* 1 when the library carries the superseded fill forms (STB_FILL_SCALED_STEP / _SPLIT / _FUSED /
 * _CHAINX; tools/ablation); the default build refuses those variants with a message */
int stb_has_ablation(void);
/* release the device buffers the library keeps for reuse between stb_groups_create / samplea calls
 * (capped at STB_POOL_MB, default 4096) */
void stb_pool_trim(void);
/* kernel-only timing of the fills issued by THIS thread between begin and end (the stream must be
 * synchronised before _end): sum of the per-launch device durations in ms and their count */
void stb_fill_profile_begin(void);
int stb_fill_profile_end(double *kernel_ms_total, int *launches);
/* device time in ms from the first of those launches' start to the last one's end (large batches of
 * the producer/consumer form run two sub-batches side by side, so the sum counts that time twice) */
double stb_fill_profile_span(void);
/* V tables (next row 8f-1): V^n_m = S^n_m / S^n_{m-1} for 2<=n<=N, 2<=m<=min(n,M); lib/stable.c:451-482.  Tables of 512 rows
 * or more are taken from the S recurrence's block-floating cells (one division per cell off the serial path; within
 * 1e-10 of the reference); smaller ones, and every table under STB_FILLV_EXACT=1, walk the reference's own V
 * recurrence, bit for bit. */
int stb_fill_V(const double *a_host, int D, unsigned N, unsigned M, double *d_vtables,
               uint64_t vtable_stride, void *d_ws, size_t ws_bytes, void *stream);
int stb_fill_V_exact(const double *a_host, int D, unsigned N, unsigned M, double *d_vtables,
                     uint64_t vtable_stride, void *d_ws, size_t ws_bytes, void *stream);

/* the same tables written once as floats (S_FLOAT, reference lib/stable.c:389-449 and :483-537: all arithmetic in
 * double, only the stored value is a float): D float slabs with the double slabs' element offsets, no double slab
 * anywhere.  Only the halo-block form narrows before the store: stb_fill_takes_kind(N, M, D, kind) says whether a fill
 * of kind 1 (log S as float), 2 (V as double, taken from the S recurrence's cells) or 3 (V as float) applies to these
 * sizes; where it does not, stb_fill_Sf / stb_fill_Vf fail and the caller narrows a double table (stb_table_to_float). */
int stb_fill_takes_kind(unsigned N, unsigned M, int D, int kind);
int stb_fill_Sf(const double *a_host, int D, unsigned N, unsigned M, float *d_tables, uint64_t table_stride,
                double *d_S1, uint64_t s1_stride, void *d_ws, size_t ws_bytes, void *stream);
int stb_fill_Vf(const double *a_host, int D, unsigned N, unsigned M, float *d_vtables, uint64_t vtable_stride,
                void *d_ws, size_t ws_bytes, void *stream);

/* narrow a table slab to float, element for element (S_FLOAT storage, reference lib/stable.h:31-33:
 * "keep final table values in float, but all intermediate calcs done in double") */
int stb_table_to_float(const double *d_src, float *d_dst, uint64_t elems, void *stream);

/* ---- lookups with S_S semantics (lib/stable.c:941-974, no growth): out[g] = S_S(n[g], t[g]) ---- */
int stb_lookup_S(const double *d_table, const double *d_S1, unsigned N, unsigned M,
                 const uint32_t *d_n, const uint32_t *d_m, uint64_t G, double *d_out, void *stream);

/* ... and of the ratio table (a slab filled by stb_fill_V): out[g] = S_V / S_U / S_UV (n[g], m[g]) with the reference's
 * identities and bounds (lib/stable.c:875-939; no growth, no asymptotic branch: outside the slab's bounds S_V is 0 as in
 * :922).  What the table-indicator sampling step either side of the path reads (test/demo.c:405-445). */
int stb_lookup_V(const double *d_vtable, unsigned N, unsigned M, const uint32_t *d_n, const uint32_t *d_m, uint64_t G,
                 double *d_out, void *stream);
int stb_lookup_U(const double *d_vtable, unsigned N, unsigned M, double a, const uint32_t *d_n, const uint32_t *d_m,
                 uint64_t G, double *d_out, void *stream);
int stb_lookup_UV(const double *d_vtable, unsigned N, unsigned M, double a, const uint32_t *d_n, const uint32_t *d_m,
                  uint64_t G, double *d_out, void *stream);

/* ---- K3: sweep.  out[d] = sum over pairs g with n[g]>1 of S_S_d(n[g], t[g])  (samplea.c:68-80) ---- */
size_t stb_sweep_workspace_bytes(uint64_t G, int D);
int stb_sweep_S(const double *d_tables, uint64_t table_stride, const double *d_S1,
                uint64_t s1_stride, int D, unsigned N, unsigned M, const uint32_t *d_n,
                const uint16_t *d_t, uint64_t G, double *d_out, void *d_ws, size_t ws_bytes,
                void *stream);

/* ---- K4: per-restaurant terms.
 * restaurant: out[d] = sum_i T_i*log(x_d) + lgamma(T_i + b_i/x_d) - lgamma(b_i/x_d)  (samplea.c:65-67)
 * bterms:     out[j] = -Q*x_j + (shape-1)*log(x_j) + sum_i (lgamma(T_i + x_j/apar) - lgamma(x_j/apar))
 *                                                                            (sampleb.c:33-41)
 * x_host[D] are host doubles; d_T (uint32) and d_bpar (double) device arrays of I entries. */
size_t stb_terms_workspace_bytes(uint64_t I, int D);
int stb_restaurant_terms(const double *x_host, int D, const uint32_t *d_T, const double *d_bpar,
                         uint64_t I, double *d_out, void *d_ws, size_t ws_bytes, void *stream);
int stb_bterms(const double *x_host, int J, double Q, double shape, double apar,
               const uint32_t *d_T, uint64_t I, double *d_out, void *d_ws, size_t ws_bytes,
               void *stream);
/* the same for a host sampler that evaluates bterms many times over one T[] (sampleb's ARMS / slice callback,
 * lib/sampleb.c:33-41): T resident, an evaluation of J <= 64 abscissae is two kernel launches and ONE wait, the
 * values arriving in pinned host memory without a copy call */
typedef struct stb_bctx stb_bctx_t;
stb_bctx_t *stb_bterms_create(const uint32_t *T, int I);
int stb_bterms_update(stb_bctx_t *c, const uint32_t *T, int I); /* new totals, I <= the I it was created with; else non-zero */
int stb_bterms_eval(stb_bctx_t *c, const double *x_host, int J, double Q, double shape, double apar, double *out_host);
void stb_bterms_free(stb_bctx_t *c);

/* ---- device-resident group set + grid evaluation (host-friendly wrappers over the above) ----
 * stb_groups_t owns device copies of the flat (n,t) pairs and the per-restaurant T, bpar, plus
 * the scratch needed to evaluate aterms on up to Dmax discounts with table bounds (N,M). */
typedef struct stb_groups stb_groups_t;
stb_groups_t *stb_groups_create(int I, const int *K, const uint32_t *T, const uint32_t *nflat,
                                const uint16_t *tflat, const double *bpar, unsigned N, unsigned M,
                                int Dmax);
void stb_groups_free(stb_groups_t *g);
/* out_host[D] = aterms(x_d) for every d: table build + sweep + restaurant terms.  D <= Dmax. */
int stb_groups_aterms(stb_groups_t *g, const double *x_host, int D, double *out_host);
/* The same evaluation in two calls, so that ONE host thread can keep several GPUs (or several group sets)
 * busy: _async queues it on the set's own stream -- behind whatever `stream` (a hipStream_t, or NULL)
 * holds at this moment -- and returns without waiting; stb_groups_wait blocks until it is through and
 * only then writes out_host[0..D-1].  x_host may be reused at once, out_host must stay valid until the
 * wait; one evaluation per set at a time.  (All entry points of this library may be called from several
 * host threads at once -- one per GPU, or several on one GPU; see INTEGRATION.md.) */
int stb_groups_aterms_async(stb_groups_t *g, const double *x_host, int D, double *out_host, void *stream);
int stb_groups_wait(stb_groups_t *g);
/* The node from ONE host thread, for a C caller (the reference's callers are C: lib/samplea.c:155, test/demo.c:478-480).
 * `sets` are k group sets made from the same pairs -- one per GPU: stb_groups_create_node makes min(ndev,
 * stb_device_count()) of them on devices 0, 1, ... and returns how many (0 on failure); or the caller's own, e.g. several
 * on one GPU -- and the D abscissae are sharded over them in contiguous blocks whose sizes differ by at most one (each
 * must fit its set's Dmax): all are queued before any is waited for, and out_host[0..D) comes back in the grid's order,
 * 8 bytes a discount through pinned host memory.  This is the discount-axis sharding of SURVEY 8e without a collective;
 * the one-process-per-GPU layout (libstb_amd/shard.py, bench.py --gpus N) keeps the values on the devices and gathers
 * them over RCCL instead. */
int stb_groups_aterms_multi(stb_groups_t *const *sets, int k, const double *x_host, int D, double *out_host);
int stb_groups_create_node(int ndev, int I, const int *K, const uint32_t *T, const uint32_t *nflat, const uint16_t *tflat,
                           const double *bpar, unsigned N, unsigned M, int Dmax, stb_groups_t **sets_out);
/* ... with the D log-posteriors left on the DEVICE, in d_out[0..D) (a device address), for a caller that hands them to a
 * collective: the discount axis sharded over the GPUs of a node, every rank all-gathers its share.  Queued
 * like _async; `stream` then waits on the device for the values, so work queued on it afterwards sees them.
 * stb_groups_wait(g) must still follow, before the values are trusted: it reports a table walk that gave up waiting
 * for a neighbour, in which case it re-evaluates through stored tables and rewrites d_out. */
int stb_groups_aterms_device(stb_groups_t *g, const double *x_host, int D, double *d_out, void *stream);
/* the same values through stored tables and the sorted gather whatever D is (stb_groups_aterms sums
 * inside the fill when D >= 2, which needs a set-up pass over the pairs on first use) */
int stb_groups_aterms_tables(stb_groups_t *g, const double *x_host, int D, double *out_host);
/* new per-restaurant totals T[I] and concentrations bpar[I] for the same pairs */
int stb_groups_update_restaurants(stb_groups_t *g, const uint32_t *T, const double *bpar);
/* NEW PAIRS for a set of the same shape (I restaurants, G = sum K pairs): what a caller whose counts change between
 * calls does instead of stb_groups_free + stb_groups_create -- the reference's own Gibbs loop rewrites t[j][i] and T[j]
 * in every iteration (test/demo.c:405-445) and then resamples a (:478-480).  No allocation, no sort: the pairs are
 * copied once into pinned memory, piece by piece, each piece on its way to the device while the caller hands over the
 * next; the cell lists of the fused evaluation are rebuilt on the device when the next evaluation is queued.
 *   stb_groups_pairs_begin(g)                       waits for whatever still uses the old pairs
 *   stb_groups_pairs_put(g, n, t, count, &mn, &mt)  the next `count` pairs, in order (e.g. restaurant after restaurant:
 *                                                   n[i], t[i], K[i]); mn / mt (may be NULL) receive the largest n and t
 *                                                   so far -- what samplea derives its table bounds from
 *   stb_groups_pairs_commit(g, T, bpar, N, M)       all G pairs are in; T, bpar (both or neither NULL) new restaurant
 *                                                   totals; N, M the table bounds (0, 0: unchanged).  New bounds
 *                                                   re-size what depends on them (buffers come from the library's cache)
 * stb_groups_update_pairs = begin + put + commit(NULL, NULL, 0, 0) from flat arrays.  A set may be created EMPTY
 * (stb_groups_create with nflat = tflat = NULL, and N = M = 0 when the bounds are not known yet) and filled this way.
 * Results are the same bits as those of a set created from the same pairs, in whatever order they are handed over. */
int stb_groups_pairs_begin(stb_groups_t *g);
int stb_groups_pairs_put(stb_groups_t *g, const uint32_t *n, const uint16_t *t, uint64_t count, unsigned *maxn, unsigned *maxt);
/* all G pairs at once from samplea's ragged arrays (restaurant i: n[i][0..K[i]), t[i][0..K[i])), between begin and
 * commit instead of the puts: a few host threads of the library's own copy (STB_PUT_THREADS, default 4; 0: the caller's
 * thread alone) while the calling thread hands what is copied to the device */
int stb_groups_pairs_put_ragged(stb_groups_t *g, int I, const int *K, uint32_t *const *n, uint16_t *const *t, unsigned *maxn, unsigned *maxt);
int stb_groups_pairs_commit(stb_groups_t *g, const uint32_t *T, const double *bpar, unsigned N, unsigned M);
int stb_groups_update_pairs(stb_groups_t *g, const uint32_t *nflat, const uint16_t *tflat);
/* how often, in this process, a fused evaluation (halo-block, grid or chain form) gave up waiting for a neighbour and
 * was repeated through stored tables (stb_fill_fallbacks counts the repeated FILLS of the calling thread) */
unsigned stb_groups_fallbacks(void);
/* the strip shape the grid form (k_grid_hb) takes for D discounts of an N x M table: columns per lane, rows per group,
 * every K-th row staged (any pointer may be NULL); non-zero where that form does not apply.  Diagnostics. */
int stb_grid_shape(unsigned N, unsigned M, int D, int *C_out, int *G_out, int *K_out);
/* what the set was created with (any pointer may be NULL) */
int stb_groups_shape(const stb_groups_t *g, int *I, uint64_t *G, unsigned *N, unsigned *M, int *Dmax);
/* pieces of the same evaluation, for timing: ms of device time per stage (may be NULL) */
int stb_groups_aterms_timed(stb_groups_t *g, const double *x_host, int D, double *out_host,
                            float *ms_fill, float *ms_sweep, float *ms_terms);

/* ---- aterms2, the S-free discount posterior of samplea2 (lib/samplea.c:85-150) ----
 * For a sampled partition of the customers into tables the posterior needs only how many tables have
 * each size: cnt[s] = number of tables with s customers (s = 2 .. S-1; entries 0 and 1 are ignored),
 * plus the per-restaurant T[I], bpar[I] of aterms.  out[d] = sum_i restaurant terms(x_d)
 * + sum_s cnt[s] * log((1-x_d)(2-x_d)...(s-1-x_d)), evaluated as lib/lgamma.c:36-52 does. */
typedef struct stb_hist stb_hist_t;
stb_hist_t *stb_hist_create(const uint32_t *cnt, unsigned S, int I, const uint32_t *T, const double *bpar);
int stb_hist_aterms2(stb_hist_t *h, const double *x_host, int D, double *out_host);
void stb_hist_free(stb_hist_t *h);
/* the table sizes the most recent samplea2() sampled, in the reference's layout (ALData.m,
 * lib/samplea.c:283-320): returns the number of entries and, through m, the array (uint16) */
size_t stb_samplea2_partition(const uint16_t **m);

/* ---- diagnostics of the host samplers (include/psample.h) ----
 * the log-posterior evaluations of the most recent samplea()/sampleb() call on this process:
 * how many, ARMS' return code (ignored by the samplers themselves, as in the reference), and the
 * i-th (abscissa, value) pair */
/* samplea() keeps, per calling thread, ONE group set as a container (device buffers, pinned staging, stream, count
 * slab) and hands every call's pairs over into it (stb_groups_pairs_*): a call of the same shape allocates nothing.
 * The pairs themselves are NOT assumed to be the last call's: that reuse is opt-in (STB_SAMPLEA_CACHE=1 skips the
 * hand-over when a 128-bit fingerprint of K, n, t and the shapes equals the last call's; INTEGRATION.md section 7 has the
 * failure mode).  This call frees the calling thread's kept set and its device memory. */
void stb_sampler_cache_clear(void);
int stb_sampler_trace_count(void);
int stb_sampler_trace_code(void);
int stb_sampler_trace_get(int i, double *x, double *y);
/* entry i of the ziggurat tables the Gaussian generator rebuilt (which: 0 heights, 1 widths,
 * 2 integer thresholds); for tests */
double stb_zig_table(int which, int i);

#ifdef __cplusplus
}
#endif
#endif

#!/usr/bin/env python3
"""bench.py -- libstb_amd on MI355X: "S-table cells/s + sampler grid-evals/s at N=M=10000" (BASELINE.json).

A "step" is one pass of the hot path over one batch of synthetic input: every rank fills the
log-Stirling tables S^n_{m,a} (N = M = 10000) of ITS discounts through the C ABI (stb_fill_S, the
device form of the reference's S_make/S_remake, lib/stable.c:321-388), then the per-discount probe
scalars are all-gathered over RCCL (the path's only exchange: 8 bytes per discount).  Per-GPU work
is fixed as N grows (weak scaling); value = cells filled by all ranks per second of the slowest rank.

    python bench.py --gpus 1 --steps 20 --warmup 3          # configs[1]: single discount a=0.5
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8   # one discount per GPU
    python bench.py --discounts-per-gpu 8                    # the batched mode (configs[2] shape)

One JSON line on stdout (rank 0):
  value / roofline   configs[1]; the roofline prices the fill kernel against HBM (8 B per stored cell,
                     SURVEY 8d), kernel time from per-launch HIP start/stop events inside bench.py
  roofline_sweep     the second half of the metric, as the product runs it for a grid of discounts: the
                     fused evaluation (the fill sums count * log S for the occurring cells itself, no
                     table stored) of 64 discounts x 10^6 pairs at N=M=10000, (8 + 6/D) B per grid-eval;
                     the two-pass form (stored tables + gather) and the CPU sweep are in extra
  scale_job          at EVERY --gpus N: the north-star job -- the 64-discount grid (0.05,0.95) sharded
                     64/N per rank: batched fill, and the fused grid aterms over 10^6 pairs whose 64
                     log-posteriors are all-gathered over RCCL; >= 50 steps each, medians of the first
                     and of the last five (the part slows under sustained load).  A 1 -> 8 GPU curve of
                     scale_job.fill.ms / scale_job.grid_aterms.ms is the strong-scaling speed-up the
                     north star asks for (`value` is weak scaling of one table per GPU and is not).
  cpu_baseline       the reference's (or the oracle's) fill of the same table on this box's host CPU
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from libstb_amd import capi, shard, synth

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
TRAFFIC_DB = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_hbm_traffic.json") for r in (6, 5, 4, 3)) if os.path.exists(p)),
                  os.path.join(ROOT, "profiles", "r06_hbm_traffic.json"))
PROFILE_ROUND = next((r for r in (6, 5, 4) if os.path.exists(os.path.join(ROOT, "profiles", f"r0{r}_sq_counters_fill1.txt"))), 4)
N_SIMD = 1024           # 256 compute units x 4 SIMDs (MI355X)
NOMINAL_CLOCK_HZ = 2.4e9
FORM_NAMES = {2: "pc", 3: "chain", 4: "ck", 6: "hb"}


def cpu_baseline(N: int, M: int, a: float):
    """Single-core fill of the same (N,M,a) table on the host: the compiled reference when
    oracle/_ref is present, else the oracle's restatement (both test infrastructure)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    cells = synth.cells(N, M)
    if orc.have_ref():
        R = orc.ref()
        sp = R.S_make(N, M, N, M, a, 1)
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            R.S_remake(sp, a)
            best = min(best, time.perf_counter() - t0)
        R.S_free(sp)
        kind = "reference"
        what = f"S_remake of the full N={N} M={M} a={a} table, best of 3 (oracle/_ref/libstb_ref.so)"
        # the batched case on the host cores this box gives a one-GPU job (16 of them),
        # one table per thread (tables of different discounts are independent; ctypes releases the GIL)
        try:
            import threading

            share = len(os.sched_getaffinity(0))
            nthr = max(1, min(16, share))
            grid = synth.discount_grid(64)[:nthr]
            sps = [None] * nthr

            def make(i):
                sps[i] = R.S_make(N, M, N, M, float(grid[i]), 1)

            def remake(i):
                R.S_remake(sps[i], float(grid[i]))

            for fn in (make, remake):
                th = [threading.Thread(target=fn, args=(i,)) for i in range(nthr)]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                wall = time.perf_counter() - t0
            for sp2 in sps:
                if sp2:
                    R.S_free(sp2)
            many = {"value": cells * nthr / wall, "unit": "cells/s", "cores": nthr, "tables": nthr, "seconds": wall, "share_of_host": f"{nthr} of {os.cpu_count()} cpus: the pool gives a one-GPU job 16",
                    "cpus_in_affinity_mask": share,
                    "sample": f"{nthr} threads (the one-GPU job's share of the host), one S_remake of an N={N} M={M} table each, concurrently"}
        except Exception as e:  # never take the contract line down
            many = {"error": repr(e)}
    else:
        L = orc.oracle()
        S1 = np.zeros(N)
        tab = np.zeros(cells)
        L.orc_fill_S(a, N, M, orc.dp(S1), orc.dp(tab))  # touch pages
        best = L.orc_time_fill(a, N, M, 3, orc.dp(S1), orc.dp(tab))
        kind = "port"
        what = f"orc_fill_S of the full N={N} M={M} a={a} table, best of 3 (oracle/liboracle.so)"
        many = None
    out = {"value": cells / best, "unit": "cells/s", "cores": 1, "kind": kind, "sample": what,
           "seconds": best, "host_cpus": os.cpu_count()}
    if many is not None:
        out["multi_core_job_share"] = many  # (SURVEY 8d-ii asks for all host cores on the 64-discount batch; a one-GPU job on this pool has 16)
    return out


def dropin_section(N: int, M: int, a: float, probes: int = 1000000):
    """SURVEY 8d metric 1 "with D2H mirror": what a caller of the reference's OWN interface pays -- S_remake of the table
    (device fill) and then host look-ups through the pinned mirror, as the reference's Gibbs sweep does after every
    S_remake (test/demo.c:427-428 reads S_V per customer, :487 rebuilds).  `probes` random (n, m) S_S calls in a C loop
    (stb_table_probe), the mirror in its three modes, beside the reference's same loop on one host core; then a pass
    over EVERY row in order (the full-table touch)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    rng = np.random.default_rng(20261005)
    n = rng.integers(3, N + 1, probes).astype(np.uint32)
    m = (2 + (rng.random(probes) * (np.minimum(n - 1, M) - 1))).astype(np.uint32)
    rows_n = np.arange(3, N + 1, dtype=np.uint32)              # one cell of every row, top to bottom
    rows_m = np.minimum(rows_n - 1, M).astype(np.uint32)
    u32p = C.POINTER(C.c_uint)
    res = {"N": N, "M": M, "probes": probes, "table_MB": 8.0 * synth.cells(N, M) / 1e6}
    L = capi.lib()
    outv = np.zeros(probes)
    for mode in ("ahead", "lazy", "eager", "ahead_small_pages"):
        old = os.environ.get("STB_MIRROR")
        if mode.startswith("ahead"):
            os.environ.pop("STB_MIRROR", None)
        else:
            os.environ["STB_MIRROR"] = mode
        os.environ.pop("STB_MIRROR_PAGES", None)
        if mode == "ahead_small_pages":   # (hipHostMalloc's memory instead of 2 MB pages: the same on some boxes, 40 % slower look-ups on others)
            os.environ["STB_MIRROR_PAGES"] = "small"
        try:
            t = capi.Table(N, M, N, M, a, capi.S_STABLE)
            r = {}
            for what, nn, mm in (("random", n, m), ("every_row", rows_n, rows_m)):
                best = None
                for rep in range(3):
                    a2 = a + 0.01 * (rep + 1)
                    t0 = time.perf_counter()
                    t.remake(a2)
                    t1 = time.perf_counter()
                    L.stb_table_probe(t.sp, 0, nn.ctypes.data_as(u32p), mm.ctypes.data_as(u32p), len(nn), capi.dp(outv[: len(nn)] if len(nn) < probes else outv))
                    t2 = time.perf_counter()
                    cur = {"remake_ms": (t1 - t0) * 1e3, "lookups_ms": (t2 - t1) * 1e3, "total_ms": (t2 - t0) * 1e3}
                    if best is None or cur["total_ms"] < best["total_ms"]:
                        best = cur
                best["mirror_blocks"] = t.mirrored()[0]
                r[what] = best
            res[mode] = r
            if mode == "ahead":
                keep = outv.copy()
                keep_a = a + 0.03
            t.free()
        finally:
            os.environ.pop("STB_MIRROR_PAGES", None)
            if old is None:
                os.environ.pop("STB_MIRROR", None)
            else:
                os.environ["STB_MIRROR"] = old
    if orc.have_ref():
        R = orc.ref()
        sp = R.S_make(N, M, N, M, a, 1)
        r = {}
        ref_out = np.zeros(probes)
        for what, nn, mm in (("random", n, m), ("every_row", rows_n, rows_m)):
            t0 = time.perf_counter()
            R.S_remake(sp, a + 0.03)
            t1 = time.perf_counter()
            R.ref_probe(sp, 0, nn.ctypes.data_as(u32p), mm.ctypes.data_as(u32p), len(nn), orc.dp(ref_out))
            t2 = time.perf_counter()
            r[what] = {"remake_ms": (t1 - t0) * 1e3, "lookups_ms": (t2 - t1) * 1e3, "total_ms": (t2 - t0) * 1e3}
        R.S_free(sp)
        res["cpu_reference_1core"] = r
    res["speedup_every_row_lazy_over_ahead"] = res["lazy"]["every_row"]["total_ms"] / res["ahead"]["every_row"]["total_ms"]
    res["what"] = ("S_remake + host look-ups through the reference's own interface (S_S in a C loop); ahead: a miss copies its block and sends "
                   "the following >= 16 MB on their way asynchronously (default); lazy: every 128-row block a synchronous copy on first touch "
                   "(rounds 1-5); eager: the whole mirror copied inside S_remake; best of 3, ms")
    return res


def cpu_sweep_baseline(g, N: int, M: int):
    """the sampler sweep on the host -- aterms' gather-sum and restaurant terms over the same
    10^6 pairs against a table already filled (lib/samplea.c:62-80 without the rebuild), on one core and on
    the one-GPU job's share of the cores with one discount's table per thread (the oracle's restatement)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import threading

    import orc

    L = orc.oracle()
    share = len(os.sched_getaffinity(0))
    nthr = max(1, min(16, share))
    grid = synth.discount_grid(64)[:nthr]
    tabs = [orc.fill_S(float(a), N, M) for a in grid]          # (S1, packed table) per discount

    def one(i):
        S1, tab = tabs[i]
        return L.orc_aterms_sum(float(grid[i]), g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar),
                                orc.dp(tab), orc.dp(S1), N, M)

    best1 = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        one(0)
        best1 = min(best1, time.perf_counter() - t0)
    bestn = 1e30
    for _ in range(3):
        th = [threading.Thread(target=one, args=(i,)) for i in range(nthr)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        bestn = min(bestn, time.perf_counter() - t0)
    return {"kind": "port", "pairs": g.pairs, "N": N, "M": M,
            "one_core": {"grid_evals_per_s": g.pairs / best1, "seconds": best1, "cores": 1},
            "all_cores": {"grid_evals_per_s": g.pairs * nthr / bestn, "seconds": bestn, "cores": nthr, "discounts": nthr},
            "sample": "orc_aterms_sum (gather-sum + restaurant terms, table already filled) over the same pairs, best of 3"}


def traffic_lookup(key: str, kernel_prefix: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py:
    WRITE_SIZE + 2*FETCH_SIZE from separate passes, KiB -> bytes)"""
    if not os.path.exists(TRAFFIC_DB):
        return None, None
    db = json.load(open(TRAFFIC_DB))
    for kname, rec in db.get(key, {}).items():
        if kname.startswith(kernel_prefix):
            return rec["hbm_bytes_per_launch"], f"profiles/{os.path.basename(TRAFFIC_DB)}[{key}][{kname}]"
    return None, None


def valu_busy(workload: str, kernel_prefix: str):
    """fraction of the chip's vector-issue slots a kernel uses (SURVEY 8d asks for it beside the HBM fraction), from the
    COMMITTED profiles of the same workload -- static numbers, NOT measured by this run (PMC counters need rocprofv3):
    SQ_INSTS_VALU wave-instructions (profiles/r0N_sq_counters_<workload>.txt) x 4 cycles / (1024 SIMDs x the kernel's
    average duration (profiles/r0N_<workload>_kernel_stats.csv) x 2.4 GHz); also SALU per VALU instruction and the share
    of LDS cycles lost to bank conflicts"""
    import csv
    cpath = os.path.join(ROOT, "profiles", f"r0{PROFILE_ROUND}_sq_counters_{workload}.txt")
    spath = os.path.join(ROOT, "profiles", f"r0{PROFILE_ROUND}_{workload}_kernel_stats.csv")
    if not (os.path.exists(cpath) and os.path.exists(spath)):
        return None
    vals, on = {}, False
    for line in open(cpath):
        if not line.startswith(" "):
            on = kernel_prefix in line
            continue
        if on:
            parts = line.split()
            vals.setdefault(parts[0], float(parts[1]))
    dur_ns = None
    for r in csv.DictReader(open(spath)):
        if kernel_prefix in r["Name"]:
            dur_ns = float(r["AverageNs"])
            break
    if not vals.get("SQ_INSTS_VALU") or not dur_ns:
        return None
    out = {"valu_busy_frac": vals["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * dur_ns * 1e-9 * NOMINAL_CLOCK_HZ),
           "valu_busy_source": f"STATIC, from committed profiles (not this run): profiles/r0{PROFILE_ROUND}_sq_counters_{workload}.txt, "
                               f"profiles/r0{PROFILE_ROUND}_{workload}_kernel_stats.csv ({kernel_prefix}, {dur_ns / 1e3:.1f} us)"}
    if vals.get("SQ_INSTS_SALU"):
        out["salu_per_valu"] = vals["SQ_INSTS_SALU"] / vals["SQ_INSTS_VALU"]
    if vals.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_frac"] = vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / vals["SQ_LDS_IDX_ACTIVE"]
    return out


def grid_kernel_label(N, M, D):
    """the instantiation of k_grid_hb this run's geometry selects (env switches included), from the library itself"""
    L = capi.lib()
    try:
        C_, G_, K_ = C.c_int(), C.c_int(), C.c_int()
        if L.stb_grid_shape(N, M, D, C.byref(C_), C.byref(G_), C.byref(K_)) == 0:
            return f"k_grid_hb<{C_.value},{G_.value},{K_.value}>"
    except AttributeError:
        pass
    return "k_grid_hb"


def groups_handle(L, g, N, M, Dmax):
    h = L.stb_groups_create(g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p),
                            g.n.ctypes.data_as(capi.c_u32_p), g.t.ctypes.data_as(capi.c_u16_p), capi.dp(g.bpar), N, M, Dmax)
    if not h:
        raise capi.StbError(capi.last_error())
    return h


def grid_bounds(g):
    M = max(int(g.t.max()) + 1, 10)
    return max(int(g.n.max()) + 1, M), M


def timed_aterms(L, h, x, reps=4):
    D = len(x)
    res = np.zeros(D)
    mf, ms, mt = C.c_float(), C.c_float(), C.c_float()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        capi.check(L.stb_groups_aterms_timed(h, capi.dp(x), D, capi.dp(res), C.byref(mf), C.byref(ms), C.byref(mt)))
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]:
            best = (wall, mf.value, ms.value, mt.value)
    return best, res


def sweep_section(n_max: int):
    """sampler grid-evals/s on ONE GPU: 10^6 synthetic (n,t) pairs (I=1000 x K=1000, n < n_max, wide t)
    against N=M~n_max tables; one grid-eval = one (discount, pair) term of the log-posterior.
    D=1 is one aterms() evaluation as samplea's ARMS makes them; D=64 is the batched grid."""
    L = capi.lib()
    g = synth.groups(1000, 1000, n_max, "wide")
    N, M = grid_bounds(g)
    out = {"pairs": g.pairs, "N": N, "M": M,
           "note": "fused: the chain fill sums count*log S itself (no table stored, no second pass); "
                   "two_pass: tables stored, then the sorted gather-sum (fill_ms / sweep_ms / terms_ms are device times)"}
    for label, fused in (("fused", "1"), ("two_pass", "0")):
        os.environ["STB_ATERMS_FUSED"] = fused
        h = groups_handle(L, g, N, M, 64)
        try:
            for D in (1, 64):
                x = np.ascontiguousarray(synth.discount_grid(64)[:D] if D > 1 else np.array([0.45]))
                (wall, f_ms, s_ms, t_ms), _ = timed_aterms(L, h, x)
                ge = D * g.pairs
                rec = {"grid_evals": ge, "fill_ms": f_ms, "sweep_ms": s_ms, "terms_ms": t_ms, "wall_ms": wall * 1e3,
                       "grid_evals_per_s_end_to_end": ge / wall}
                if label == "two_pass":
                    rec["grid_evals_per_s_sweep_only"] = ge / (s_ms * 1e-3)
                    rec["sweep_algorithmic_GBs"] = ge * (8 + 6.0 / D) / (s_ms * 1e-3) / 1e9
                out.setdefault(f"D{D}", {})[label] = rec
        finally:
            L.stb_groups_free(h)
    os.environ.pop("STB_ATERMS_FUSED", None)
    return out, g


def sampler_calls(g):
    """one whole samplea() / sampleb() call (host ARMS + device posteriors) on the groups, and the
    reference's own samplea on this box's host CPU when oracle/_ref is present"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    L = capi.lib()
    out = {}
    NP = C.POINTER(C.c_uint32) * g.I
    TP = C.POINTER(C.c_uint16) * g.I
    nn, tt = NP(), TP()
    off = 0
    for i in range(g.I):
        nn[i] = C.cast(g.n.ctypes.data + 4 * off, C.POINTER(C.c_uint32))
        tt[i] = C.cast(g.t.ctypes.data + 2 * off, C.POINTER(C.c_uint16))
        off += int(g.K[i])
    def one_samplea():
        orc.seed_libc(777, 12345)
        t0 = time.perf_counter()
        a = L.samplea(0.5, g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p), nn, tt, None,
                      capi.dp(g.bpar), None, 1, 0)
        return time.perf_counter() - t0, a

    # What a real caller sees: the reference's Gibbs loop rewrites its counts between two samplea calls
    # (test/demo.c:405-445, then :478-480), so the pairs are NEW on every call.  Here one customer joins one table
    # between calls (the table bounds stay); the pairs go to the device, the cell lists are rebuilt, then ARMS runs.
    for _ in range(2):
        one_samplea()                       # (the thread's device set and the pinned staging come into being)
    n_keep = g.n.copy()
    fresh = []
    for r in range(7):
        k = 12345 + 977 * r
        g.n[k] += 1
        fresh.append(one_samplea())
    g.n[:] = n_keep
    best, a_new = min(t for t, _ in fresh), fresh[-1][1]
    fresh_median = float(np.median([t for t, _ in fresh]))
    same = [one_samplea() for _ in range(5)]  # the same pairs, handed over anew: the same work
    a_new = same[-1][1]
    evals = L.stb_sampler_trace_count()
    os.environ["STB_SAMPLEA_CACHE"] = "1"     # opt-in: contents reused when a 128-bit fingerprint says so
    try:
        one_samplea()
        kept = [one_samplea() for _ in range(5)]
    finally:
        os.environ.pop("STB_SAMPLEA_CACHE", None)
        L.stb_sampler_cache_clear()
    tb = None
    for _ in range(3):  # (best of three calls, like samplea: the first one makes the thread's device context)
        orc.seed_libc(777, 12345)
        t0 = time.perf_counter()
        b_new = L.sampleb(10.0, g.I, g.shape, g.scale, g.N.ctypes.data_as(capi.c_u32_p), g.T.ctypes.data_as(capi.c_u32_p),
                          0.5, None, 1, 0)
        t1 = time.perf_counter() - t0
        tb = t1 if tb is None else min(tb, t1)
    out["samplea"] = {"seconds": fresh_median, "aterms_evaluations": evals, "a": a_new, "grid_evals_per_s": evals * g.pairs / fresh_median,
                      "what": "samplea_fresh: the (n,t) pairs change between calls (one count moves), as in the reference's Gibbs loop; "
                              "median of 7 calls; upload of 6 MB of pairs + cell lists + ARMS with 8 device evaluations",
                      "seconds_best": best,
                      "seconds_same_pairs_median": float(np.median([t for t, _ in same])),
                      "seconds_cache_hit_median": float(np.median([t for t, _ in kept])),
                      "cache_note": "seconds_cache_hit: STB_SAMPLEA_CACHE=1 (opt-in) and identical pairs -- rounds 3-4 quoted this number"}
    out["sampleb"] = {"seconds": tb, "bterms_evaluations": L.stb_sampler_trace_count(), "b": b_new}
    if orc.have_ref():
        R = orc.ref()
        orc.seed_libc(777, 12345)
        t0 = time.perf_counter()
        a_ref = R.ref_samplea_flat(0.5, g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), 1, 0)
        tr = time.perf_counter() - t0
        ev_ref = R.ref_trace_count()
        out["samplea"]["cpu_reference"] = {"seconds": tr, "cores": 1, "a": a_ref, "aterms_evaluations": ev_ref,
                                           "grid_evals_per_s": ev_ref * g.pairs / tr}
    return out


def speedup_vs_n1(world: int, fill_ms: float, grid_ms: float, fresh_ms: float):
    """scale_job's speed-up against the newest COMMITTED one-GPU line (profiles/r0X_bench_n1.json): ms(1) / ms(N) of the
    same 64-discount job.  Another box, another day: the driver's own curve from back-to-back runs is the one that counts."""
    for r in (6, 5, 4):
        p = os.path.join(ROOT, "profiles", f"r0{r}_bench_n1.json")
        if not os.path.exists(p):
            continue
        try:
            with open(p) as f:
                txt = f.read()
            one = json.loads(txt[txt.index("{"):])["scale_job"]
            base = {"fill": one["fill"]["ms"], "grid_aterms": one["grid_aterms"]["ms"], "grid_aterms_fresh": one.get("grid_aterms_fresh", {}).get("ms")}
        except (ValueError, KeyError, TypeError):
            continue
        now = {"fill": fill_ms, "grid_aterms": grid_ms, "grid_aterms_fresh": fresh_ms}
        return {"n1_source": os.path.relpath(p, ROOT), "ranks": world, "n1_ms": base,
                **{k: (base[k] / now[k] if base[k] and now[k] else None) for k in now}}
    return None


def launcher_command(gpus: int, argv, port: int):
    """the child that `python bench.py --gpus N` (no launcher in the environment) starts: one rank per GPU
    under torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container's host name may not resolve)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(gpus: int, argv) -> int:
    """run the N-rank job as a child process, pass its stdout / stderr through, return its exit code"""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", "4")
    port = int(env.get("MASTER_PORT", 0)) or free_port()
    return subprocess.run(launcher_command(gpus, list(argv), port), env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=10000, help="table rows N")
    ap.add_argument("--m", type=int, default=10000, help="table columns M")
    ap.add_argument("--discounts-per-gpu", type=int, default=1)
    ap.add_argument("--variant", type=int, default=capi.FILL_SCALED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the sweep (grid-evals/s) sections")
    ap.add_argument("--no-batch64", action="store_true", help="skip the 64-discount sharded batch")
    ap.add_argument("--batch-steps", type=int, default=50, help="timed steps of the 64-discount batch (scale_job)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher starts its own N ranks: a FRESH child process
    # (python -m torch.distributed.run ... bench.py <the same arguments>), started before this process
    # has made any GPU call -- never an exec of a process that has touched the GPU -- whose rank 0
    # prints the line; this process only relays the child's output and exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    # STB_BENCH_SHARE_GPU=1 is a rehearsal mode for boxes with one GPU: every rank uses cuda:0 and
    # the scalars travel over gloo.  The driver's multi-GPU runs use one GPU per rank over RCCL.
    share = os.environ.get("STB_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    L = capi.lib()
    capi.check(L.stb_set_device(local))  # the library's objects go where torch's buffers are
    N, M, Dl = args.n, args.m, args.discounts_per_gpu
    Dg = Dl * world

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            if share:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    def must_be_clean(what, fb0):
        """a chain fill that gave up (and was silently refilled) would make the timing meaningless"""
        capi.check(L.stb_fill_status())
        if L.stb_fill_fallbacks() != fb0:
            raise SystemExit(f"bench: {what}: a chain-form fill gave up and fell back to the producer/consumer form")

    # ------------------------------------------------------------------ configs[1] (and --discounts-per-gpu)
    # configs[1] is the single discount a=0.5; any larger job shards the (0.05,0.95) grid
    grid = np.array([0.5]) if Dg == 1 else synth.discount_grid(Dg)
    mine = np.ascontiguousarray(grid[shard.my_slice(Dg, rank, world)])
    T = capi.DeviceTables(N, M, D=Dl, device=dev)
    cells_rank = T.cells * Dl
    probe_idx = torch.tensor([T.rowoff(N) + max(M // 2, 2) - 2], device=dev)
    gathered = torch.empty(Dg, dtype=torch.float64, device=dev)
    fb0 = L.stb_fill_fallbacks()

    # (one rank: the probes ARE the gathered values -- index_select writes them there, nothing is exchanged or copied)
    probes = gathered.view(Dg, 1) if world == 1 else torch.empty((Dl, 1), dtype=torch.float64, device=dev)

    def step():
        T.fill(mine, args.variant)
        torch.index_select(T.tables, 1, probe_idx, out=probes)  # log S^N_{M/2} per discount
        if world > 1:
            shard.gather_scalars(probes.view(-1), Dg, dist, out=gathered)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    must_be_clean("timed steps", fb0)

    # Kernel-only durations for the roofline: the same K steps once more, live, with a HIP
    # start/stop event pair on every fill launch.  Kept out of the timed region above because an
    # event-carrying launch costs ~5 us of host time; the device durations themselves are unaffected
    # (profiles/ agrees).
    # ... and the status of EVERY one of these fills is checked (a fill that gave up and was silently repeated
    # in another form would make a timing meaningless; the timed region above runs the same launches)
    L.stb_fill_profile_begin()
    for _ in range(args.steps):
        step()
        must_be_clean("profiled steps", fb0)
    fence()
    kms, kn = C.c_double(0.0), C.c_int(0)
    capi.check(L.stb_fill_profile_end(C.byref(kms), C.byref(kn)))

    dt = shard.max_over_ranks(dt, dev, dist)
    total_cells = cells_rank * world * args.steps
    value = total_cells / dt
    Cc, Rr, nl = C.c_int(), C.c_int(), C.c_int()
    form = L.stb_fill_tuning(N, M, Dl, C.byref(Cc), C.byref(Rr), C.byref(nl))
    del T
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ the 64-discount batch, sharded
    batch64 = None
    if not args.no_batch64 and 64 % world == 0:
        D64 = 64 // world
        grid64 = synth.discount_grid(64)
        mine64 = np.ascontiguousarray(grid64[shard.my_slice(64, rank, world)])
        T64 = capi.DeviceTables(N, M, D=D64, device=dev)
        pidx = torch.tensor([T64.rowoff(N) + max(M // 2, 2) - 2], device=dev)
        fb0 = L.stb_fill_fallbacks()

        def step64():
            T64.fill(mine64, capi.FILL_SCALED)
            pr = T64.tables.index_select(1, pidx).reshape(-1)
            return shard.gather_scalars(pr, 64, dist)

        def first_last(ms):
            """median of the first five and of the last five steps, slowest rank"""
            k = min(5, len(ms))
            return (shard.max_over_ranks(float(np.median(ms[:k])), dev, dist), shard.max_over_ranks(float(np.median(ms[-k:])), dev, dist))

        step64()
        fence()
        per_step = []
        t0 = time.perf_counter()
        for _ in range(args.batch_steps):
            t1 = time.perf_counter()
            got = step64()
            torch.cuda.synchronize()
            per_step.append((time.perf_counter() - t1) * 1e3)
        fence()
        dt64 = shard.max_over_ranks(time.perf_counter() - t0, dev, dist) / args.batch_steps
        fill_first, fill_last = first_last(per_step)
        # where a step's time goes: a few more steps with a wait between the fill and the gather
        f_ms, g_ms = [], []
        for _ in range(min(10, args.batch_steps)):
            t1 = time.perf_counter()
            T64.fill(mine64, capi.FILL_SCALED)
            pr = T64.tables.index_select(1, pidx).reshape(-1)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            shard.gather_scalars(pr, 64, dist)
            torch.cuda.synchronize()
            f_ms.append((t2 - t1) * 1e3)
            g_ms.append((time.perf_counter() - t2) * 1e3)
        fence()
        fill_only_ms = shard.max_over_ranks(float(np.median(f_ms)), dev, dist)
        fill_gather_ms = shard.max_over_ranks(float(np.median(g_ms)), dev, dist)
        must_be_clean("batch64 fill", fb0)
        L.stb_fill_profile_begin()
        step64()
        fence()
        k64, n64 = C.c_double(0.0), C.c_int(0)
        capi.check(L.stb_fill_profile_end(C.byref(k64), C.byref(n64)))
        k64 = shard.max_over_ranks(k64.value, dev, dist)
        span64 = shard.max_over_ranks(L.stb_fill_profile_span(), dev, dist)
        ranks_seen = int(torch.isfinite(got).sum().item()) // D64
        cells64 = T64.cells * 64
        fT = L.stb_fill_tuning(N, M, D64, None, None, None)
        del T64
        torch.cuda.empty_cache()
        # fused grid aterms: 10^6 pairs with n < N, every rank evaluates its share of the grid
        g = synth.groups(1000, 1000, N, "wide")
        Ng, Mg = grid_bounds(g)
        h = groups_handle(L, g, Ng, Mg, D64)
        post = np.zeros(D64)
        capi.check(L.stb_groups_aterms(h, capi.dp(mine64), D64, capi.dp(post)))  # set-up + warm-up
        # the log-posteriors stay on the device: the gather takes device scalars (no copy to the host and back)
        d_post = torch.empty(D64, dtype=torch.float64, device=dev)
        fence()
        per_step, eval_ms, gather_ms = [], [], []
        t0 = time.perf_counter()
        for _ in range(args.batch_steps):
            t1 = time.perf_counter()
            capi.check(L.stb_groups_aterms_device(h, capi.dp(mine64), D64, d_post.data_ptr(), capi.stream_ptr()))
            capi.check(L.stb_groups_wait(h))
            t2 = time.perf_counter()
            allpost = shard.gather_scalars(d_post, 64, dist)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            per_step.append((t3 - t1) * 1e3)
            eval_ms.append((t2 - t1) * 1e3)
            gather_ms.append((t3 - t2) * 1e3)
        fence()
        dtg = shard.max_over_ranks(time.perf_counter() - t0, dev, dist) / args.batch_steps
        grid_first, grid_last = first_last(per_step)
        grid_eval_ms = shard.max_over_ranks(float(np.median(eval_ms)), dev, dist)
        grid_gather_ms = shard.max_over_ranks(float(np.median(gather_ms)), dev, dist)
        # ... and on NEW pairs (a caller's counts change between two resamples): the pairs go into the existing set
        # (stb_groups_update_pairs: pinned staging, no allocation, no sort), the cell lists are rebuilt on the device,
        # then the rank's share of the grid is evaluated -- per step, everything a resample pays but the gather
        n_keep = g.n.copy()
        fresh_ms, upd_ms, first_ms = [], [], []
        for r in range(9):
            g.n[12345 + 977 * r] += 1
            fence()
            t1 = time.perf_counter()
            capi.check(L.stb_groups_update_pairs(h, g.n.ctypes.data_as(capi.c_u32_p), g.t.ctypes.data_as(capi.c_u16_p)))
            t2 = time.perf_counter()
            capi.check(L.stb_groups_aterms_device(h, capi.dp(mine64), D64, d_post.data_ptr(), capi.stream_ptr()))
            capi.check(L.stb_groups_wait(h))
            t3 = time.perf_counter()
            if r >= 2:  # (the first two bring the staging area and the count slab into being)
                fresh_ms.append((t3 - t1) * 1e3)
                upd_ms.append((t2 - t1) * 1e3)
                first_ms.append((t3 - t2) * 1e3)
        g.n[:] = n_keep
        fresh_grid = {"ms": shard.max_over_ranks(float(np.median(fresh_ms)), dev, dist),
                      "ms_update_pairs": shard.max_over_ranks(float(np.median(upd_ms)), dev, dist),
                      "ms_first_evaluation": shard.max_over_ranks(float(np.median(first_ms)), dev, dist),
                      "fused_fallbacks": int(L.stb_groups_fallbacks()),
                      "what": "grid_aterms_fresh: stb_groups_update_pairs (10^6 new pairs into the kept set) + the first evaluation "
                              "of the rank's discounts (cell lists rebuilt on the device); median of 7, slowest rank"}
        L.stb_groups_free(h)
        batch64 = {
            "discounts_total": 64, "discounts_per_gpu": D64, "ranks": world, "ranks_seen": ranks_seen,
            "steps": args.batch_steps,
            "fill": {"ms": dt64 * 1e3, "ms_first5_median": fill_first, "ms_last5_median": fill_last,
                     "ms_fill_median": fill_only_ms, "ms_gather_median": fill_gather_ms,
                     "cells_per_s": cells64 / dt64, "form": FORM_NAMES.get(fT, str(fT)),
                     "kernel_ms_sum": k64, "kernel_span_ms": span64, "launches": n64.value,
                     "frac_of_hbm_peak_per_gpu": (8.0 * cells64 / world / (span64 * 1e-3) / 1e9 / HBM_PEAK_GBS) if span64 > 0 else None},
            "grid_aterms": {"ms": dtg * 1e3, "ms_first5_median": grid_first, "ms_last5_median": grid_last,
                            "ms_evaluation_median": grid_eval_ms, "ms_gather_median": grid_gather_ms, "pairs": g.pairs, "N": Ng, "M": Mg, "grid_evals_per_s": 64 * g.pairs / dtg,
                            "log_posteriors_finite": int(torch.isfinite(allpost).sum().item()),
                            "log_posterior_d0_d63": [float(allpost[0]), float(allpost[63])],
                            "what": "steady state: the set's pairs unchanged between steps (lists built once)"},
            "grid_aterms_fresh": fresh_grid,
            "speedup_vs_n1": speedup_vs_n1(world, dt64 * 1e3, dtg * 1e3, fresh_grid["ms"]),
            "note": "strong scaling: the same 64 tables / 64 x 10^6 grid-evals at every --gpus N; speed-up(N) = ms(1) / ms(N); "
                    "ms = mean over all steps (slowest rank, barriers outside), first5 / last5 = medians of single steps",
        }

    if rank == 0:
        FORMS = {2: ("pc", "k_fill_pc", "k_fill_pc (producer wave + consumer waves per column block, launched per 128 rows)"),
                 3: ("chain", "k_fill_chain", "k_fill_chain (one launch per fill: producer, consumer and fetcher waves per column block)"),
                 4: ("ck", "k_fill_ck", "k_fill_ck (one launch per fill: recurrence-only spine waves + tile workers)"),
                 6: ("hb", "k_fill_hb", "k_fill_hb (one launch per fill: spine waves that walk blocks of rows alone behind a halo + tile workers)")}
        fname, kprefix, knote = FORMS.get(form if args.variant == capi.FILL_SCALED else -1, ("other", "k_fill", f"fill variant {args.variant}"))
        launches = max(kn.value, 1)
        avg_launch_ms = kms.value / launches
        bytes_per_launch = 8.0 * cells_rank * args.steps / launches  # 8 B per stored cell
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        name = bytearray(128)
        L.stb_device_name((C.c_char * 128).from_buffer(name), 128)
        traffic, traffic_src = traffic_lookup(f"N{N}_M{M}_D{Dl}_{fname}", kprefix)
        out = {
            "metric": "S-table cells/s",
            "value": value,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("configs[1]: single-discount S-table N=M=10000 a=0.5" if (Dg == 1 and N == 10000 and M == 10000)
                             else f"{Dl} discount(s) per GPU of the {Dg}-point grid, S-table N={N} M={M}"),
                "N": N, "M": M, "discounts_per_gpu": Dl, "discounts_total": Dg,
                "cells_per_table": cells_rank // Dl, "variant": args.variant,
                "columns_per_block": Cc.value, "rows_per_launch": Rr.value,
                "parallelism": f"discount-sharded x{world}, all_gather of {Dg} probe scalars per step",
                "device": name.split(b"\0")[0].decode(),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": knote,
                "form": fname,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launches_per_step": launches / args.steps,
                "avg_launch_us": avg_launch_ms * 1e3,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "kernel_ms_per_step": kms.value / args.steps,
                "note": f"one table per GPU is bound by the N serial row steps of the recurrence -- this run: {N} rows x "
                        f"{avg_launch_ms * 1e6 / max(N, 1):.1f} ns of kernel time per row (the spine's walk + the hand-overs between its "
                        "workgroups + the last tiles) -- not by HBM; see DESIGN.md",
            },
        }
        vb = valu_busy("fill1" if Dl == 1 else "fill8", "k_fill_hb") if fname == "hb" else None
        if vb:
            out["roofline"].update(vb)
        extra = {}
        if batch64 is not None:
            out["scale_job"] = batch64
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, M, float(mine[0]))
        if world == 1 and not args.no_extra:
            try:
                s10k, _ = sweep_section(10000)
                tp = s10k["D64"]["two_pass"]
                fu = s10k["D64"]["fused"]
                alg = fu["grid_evals"] * (8 + 6.0 / 64)
                dev_ms = fu["fill_ms"] + fu["sweep_ms"] + fu["terms_ms"]
                tr, tr_src = traffic_lookup("grid_N10000_D64", "k_grid_hb")
                vbg = valu_busy("grid64", "k_grid_hb")
                out["roofline_sweep"] = {
                    "metric": "sampler grid-evals/s", "bound": "hbm",
                    "kernel": f"{grid_kernel_label(s10k['N'], s10k['M'], 64)}: the table walk of 64 discounts at N=M=10000 whose walking waves sum "
                              "count * log S over their own strips' listed cells (no table stored, no tile workers, every K-th row staged in "
                              "LDS), + the restaurant terms; what stb_groups_aterms runs for a grid beyond the tile-worker form's range",
                    "grid_evals": fu["grid_evals"], "device_ms": dev_ms, "fill_ms": fu["fill_ms"],
                    "value": fu["grid_evals"] / (dev_ms * 1e-3), "unit": "grid-evals/s",
                    "algorithmic_bytes": alg, "achieved": alg / (dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                    "frac": alg / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_src,
                    "note": "not an HBM-bound kernel (its traffic is the cell lists and the records between workgroups, not tables): every "
                            "table is a chain of N row steps walked with two waves a SIMD, the look-ups of the leftmost strips sit on it, and "
                            "a third of the workgroups start when the first end -- see valu_busy_frac (a static figure from the committed "
                            "profiles) and DESIGN.md section 4",
                    **(vbg or {}),
                    "end_to_end": {"fused_grid_evals_per_s": fu["grid_evals_per_s_end_to_end"], "fused_wall_ms": fu["wall_ms"],
                                   "two_pass_grid_evals_per_s": tp["grid_evals_per_s_end_to_end"], "two_pass_wall_ms": tp["wall_ms"]},
                }
                atr, atr_src = traffic_lookup("sweep_N10000_D64", "k_sweep_partial")
                extra["two_pass_sweep_N10000_D64"] = {
                    "kernel": "k_sweep_partial (+ k_reduce_final): sum of S_S(n,t) over 10^6 sorted pairs x 64 STORED tables",
                    "device_ms": tp["sweep_ms"], "grid_evals_per_s_sweep_only": tp["grid_evals"] / (tp["sweep_ms"] * 1e-3),
                    "algorithmic_GBs": tp["grid_evals"] * (8 + 6.0 / 64) / (tp["sweep_ms"] * 1e-3) / 1e9,
                    "frac_of_hbm_peak": tp["grid_evals"] * (8 + 6.0 / 64) / (tp["sweep_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": atr, "traffic_source": atr_src,
                    "note": "the form the product takes only when tables are wanted anyway (one discount, samplea's ARMS); each 8-byte gather moves a line"}
                extra["sampler_sweep_N10000"] = s10k
                s4k, g4k = sweep_section(4000)
                extra["sampler_sweep"] = s4k
                extra["sampler_sweep"].update(sampler_calls(g4k))
                if not args.no_cpu_baseline:
                    extra["sampler_sweep"]["cpu_sweep_baseline"] = cpu_sweep_baseline(g4k, s4k["N"], s4k["M"])
            except Exception as e:  # the extras must never take the contract line down
                extra["sampler_error"] = repr(e)
            try:
                torch.cuda.empty_cache()
                extra["dropin"] = {"N10000": dropin_section(10000, 10000, 0.5), "N4000": dropin_section(4000, 4000, 0.5)}
            except Exception as e:
                extra["dropin_error"] = repr(e)
        if extra:
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

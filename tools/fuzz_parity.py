#!/usr/bin/env python3
"""Randomised parity campaign beyond the fixed cases of tests/: for a time budget, random table bounds, discounts (0 and
values next to 0 and 1 among them), numbers of tables and output kinds through the DEFAULT dispatch (whatever fill form the
library picks), and random group sets (ragged restaurants, pairs on the diagonal, in column 1, beyond the bounds, repeated
cells) evaluated with 1-64 discounts through the fused forms and through new-pairs hand-overs -- every table cell and every
log-posterior against the oracle (tests/orc.py; |x-y| <= 1e-10 max(1,|y|), floats 3e-7).  Test infrastructure:
uses oracle/ as the checker only.
usage: python tools/fuzz_parity.py [seconds] [seed]      (repo root, GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import orc
from libstb_amd import capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261005
rng = np.random.default_rng(seed)
L = capi.lib()
O = orc.oracle()
TOL = 1e-10
t_end = time.time() + budget
worst = {"S": 0.0, "S1": 0.0, "Sf": 0.0, "V": 0.0, "aterms": 0.0}
count = {"S": 0, "Sf": 0, "V": 0, "aterms": 0, "cells": 0, "pairs": 0}
forms = {}
fb0, gfb0 = L.stb_fill_fallbacks(), L.stb_groups_fallbacks()


def pick_a(D):
    a = rng.uniform(0.001, 0.995, size=D)
    for i in range(D):
        u = rng.random()
        if u < 0.06:
            a[i] = 0.0
        elif u < 0.12:
            a[i] = rng.choice([1e-9, 1e-4, 0.999, 0.99999, 0.5, 2.0 / 3.0])
    return a


def rel(x, y):
    return float(np.max(np.abs(x - y) / np.maximum(1.0, np.abs(y)))) if x.size else 0.0


def fuzz_fill():
    kind = rng.choice(["S", "S", "Sf", "V"])
    N = int(rng.integers(3, 3300)) if rng.random() < 0.85 else int(rng.integers(3300, 9000))
    M = int(rng.integers(2, N + 1)) if rng.random() < 0.7 else int(rng.integers(2, min(N, 120) + 1))
    if N > 3300:
        M = min(M, 700)            # (the oracle's time)
    D = int(rng.integers(1, 13)) if N * M < 3e6 else int(rng.integers(1, 4))
    a = pick_a(D)
    if kind == "Sf" and L.stb_fill_takes_kind(N, M, D, 1) != 1:
        kind = "S"                  # (small tables: no kernel narrows, the table object narrows a double table)
    if kind == "V":
        T = capi.DeviceVTables(N, M, D=D)
        T.tables.fill_(float("nan"))
        T.fill(a)
        torch.cuda.synchronize()
        capi.check(L.stb_fill_status())
        for d in range(D):
            want = orc.fill_V(float(a[d]), N, M)
            got = T.packed_host(d)
            assert np.all(np.isfinite(got)), ("V", N, M, a[d])
            e = rel(got, want)
            assert e <= TOL, ("V", N, M, float(a[d]), e)
            worst["V"] = max(worst["V"], e)
            count["cells"] += got.size
        count["V"] += 1
        return
    T = capi.DeviceFloatTables(N, M, D=D) if kind == "Sf" else capi.DeviceTables(N, M, D=D)
    T.tables.fill_(float("nan"))
    T.fill(a)
    torch.cuda.synchronize()
    T.status()
    Cc, Rr, nl = (capi.C.c_int(), capi.C.c_int(), capi.C.c_int())
    f = L.stb_fill_tuning(N, M, D, capi.C.byref(Cc), capi.C.byref(Rr), capi.C.byref(nl))
    forms[f] = forms.get(f, 0) + 1
    for d in range(D):
        S1, want = orc.fill_S(float(a[d]), N, M)
        got = T.packed_host(d).astype(np.float64)
        assert np.all(np.isfinite(got)), (kind, N, M, a[d])
        e = rel(got, want)
        assert e <= (3e-7 if kind == "Sf" else TOL), (kind, N, M, float(a[d]), e)
        worst[kind] = max(worst[kind], e)
        e1 = rel(T.S1[d].cpu().numpy(), S1)
        assert e1 <= TOL, ("S1", N, M, float(a[d]), e1)
        worst["S1"] = max(worst["S1"], e1)
        count["cells"] += got.size
    count[kind] += 1


def fuzz_fill_big():
    """round 6: tables past the configs -- M in [10^4, 3 x 10^4], N up to 1.5 M -- through the default dispatch; four rows
    (the last among them) of every table against the oracle's streamed rows (the whole table is gigabytes)"""
    M = int(rng.integers(10000, 30001))
    N = int(min(45000, M + rng.integers(0, M // 2 + 1)))
    D = int(rng.integers(1, 3))
    a = pick_a(D)
    T = capi.DeviceTables(N, M, D=D)
    T.fill(a)
    torch.cuda.synchronize()
    T.status()
    f = L.stb_fill_tuning(N, M, D, None, None, None)
    forms[f] = forms.get(f, 0) + 1
    rows = sorted({N, N - int(rng.integers(1, 200)), int(rng.integers(M // 2, N)), int(rng.integers(3, M // 2))})
    for d in range(D):
        want = orc.rows_stream(float(a[d]), N, M, rows, threads=16)
        for r in rows:
            o = T.rowoff(r)
            got = T.tables[d, o:o + len(want[r])].cpu().numpy()
            e = rel(got, want[r])
            assert np.all(np.isfinite(got)) and e <= TOL, ("S big", N, M, float(a[d]), r, e)
            worst["S"] = max(worst["S"], e)
            count["cells"] += got.size
    count["S"] += 1
    count["big"] = count.get("big", 0) + 1
    del T
    torch.cuda.empty_cache()


def fuzz_aterms():
    N = int(rng.integers(520, 3000))
    M = int(rng.integers(10, N + 1))
    I = int(rng.integers(1, 60))
    K = rng.integers(0, 600, I).astype(np.int32)
    if K.sum() == 0:
        K[0] = 7
    G = int(K.sum())

    def pairs():
        n = rng.integers(2, N + 1, G).astype(np.uint32)
        if rng.random() < 0.5:
            t = (1 + rng.random(G) * np.minimum(n, M)).astype(np.int64)          # wide
        else:
            t = (1 + rng.random(G) * np.sqrt(n)).astype(np.int64)                # realistic
        t = np.minimum(np.minimum(t, n), M)
        k = rng.integers(0, G, 8)
        n[k[0]] = 1                                  # skipped
        t[k[1]] = min(int(n[k[1]]), M)               # the diagonal (when it lies inside the table)
        t[k[2]] = 1                                  # column 1
        n[k[3]], t[k[3]] = 3, 2
        n[k[4]], t[k[4]] = N, min(N - 1, M)
        n[k[5:8]], t[k[5:8]] = n[k[5]], t[k[5]]      # a repeated cell
        t = np.minimum(np.minimum(t, n), M)
        return n, t.astype(np.uint16)

    n, t = pairs()
    T = np.array([int(t[K[:i].sum():K[:i + 1].sum()].sum()) for i in range(I)], dtype=np.uint32)
    bpar = rng.uniform(0.1, 80.0, I)
    D = int(rng.choice([1, 1, 2, 3, 5, 8, 8, 13, 24, 32, 48, 64]))
    x = np.sort(rng.uniform(0.01, 0.98, D))
    h = L.stb_groups_create(I, orc.i32p(K), orc.u32p(T), orc.u32p(n), orc.u16p(t), orc.dp(bpar), N, M, D)
    assert h, capi.last_error()
    try:
        for rnd in range(2):
            out = np.zeros(D)
            capi.check(L.stb_groups_aterms(h, capi.dp(np.ascontiguousarray(x)), D, capi.dp(out)))
            assert np.all(np.isfinite(out)), (N, M, D, out)
            for d in sorted(set([0, D // 2, D - 1])):
                S1, tab = orc.fill_S(float(x[d]), N, M)
                want = O.orc_aterms_sum(float(x[d]), I, orc.i32p(K), orc.u32p(T), orc.u32p(n), orc.u16p(t), orc.dp(bpar), orc.dp(tab), orc.dp(S1), N, M)
                e = abs(out[d] - want) / max(1.0, abs(want))
                assert e <= TOL, ("aterms", N, M, D, d, float(x[d]), out[d], want)
                worst["aterms"] = max(worst["aterms"], e)
            count["aterms"] += 1
            count["pairs"] += G * D
            if rnd == 0:                              # new pairs into the kept set, then once more
                n, t = pairs()
                T = np.array([int(t[K[:i].sum():K[:i + 1].sum()].sum()) for i in range(I)], dtype=np.uint32)
                capi.check(L.stb_groups_pairs_begin(h))
                capi.check(L.stb_groups_pairs_put(h, orc.u32p(n), orc.u16p(t), G, None, None))
                capi.check(L.stb_groups_pairs_commit(h, orc.u32p(T), orc.dp(bpar), N, M))
    finally:
        L.stb_groups_free(h)


it = 0
t_log = time.time()
while time.time() < t_end:
    if it % 40 == 17 and os.environ.get("FUZZ_BIG", "1") != "0":
        fuzz_fill_big()
    else:
        (fuzz_fill if it % 3 else fuzz_aterms)()
    it += 1
    if time.time() - t_log > 30:
        t_log = time.time()
        print(f"... {it} cases, {count['cells']} cells, {count['pairs']} grid-evals", flush=True)
assert L.stb_fill_fallbacks() == fb0 and L.stb_groups_fallbacks() == gfb0, "a one-launch form gave up"
print(f"fuzz ok (seed {seed}, {budget:.0f} s): {count['S']} S fills, {count['Sf']} float fills, {count['V']} V fills = {count['cells']} cells; "
      f"{count['aterms']} grid evaluations = {count['pairs']} grid-evals; {count.get('big', 0)} tables with M in [10^4, 3 x 10^4] (four rows each "
      f"against the streamed oracle); forms picked {forms}")
print("worst deviation from the oracle, |x-y| / max(1,|y|): " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()) + "; none gave up")

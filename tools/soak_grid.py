#!/usr/bin/env python3
"""Soak test of the fused grid aterms (summing halo-block fills, tile workers or walking waves summing, and chain fills): the same evaluation over and over must give
the same bits and never fall back.   usage: python tools/soak_grid.py [seconds]      (repo root, GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import orc
from libstb_amd import capi, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
L = capi.lib()
cases = [(10000, 8), (10000, 24), (10000, 3), (4000, 32), (1500, 5), (10000, 64), (10000, 32), (10000, 48), (10000, 1)]
total = 0
for Nmax, D in cases:
    g = synth.groups(1000, 1000, Nmax, "wide")
    M = max(int(g.t.max()) + 1, 10)
    N = max(int(g.n.max()) + 1, M)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
    assert h, capi.last_error()
    x = np.ascontiguousarray(synth.discount_grid(64)[:D])
    ref, out = np.zeros(D), np.zeros(D)
    fb0 = L.stb_fill_fallbacks()
    capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(ref)))
    n, t_end = 0, time.time() + budget / len(cases)
    while time.time() < t_end:
        for _ in range(20):
            capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(out)))
            if not np.array_equal(out, ref):
                print(f"MISMATCH N={N} D={D} after {n} evaluations: {out} != {ref}", flush=True)
                sys.exit(1)
            n += 1
    if L.stb_fill_fallbacks() != fb0:
        print(f"FELL BACK N={N} D={D}: {capi.last_error()}", flush=True)
        sys.exit(1)
    L.stb_groups_free(h)
    total += n
    print(f"N={N} M={M} D={D}: {n} evaluations identical, none fell back", flush=True)
print(f"soak ok: {total} grid evaluations", flush=True)

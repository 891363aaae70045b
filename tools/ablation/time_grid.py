"""Time the fused grid aterms (64 / N discounts x 10^6 pairs, n < 10000) in the forms the library can take it.
usage: python tools/time_grid.py [D ...]      (repo root, GPU box)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import orc
from libstb_amd import capi, synth

L = capi.lib()
Ds = [int(x) for x in sys.argv[1:]] or [8, 64]
Nmax = int(os.environ.get("GRID_N", "10000"))
g = synth.groups(1000, 1000, Nmax, "wide")
M = max(int(g.t.max()) + 1, 10)
N = max(int(g.n.max()) + 1, M)
for D in Ds:
    x = np.ascontiguousarray(synth.discount_grid(64)[:D])
    ref = None
    for label, env in (("grid (walking waves sum)", {"STB_ATERMS_GRID": "1"}),
                       ("hb (halo blocks)", {"STB_ATERMS_GRID": "0", "STB_ATERMS_HB": "1"}),
                       ("ck (spine + workers)", {"STB_ATERMS_GRID": "0", "STB_ATERMS_HB": "0", "STB_ATERMS_CK": "1", "STB_ATERMS_CK_MAX_SPINE": "100000"}),
                       ("chain", {"STB_ATERMS_GRID": "0", "STB_ATERMS_HB": "0", "STB_ATERMS_CK": "0"}),
                       ("two-pass", {"STB_ATERMS_FUSED": "0"})):
        os.environ.update(env)
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
        assert h, capi.last_error()
        out = np.zeros(D)
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(out)))
        best = 1e9
        mf, ms, mt = C.c_float(), C.c_float(), C.c_float()
        for _ in range(8):
            t0 = time.perf_counter()
            capi.check(L.stb_groups_aterms_timed(h, capi.dp(x), D, capi.dp(out), C.byref(mf), C.byref(ms), C.byref(mt)))
            best = min(best, time.perf_counter() - t0)
        L.stb_groups_free(h)
        for k in env:
            os.environ.pop(k, None)
        if ref is None:
            ref = out.copy()
        err = float(np.max(np.abs(out - ref) / np.abs(ref)))
        print(f"N={N} M={M} D={D} {label:22s} wall {best * 1e3:7.3f} ms  (fill {mf.value:.3f} sweep {ms.value:.3f} terms {mt.value:.3f})  "
              f"{D * g.pairs / best / 1e9:6.2f} G grid-evals/s  max rel diff to first {err:.1e}", flush=True)

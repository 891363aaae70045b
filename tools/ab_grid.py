"""A/B timing of the fused grid aterms (D discounts x 10^6 pairs) in ONE process: forms and tunings alternating,
several rounds, medians of the best wall time and of the best device time of the summing kernel of each round (16 evaluations), and the mean over all of them; every result is compared with the
first spec's.
usage: python tools/ab_grid.py NMAX D "hb2,hb2@STB_HB2_C=4,hb,chain,..." [rounds] [profile]     (repo root, GPU box)
A spec is a form name (auto, hb2 the grid form whose walking waves sum, hb tile workers sum, chain, twopass) followed by @ENV=VALUE settings."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import orc
from libstb_amd import capi, synth

FORMS = {"auto": {},
         "hb2": {"STB_ATERMS_GRID": "1"},
         "hb": {"STB_ATERMS_GRID": "0", "STB_ATERMS_HB": "1"},
         "chain": {"STB_ATERMS_GRID": "0", "STB_ATERMS_HB": "0", "STB_ATERMS_CK": "0"},
         "twopass": {"STB_ATERMS_FUSED": "0"}}
Nmax = int(sys.argv[1])
D = int(sys.argv[2])
specs = sys.argv[3].split(",")
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
profile = sys.argv[5] if len(sys.argv) > 5 else "wide"
L = capi.lib()
g = synth.groups(1000, 1000, Nmax, profile)
M = max(int(g.t.max()) + 1, 10)
N = max(int(g.n.max()) + 1, M)
x = np.ascontiguousarray(np.resize(synth.discount_grid(64), D) if D > 1 else np.array([0.5]))
res = {s: ([], []) for s in specs}
allk = {}
ref = None
for r in range(rounds):
    for s in specs:
        form, *envs = s.split("@")
        env = dict(FORMS[form])
        for kv in envs:
            k_, v_ = kv.split("=")
            env[k_] = v_
        os.environ.update(env)
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
        assert h, capi.last_error()
        out = np.zeros(D)
        fb = L.stb_fill_fallbacks()
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(out)))
        wall, kern, ks = 1e9, 1e9, []
        mf, ms, mt = C.c_float(), C.c_float(), C.c_float()
        for _ in range(16):
            t0 = time.perf_counter()
            capi.check(L.stb_groups_aterms_timed(h, capi.dp(x), D, capi.dp(out), C.byref(mf), C.byref(ms), C.byref(mt)))
            wall = min(wall, time.perf_counter() - t0)
            kern = min(kern, mf.value)
            ks.append(mf.value)
        allk.setdefault(s, []).extend(ks)
        L.stb_groups_free(h)
        if L.stb_fill_fallbacks() != fb:
            print(f"{s}: FELL BACK", flush=True)
        for k_ in env:
            os.environ.pop(k_, None)
        if ref is None:
            ref = out.copy()
        err = float(np.max(np.abs(out - ref) / np.maximum(1.0, np.abs(ref))))
        if err > 1e-11:
            print(f"{s}: max rel diff to the first spec {err:.2e}", flush=True)
        res[s][0].append(wall * 1e3)
        res[s][1].append(kern)
for s in specs:
    w, k = res[s]
    print(f"N={N} M={M} D={D} {profile} {s:56s} wall " + " ".join(f"{v:.3f}" for v in w) + f"  median {np.median(w):.3f} ms | fill kernel median {np.median(k):.3f} ms"
          f" (all launches: mean {np.mean(allk[s]):.3f}, max {np.max(allk[s]):.3f}) | {D * g.pairs / np.median(w) / 1e6:7.2f} G grid-evals/s", flush=True)

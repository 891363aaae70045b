"""A/B timing of fill forms and tunings in ONE process, alternating, several rounds, medians.
usage: python tools/ab_ck.py N D "ck@STB_CK_C=2@STB_CK_P=4,chain,pc,..." [rounds] [M]     (repo root, GPU box)
A spec is a form name (ck, chain, pc, auto) followed by @ENV=VALUE settings that hold for that spec only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from libstb_amd import capi, synth

N = int(sys.argv[1])
D = int(sys.argv[2])
specs = sys.argv[3].split(",")
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
M = int(sys.argv[5]) if len(sys.argv) > 5 else N
a = (np.resize(synth.discount_grid(64), D) if D > 64 else synth.discount_grid(64)[:D]) if D > 1 else np.array([0.5])
FORMS = {"hb": capi.FILL_HB, "ck": capi.FILL_CK, "chain": capi.FILL_CHAIN, "pc": capi.FILL_PC, "auto": capi.FILL_SCALED}
T = capi.DeviceTables(N, M, D=D)
L = capi.lib()
res = {s: [] for s in specs}
for r in range(rounds):
    for s in specs:
        form, *envs = s.split("@")
        for kv in envs:
            k_, v_ = kv.split("=")
            os.environ[k_] = v_
        var = FORMS[form]
        fb = L.stb_fill_fallbacks()
        T.fill(a, var)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(6):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            T.fill(a, var)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        try:
            T.status()
        except capi.StbError as e:
            print(f"{s}: {e}", flush=True)
        if L.stb_fill_fallbacks() != fb:
            print(f"{s}: FELL BACK", flush=True)
        for kv in envs:
            os.environ.pop(kv.split("=")[0], None)
        res[s].append(best)
cells = T.cells * D
for s in specs:
    med = float(np.median(res[s]))
    print(f"N={N} M={M} D={D} {s:60s} " + " ".join(f"{x:.3f}" for x in res[s]) +
          f"   median {med:.3f} ms  {cells * 8 / med / 1e6:7.0f} GB/s", flush=True)

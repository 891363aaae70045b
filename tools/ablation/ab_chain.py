"""A/B timing of chain-fill strip shapes in ONE process, alternating, several rounds (the first fills of a
process and the part's temperature move single measurements by ~10 %).
usage: python tools/ab_chain.py N D "C:P:MG:NF:RD[@ENV=VAL...],pc,..." [rounds]      (run from the repo root)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstb_amd import capi, synth
N = int(sys.argv[1]); D = int(sys.argv[2]); shapes = sys.argv[3].split(","); rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
T = capi.DeviceTables(N, N, D=D)
res = {s: [] for s in shapes}
for r in range(rounds):
    for s in shapes:
        shape, *envs = s.split("@")
        for kv in envs:
            k_, v_ = kv.split("=")
            os.environ[k_] = v_
        if shape == "pc":
            var = capi.FILL_PC
        else:
            C_, P_, MG, NF, RD = shape.split(":")
            os.environ.update(STB_CHAIN_C=C_, STB_CHAIN_P=P_, STB_CHAIN_MG=MG, STB_CHAIN_NF=NF, STB_CHAIN_RD=RD)
            var = capi.FILL_CHAIN
        T.fill(a, var); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); T.fill(a, var); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        T.status()
        for kv in envs:
            os.environ.pop(kv.split("=")[0], None)
        res[s].append(best)
for s in shapes:
    print(f"N={N} D={D} {s:34s} ms per round: " + " ".join(f"{x:.3f}" for x in res[s]) + f"   median {np.median(res[s]):.3f}")

"""A/B of one environment switch (a tunable the library reads per call) between fills of ONE process, settings alternating over six
rounds: python tools/ab_env.py N D VAR v1,v2,...      (repo root, GPU box)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from libstb_amd import capi, synth
N, D = int(sys.argv[1]), int(sys.argv[2])
var, vals = sys.argv[3], sys.argv[4].split(",")
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
T = capi.DeviceTables(N, N, D=D)
res = {v: [] for v in vals}
for rnd in range(6):
    for v in vals:
        os.environ[var] = v
        for _ in range(3): T.fill(a)
        torch.cuda.synchronize()
        ts = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); T.fill(a); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        T.status()
        ts.sort(); res[v].append(ts[len(ts)//2])
for v in vals: print(f"N={N} D={D} {var}={v}: medians per round " + " ".join(f"{x:.3f}" for x in res[v]))

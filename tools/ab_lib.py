"""A/B of one fill workload between library builds, each in its own process run alternately (a library is loaded once
per process): python tools/ab_lib.py N D rounds libA.so libB.so ...      (repo root, GPU box)"""
import os
import subprocess
import sys

N, D, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
libs = sys.argv[4:]
child = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from libstb_amd import capi, synth
N, D = int(sys.argv[1]), int(sys.argv[2])
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
T = capi.DeviceTables(N, N, D=D)
for _ in range(5):
    T.fill(a)
torch.cuda.synchronize()
ts = []
for _ in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); T.fill(a); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
T.status()
ts.sort()
print(f"{ts[0]:.4f} {ts[len(ts)//2]:.4f}")
"""
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, STB_LIB_PATH=os.path.abspath(l))
        out = subprocess.run([sys.executable, "-c", child, N, D], env=env, capture_output=True, text=True, timeout=300)
        line = [x for x in out.stdout.strip().splitlines() if x and x[0].isdigit()]
        if not line:
            print(l, "failed:", out.stderr[-300:])
            continue
        res[l].append(tuple(float(x) for x in line[-1].split()))
for l in libs:
    print(f"N={N} D={D} {os.path.basename(l):28s} best/median ms per round: " + "  ".join(f"{b:.3f}/{m:.3f}" for b, m in res[l]), flush=True)

"""A/B of one fill workload between ENVIRONMENT settings, each in its own process run alternately:
python tools/ab_envs.py N D rounds "VAR=v VAR2=w" "VAR=x" ...      ("-" = no setting; repo root, GPU box)"""
import os
import subprocess
import sys

N, D, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
specs = sys.argv[4:]
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
res = {s: [] for s in specs}
for r in range(rounds):
    for s in specs:
        env = dict(os.environ)
        if s != "-":
            for kv in s.split():
                k, v = kv.split("=", 1)
                env[k] = v
        out = subprocess.run([sys.executable, "-c", child, N, D], env=env, capture_output=True, text=True, timeout=300)
        line = [x for x in out.stdout.strip().splitlines() if x and x[0].isdigit()]
        if not line:
            print(s, "failed:", out.stderr[-300:])
            continue
        res[s].append(tuple(float(x) for x in line[-1].split()))
for s in specs:
    print(f"N={N} D={D} {s:40s} best/median ms per round: " + "  ".join(f"{b:.3f}/{m:.3f}" for b, m in res[s]), flush=True)

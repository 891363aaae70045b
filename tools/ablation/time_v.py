#!/usr/bin/env python3
"""Time the V-table fill (stb_fill_V).  usage: time_v.py [N] [D]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstb_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = capi.DeviceVTables(N, N, D=D)
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
T.fill(a); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); T.fill(a); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"V fill N={N} D={D}: {best:.3f} ms  {T.cells * D / best / 1e6:.2f} Gcells/s", flush=True)

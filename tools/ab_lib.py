#!/usr/bin/env python3
"""A/B: time the auto-selected fill with an alternative build of the library (ALT=<suffix>)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi
alt = os.environ.get("ALT")
if alt:
    capi.LIB_PATH = capi.LIB_PATH.replace("libstb_amd.so", f"libstb_amd_{alt}.so")
import numpy as np, torch
from libstb_amd import synth
for N, D in ((10000, 1), (10000, 4), (10000, 8), (10000, 16), (4000, 1)):
    T = capi.DeviceTables(N, N, D=D)
    a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
    T.fill(a); torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); T.fill(a); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{alt or 'base':5s} N={N} D={D}: min {ts[0]:.3f} median {ts[len(ts)//2]:.3f} ms", flush=True)
    del T

#!/usr/bin/env python3
"""A/B: time the auto-selected fill with libstb_amd_old.so (an earlier commit) against the current one"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi
if os.environ.get("OLD") == "1":
    capi.LIB_PATH = capi.LIB_PATH.replace("libstb_amd.so", "libstb_amd_old.so")
import numpy as np, torch
from libstb_amd import synth
for N, D in ((10000, 1), (10000, 8), (10000, 16), (4000, 64)):
    T = capi.DeviceTables(N, N, D=D)
    a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
    T.fill(a); torch.cuda.synchronize()
    best = 1e9
    for _ in range(10):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); T.fill(a); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"{'old' if os.environ.get('OLD') == '1' else 'new'} N={N} D={D}: {best:.3f} ms", flush=True)
    del T

#!/usr/bin/env python3
"""Sweep the fill tunables (columns per lane C, rows per launch R) and print cells/s.
usage: python tools/tune_fill.py [N] [D] [variant]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from libstb_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
combos = os.environ.get("COMBOS", "1:16,1:32,1:48,2:16,2:32,2:64,2:96,4:32,4:64,4:128").split(",")
T = capi.DeviceTables(N, N, D=D)
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
cells = T.cells * D
for cb in combos:
    C, R = cb.split(":")
    os.environ["STB_FILL_C"] = C
    os.environ["STB_FILL_R"] = R
    T.fill(a, variant)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        T.fill(a, variant)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"N={N} D={D} var={variant} C={C} R={R}: {best:8.3f} ms  {cells / best / 1e6:9.2f} Gcells/s"
          f"  {cells * 8 / best / 1e6:8.1f} GB/s", flush=True)

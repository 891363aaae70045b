#!/usr/bin/env python3
"""Run a handful of fills (for profiling under rocprofv3). usage: one_fill.py N D reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstb_amd import capi, synth
N = int(sys.argv[1]); D = int(sys.argv[2]); reps = int(sys.argv[3])
T = capi.DeviceTables(N, N, D=D)
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
for _ in range(reps):
    T.fill(a, int(os.environ.get('VARIANT', '0')))
torch.cuda.synchronize()
print("done")

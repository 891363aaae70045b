#!/usr/bin/env python3
"""diagnostic: where do the waves of k_fill_pc spend their cycles (work between barriers vs total)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi
capi.LIB_PATH = capi.LIB_PATH.replace("libstb_amd.so", "libstb_amd_stamp.so")
import numpy as np, torch
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = capi.DeviceTables(10000, 10000, D=D)
a = np.linspace(0.1, 0.9, D) if D > 1 else np.array([0.5])
T.fill(a, 5); torch.cuda.synchronize()
os.environ["STB_STAMP_FILE"] = "gpurun_out/stamps_pc.txt"
T.fill(a, 5); torch.cuda.synchronize()
rows = np.loadtxt("gpurun_out/stamps_pc.txt", dtype=np.int64)
for k in (2, 20, 60):
    r = rows[rows[:, 0] == k]
    for w in (0, 1, 2):
        rw = r[r[:, 2] == w]
        if len(rw):
            print(f"launch {k} wave {w}: blocks {len(rw)} work {np.median(rw[:,3]):.0f} total {np.median(rw[:,4]):.0f} trips {np.median(rw[:,5]):.0f} -> work/trip {np.median(rw[:,3]/np.maximum(rw[:,5],1)):.0f} total/trip {np.median(rw[:,4]/np.maximum(rw[:,5],1)):.0f}")

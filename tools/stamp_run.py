#!/usr/bin/env python3
"""diagnostic: run one split fill with the stamped library and summarise where a k_rec launch spends time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi
capi.LIB_PATH = capi.LIB_PATH.replace("libstb_amd.so", "libstb_amd_stamp.so")
import numpy as np, torch
T = capi.DeviceTables(10000, 10000, D=1)
os.environ.pop("STB_STAMP_FILE", None)
T.fill([0.5], 3); torch.cuda.synchronize()
os.environ["STB_STAMP_FILE"] = "gpurun_out/stamps.txt"
T.fill([0.5], 3); torch.cuda.synchronize()
rows = np.loadtxt("gpurun_out/stamps.txt", dtype=np.int64)
for k in (1, 10, 50, 100):
    r = rows[rows[:, 0] == k]
    if len(r) == 0: continue
    t0 = r[:, 2].min()
    d = r[:, 2:] - t0
    print(f"launch {k}: strips {len(r)}; start skew max {d[:,0].max()}; per-wave cycles: prologue {np.median(r[:,3]-r[:,2]):.0f} setup {np.median(r[:,4]-r[:,3]):.0f} rows {np.median(r[:,5]-r[:,4]):.0f} (max {np.max(r[:,5]-r[:,4])}) tail {np.median(r[:,6]-r[:,5]):.0f}; last wave ends at {d[:,4].max()} cycles (100MHz ticks? see note)")

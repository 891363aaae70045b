#!/usr/bin/env python3
"""diagnostic: where do the waves of k_fill_chain spend their cycles (work between barriers vs total)
usage: stamp_chain.py N M  (needs `make -C libstb_amd/csrc stamp`)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi
capi.LIB_PATH = capi.LIB_PATH.replace("libstb_amd.so", "libstb_amd_stamp.so")
import numpy as np, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
M = int(sys.argv[2]) if len(sys.argv) > 2 else N
T = capi.DeviceTables(N, M, D=1)
a = np.array([0.5])
T.fill(a, capi.FILL_CHAIN); torch.cuda.synchronize()
os.environ["STB_STAMP_FILE"] = "gpurun_out/stamps_chain.txt"
T.fill(a, capi.FILL_CHAIN); torch.cuda.synchronize()
T.status()
rows = np.loadtxt("gpurun_out/stamps_chain.txt", dtype=np.int64, ndmin=2)
blocks = sorted(set(rows[:, 0]))
tmin = rows[:, 5].min()
for j in blocks[:2] + blocks[len(blocks) // 2:len(blocks) // 2 + 1] + blocks[-1:]:
    for r in rows[rows[:, 0] == j]:
        print(f"block {r[0]:3d} wave {r[1]:2d}: start {r[5] - tmin:9d} total {r[3]:9d} waiting {r[2]:9d} ({100.0 * r[2] / max(r[3], 1):5.1f}%) in {r[4]:6d} waits")

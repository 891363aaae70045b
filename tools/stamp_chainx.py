#!/usr/bin/env python3
"""diagnostic: timeline of k_fill_chainx (wall-clock start/end and LDS waiting per wave)
usage: stamp_chainx.py N M  (needs `make -C libstb_amd/csrc stamp`)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi
capi.LIB_PATH = capi.LIB_PATH.replace("libstb_amd.so", "libstb_amd_stamp.so")
import numpy as np, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
M = int(sys.argv[2]) if len(sys.argv) > 2 else N
T = capi.DeviceTables(N, M, D=1)
a = np.array([0.5])
T.fill(a, capi.FILL_CHAINX); torch.cuda.synchronize()
os.environ["STB_STAMP_FILE"] = "gpurun_out/stamps_chainx.txt"
T.fill(a, capi.FILL_CHAINX); torch.cuda.synchronize()
T.status()
rows = np.loadtxt("gpurun_out/stamps_chainx.txt", dtype=np.int64, ndmin=2)
t0 = rows[:, 5].min()
secs = rows[rows[:, 0] >= 500]
rows = rows[rows[:, 0] < 500]
prod = rows[rows[:, 0] < 256]
conv = rows[rows[:, 0] >= 256]
print("producer blocks (times in us from the first wave start; 100 MHz clock)")
for j in sorted(set(prod[:, 0])):
    if j < 3 or j % 5 == 0 or j == prod[:, 0].max():
        for r in prod[prod[:, 0] == j]:
            print(f" block {r[0]:3d} wave {r[1]}: start {(r[5]-t0)/100:8.1f} end {(r[3]-t0)/100:8.1f} waited {r[2]/100:8.1f} us in {r[4]:5d} waits")
print("converter blocks: start / end of the slowest wave")
for q in sorted(set(conv[:, 0])):
    if (q - 256) % 16 == 0 or q == conv[:, 0].max():
        r = conv[conv[:, 0] == q]
        print(f" chunk {q-256:3d}: start {(r[:,5].min()-t0)/100:8.1f} end {(r[:,3].max()-t0)/100:8.1f}  waits/wave {r[:,4].mean():6.1f}")
print(f"all: end {(rows[:,3].max()-t0)/100:.1f} us")

print("producer sections (s_memtime ticks summed over the trips): top/checks, look-ahead, period, pitch, rows, post+drain")
for r in secs:
    v = [r[2] & 0xffffffff, r[2] >> 32, r[3] & 0xffffffff, r[3] >> 32, r[4] & 0xffffffff, r[4] >> 32]
    print(f" block {r[0]-500} wave {r[1]}: " + " ".join(f"{x:9d}" for x in v) + f"  sum {sum(v)}")

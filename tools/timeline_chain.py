#!/usr/bin/env python3
"""diagnostic: the hand-off timeline of k_fill_chain (needs `make -C libstb_amd/csrc stamp`):
when the last producer of block j finished trip t, when its publisher stored it, when block j+1's
fetcher delivered it, and when block j+1's first producer started the trip that uses it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi
capi.LIB_PATH = capi.LIB_PATH.replace("libstb_amd.so", "libstb_amd_stamp.so")
import numpy as np, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = capi.DeviceTables(N, N, D=1)
a = np.array([0.5])
T.fill(a, capi.FILL_CHAIN); torch.cuda.synchronize()
os.environ["STB_TIMELINE_FILE"] = "gpurun_out/timeline_chain.txt"
T.fill(a, capi.FILL_CHAIN); torch.cuda.synchronize()
T.status()
r = np.loadtxt("gpurun_out/timeline_chain.txt", dtype=np.int64, ndmin=2)
tab = {(int(x[0]), int(x[1])): x[2:] for x in r}
blocks = sorted(set(int(x[0]) for x in r))
t0 = min(int(v[v > 0].min()) for v in tab.values())
us = lambda x: (x - t0) / 100.0
for j in blocks[:-1]:
    if j not in (0, 1, 2, len(blocks) // 2, blocks[-2]):
        continue
    rows = []
    for t in range(1280):
        a_ = tab.get((j, t)); b_ = tab.get((j + 1, t))
        if a_ is None or b_ is None or a_[0] == 0 or a_[1] == 0 or b_[2] == 0 or b_[3] == 0:
            continue
        rows.append((t, a_[0], a_[1], b_[2], b_[3]))
    if not rows:
        continue
    rows = np.array(rows, dtype=np.int64)
    pub = (rows[:, 2] - rows[:, 1]) / 100.0
    dlv = (rows[:, 3] - rows[:, 2]) / 100.0
    use = (rows[:, 4] - rows[:, 3]) / 100.0
    tot = (rows[:, 4] - rows[:, 1]) / 100.0
    print(f"hop {j}->{j+1}: trips {len(rows)}  produced->published {np.median(pub):6.2f} us  published->delivered {np.median(dlv):6.2f}"
          f"  delivered->used {np.median(use):6.2f}  produced->used median {np.median(tot):6.2f} (p10 {np.percentile(tot,10):.2f} p90 {np.percentile(tot,90):.2f})")
    k = len(rows) // 2
    print("   sample trips:", [(int(x[0]), round(us(x[1]), 1), round(us(x[2]), 1), round(us(x[3]), 1), round(us(x[4]), 1)) for x in rows[k:k + 4]])

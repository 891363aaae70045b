#!/usr/bin/env python3
"""diagnostic: the hand-off timeline of the chain-form fill (needs `make -C libstb_amd/csrc stamp`):
when the producer of strip j finished trip t [0], when its publishing consumer stored the edge [1],
when strip j+1's fetcher delivered it [2], and when strip j+1's producer started the trip that uses
it [3].  Prints the producer's trip time and a histogram of the hand-off per hop.
usage: python tools/timeline_chain.py [N] [out-file]        (run from the repo root; STB_CHAIN_C, _P, _MG, _NF, _RD apply)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi
capi.LIB_PATH = os.environ.get("STB_LIB_PATH") or capi.LIB_PATH.replace("libstb_amd.so", "libstb_amd_stamp.so")
import numpy as np, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/timeline_chain.txt"
T = capi.DeviceTables(N, N, D=1)
a = np.array([0.5])
T.fill(a, capi.FILL_CHAIN); torch.cuda.synchronize()
os.environ["STB_TIMELINE_FILE"] = out
T.fill(a, capi.FILL_CHAIN); torch.cuda.synchronize()
T.status()
r = np.loadtxt(out, dtype=np.int64, ndmin=2)
cons = {(int(x[0]), int(x[1]) - 200000): x[2:] for x in r if x[1] >= 200000}
r = r[r[:, 1] < 200000]
waves = {(int(x[0]), int(x[1]) - 100000): x[2:] for x in r if x[1] >= 100000}
r = r[r[:, 1] < 100000]
tab = {(int(x[0]), int(x[1])): x[2:] for x in r}
blocks = sorted(set(int(x[0]) for x in r))
t0 = min(int(v[v > 0].min()) for v in tab.values())
us = lambda x: (x - t0) / 100.0
print(f"strips stamped: {len(blocks)}; all stamps in 10 ns ticks of the 100 MHz wall clock")
for j in (0, 1, len(blocks) // 2):
    ends = np.array([tab[(j, t)][0] for t in range(1280) if (j, t) in tab and tab[(j, t)][0] > 0])
    starts = np.array([tab[(j, t)][3] for t in range(1280) if (j, t) in tab and tab[(j, t)][3] > 0])
    if len(ends) > 10:
        dt = np.diff(ends) / 100.0
        busy = (ends[-len(starts):] - starts[-len(ends):]) / 100.0 if len(starts) and len(ends) else np.array([0.0])
        print(f"strip {j}: producer trip-to-trip median {np.median(dt):.3f} us (p10 {np.percentile(dt,10):.3f}, p90 {np.percentile(dt,90):.3f}) "
              f"= {np.median(dt)*1000/8:.1f} ns/row; start->end of a trip median {np.median(busy):.3f} us")
for (j, w), x in sorted(waves.items()):
    if j in (0, 1, len(blocks) // 2):
        print(f"strip {j} producer {w}: looked again {x[0]} times, then waited {x[1]} times ({x[2]} of them for its left input), {x[3] / 100.0:.1f} us in all")
for (j, c), x in sorted(cons.items()):
    if j in (0, len(blocks) // 2) and x[0] > 0:
        print(f"strip {j} consumer {c}: {x[0]} items, {x[2] / 100.0:.1f} us in all = {x[2] / 100.0 / x[0]:.2f} us per item, of which waiting for the producer {x[1] / 100.0 / x[0]:.2f}")
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
    hist, edges = np.histogram(tot, bins=[0, 1, 1.5, 2, 2.5, 3, 3.5, 4, 5, 6, 8, 12, 1e9])
    print("   histogram of produced->used (us):", " ".join(f"<{e:g}:{h}" for h, e in zip(hist, edges[1:])))
    # how the lag develops along the strip: the first trips, then every eighth of the way
    idx = sorted(set(list(range(min(6, len(rows)))) + [len(rows) * q // 8 for q in range(1, 8)] + [len(rows) - 1]))
    print("   trip: published->delivered, delivered->used, produced->used, own trip time:",
          " ".join(f"{int(rows[i, 0])}:{dlv[i]:.2f},{use[i]:.2f},{tot[i]:.2f},{(rows[i, 4] - rows[i - 1, 4]) / 100.0 if i else 0:.2f}" for i in idx))
    k = len(rows) // 2
    print("   sample trips:", [(int(x[0]), round(us(x[1]), 1), round(us(x[2]), 1), round(us(x[3]), 1), round(us(x[4]), 1)) for x in rows[k:k + 4]])
last = blocks[-1]
ends = [tab[(last, t)][0] for t in range(1280) if (last, t) in tab and tab[(last, t)][0] > 0]
if ends:
    print(f"first stamp -> last stamped trip of strip {last}: {us(max(ends)):.1f} us")

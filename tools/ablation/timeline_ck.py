"""Timeline of one checkpointed fill (k_fill_ck): where the spine's time goes (row time per wave strip, lag between
neighbouring strips inside a workgroup and across workgroups) and what the tile workers do (wait, load, compute).
usage: python tools/timeline_ck.py N D [out.txt]      (repo root, GPU box; honours STB_CK_* tunables)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from libstb_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/timeline_ck.txt"
M = int(sys.argv[4]) if len(sys.argv) > 4 else N
raw = out + ".raw"
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
T = capi.DeviceTables(N, M, D=D)
for _ in range(3):
    T.fill(a, capi.FILL_CK)
torch.cuda.synchronize()
os.environ["STB_CK_TIMELINE"] = raw
T.fill(a, capi.FILL_CK)
torch.cuda.synchronize()
del os.environ["STB_CK_TIMELINE"]
T.status()

S, W, geo = {}, [], None
for line in open(raw):
    f = line.split()
    if f[0] == "G":
        geo = dict(zip(("C", "P", "JW", "NBK", "RB", "TP", "G", "D"), map(int, f[1:])))
    elif f[0] == "S":
        S[int(f[1])] = np.array(list(map(int, f[2:])), dtype=np.int64)
    else:
        W.append(tuple(map(int, f[1:])))
C, P, JW, NBK, RB = geo["C"], geo["P"], geo["JW"], geo["NBK"], geo["RB"]
tick = 0.01  # us per wall_clock64 tick (100 MHz)
t0 = min(s[0] for s in S.values() if s[0])
lines = [f"# k_fill_ck timeline, N=M={N}, D={D} (table 0 stamped), C={C} P={P}: {JW} wave strips of {64 * C} columns, "
         f"{NBK} blocks of {RB * 8} rows; times in us from the first spine wave's start"]
ends = {}
for jw in sorted(S):
    s = S[jw]
    blocks = [(b, s[b]) for b in range(1, NBK + 1) if s[b]]
    start, end = s[0], s[NBK + 1]
    ends[jw] = end
    if len(blocks) >= 3:
        dt = np.diff([t for _, t in blocks]) * tick
        nsrow = np.median(dt) / (RB * 8) * 1000
        p10, p90 = np.percentile(dt, 10) / (RB * 8) * 1000, np.percentile(dt, 90) / (RB * 8) * 1000
    else:
        nsrow = p10 = p90 = float("nan")
    if jw < 8 or jw % 8 == 0 or jw >= JW - 2:
        lines.append(f"strip {jw:3d}: start {(start - t0) * tick:8.1f} end {(end - t0) * tick:8.1f}  row time median {nsrow:6.1f} ns (p10 {p10:.1f} p90 {p90:.1f})")
# lag between neighbours at common block boundaries
intra, inter = [], []
for jw in range(1, JW):
    if jw not in S or jw - 1 not in S:
        continue
    x, y = S[jw - 1][1:NBK + 1], S[jw][1:NBK + 1]
    both = (x != 0) & (y != 0)
    if both.sum() == 0:
        continue
    lag = (y[both] - x[both]) * tick
    (intra if jw % P else inter).append((jw, np.median(lag), lag[0], lag[-1]))
if intra:
    lines.append("lag of a strip behind its left neighbour, same workgroup (us): median of medians %.2f; first/last block median %.2f / %.2f"
                 % (np.median([m for _, m, _, _ in intra]), np.median([f for _, _, f, _ in intra]), np.median([l for _, _, _, l in intra])))
if inter:
    lines.append("lag across workgroups (us): median of medians %.2f; first/last block median %.2f / %.2f; per hop: " %
                 (np.median([m for _, m, _, _ in inter]), np.median([f for _, _, f, _ in inter]), np.median([l for _, _, _, l in inter]))
                 + " ".join(f"{jw}:{m:.1f}" for jw, m, _, _ in inter[:12]))
last_spine = max(ends.values())
lines.append(f"spine: first start -> last end {(last_spine - t0) * tick:.1f} us; strip 0 alone {(ends[0] - S[0][0]) * tick:.1f} us")
# workers
Wa = np.array([w for w in W if w[2] and w[4]], dtype=np.int64)
if len(Wa):
    wait = (Wa[:, 3] - Wa[:, 2]) * tick
    comp = (Wa[:, 4] - Wa[:, 3]) * tick
    lines.append(f"workers: {len(Wa)} tiles of table 0; claimed->inputs loaded median {np.median(wait):.1f} us (p90 {np.percentile(wait, 90):.1f}); "
                 f"compute median {np.median(comp):.1f} us (p10 {np.percentile(comp, 10):.1f} p90 {np.percentile(comp, 90):.1f}); sum of compute {comp.sum() / 1000:.2f} ms")
    lines.append(f"last tile done {(Wa[:, 4].max() - t0) * tick:.1f} us; last spine end {(last_spine - t0) * tick:.1f} us")
    # how long after its block was finished by the spine a tile was done
    late = []
    for jw, b, c0, c1, c2, hw in Wa:
        fin = S[jw][b + 1] if b + 1 <= NBK and S[jw][b + 1] else S[jw][NBK + 1]
        late.append((c2 - fin) * tick)
    late = np.array(late)
    lines.append(f"tile done after its block left the spine: median {np.median(late):.1f} us, p90 {np.percentile(late, 90):.1f}, max {late.max():.1f}")
    cus = len(set(int(h) & 0xffffff00 | (int(h) >> 32) << 40 for h in Wa[:, 5]))
    lines.append(f"distinct (xcc, hw id sans wave) values among workers: {cus}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

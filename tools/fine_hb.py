"""Where a spine wave of k_fill_hb spends a block: shader-clock stamps at seven points of the block loop (a diagnostic
build of the library: make -C libstb_amd/csrc variant FILE=fill_hb NAME=tlf DEFS="-DHB_TL_CYCLES -DHB_TL_FINE"), medians
over the blocks of strips in the middle of the table, by position of the wave in its workgroup.
usage: STB_LIB_PATH=libstb_amd/lib/libstb_amd_tlf.so python tools/fine_hb.py [N] [D] [out.txt]     (repo root, GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from libstb_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/fine_hb.txt"
raw = out + ".raw"
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
T = capi.DeviceTables(N, N, D=D)
for _ in range(3):
    T.fill(a, capi.FILL_HB)
torch.cuda.synchronize()
os.environ["STB_HB_TIMELINE"] = raw
T.fill(a, capi.FILL_HB)
torch.cuda.synchronize()
del os.environ["STB_HB_TIMELINE"]
T.status()
buf = open(raw, "rb").read()
JW, NB, NT, C, P, R, U, Dd = np.frombuffer(buf, dtype=np.int32, count=8)
words = np.frombuffer(buf, dtype=np.uint64, offset=32 + 4 * NT).astype(np.int64)
base = JW * (NB + 2) + NT * 4
F = words[base: base + JW * NB * 8].reshape(JW, NB, 8)
os.remove(raw)
names = ["renorm", "ring store + post", "record store", "halo read + post", "scale", "rows", "loop back"]
# (a build may make only some of the seven stamps, -DHB_FINE_SET=mask: the steps between two stamps that are there are summed)
mid = F[JW // 2, (JW // 2 * U * C) // R + 8: NB - 1, :7]
have = [k for k in range(7) if (mid[:, k] > 0).mean() > 0.9]
if len(have) < 7:
    F = F.copy()
    for k in range(7):
        if k not in have:  # a missing stamp takes the value of the next one that is there (its step counts as 0)
            nxt = [h for h in have if h > k]
            F[:, :, k] = F[:, :, nxt[0]] if nxt else F[:, :, have[-1]]
    names = [n + ("" if (k + 1 in have or k == 6) and True else "") for k, n in enumerate(names)]
lines = [f"# k_fill_hb fine stamps (shader clock cycles), N={N} D={D} C={C} P={P} R={R} U={U}: {JW} strips, {NB} blocks; stamps made: {have} (a step that ends at a stamp not made shows 0 and is counted in the next)"]
for w in range(P):
    rows = []
    for j in range(w, JW, P):
        if j < JW // 4 or j >= 3 * JW // 4:
            continue
        b0 = (j * U * C) // R
        for b in range(b0 + 8, NB - 1):
            f = F[j, b]
            if (f[:7] > 0).all() and F[j, b + 1, 0] > 0:
                rows.append(list(np.diff(f[:7])) + [F[j, b + 1, 0] - f[6], F[j, b + 1, 0] - f[0]])
    if not rows:
        continue
    A = np.array(rows)
    med = np.median(A, axis=0)
    mean = A.mean(axis=0)
    lines.append(f"wave {w} of its workgroup ({len(rows)} blocks): median / mean cycles")
    for k, nm in enumerate(names):
        lines.append(f"   {nm:20s} {med[k]:8.0f} {mean[k]:8.0f}")
    lines.append(f"   {'whole block':20s} {med[7]:8.0f} {mean[7]:8.0f}   ({R} rows: {med[5] / R:.1f} cycles a row in the row loop)")
lines.append("# per strip (blocks from the 8th of the strip on): medians of halo read + post, ring store + post, whole block; share of blocks "
             "whose halo was there at the first look; median of (blocks the left neighbour had posted - this block)")
for j in list(range(0, min(JW, 14))) + list(range(JW // 2, min(JW, JW // 2 + 9))):
    b0 = (j * U * C) // R
    f = F[j, b0 + 8: NB - 1]
    nxt = F[j, b0 + 9: NB, 0]
    ok = (f[:, :7] > 0).all(axis=1) & (nxt > 0)
    if not ok.any():
        continue
    f, nxt = f[ok], nxt[ok]
    ahead = f[:, 7] - 100
    lines.append(f"  strip {j:3d} (wave {j % P}): halo {np.median(f[:, 4] - f[:, 3]):6.0f}  ring {np.median(f[:, 2] - f[:, 1]):6.0f}  rows {np.median(f[:, 6] - f[:, 5]):6.0f}"
                 f"  block {np.median(nxt - f[:, 0]):6.0f}  there at first look {np.mean(ahead >= 1) if j else 1.0:5.2f}  ahead {np.median(ahead) if j else 0:4.0f}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-30:]))

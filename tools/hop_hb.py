"""Where a hop between two spine workgroups of k_fill_hb goes: wall-clock stamps (10 ns) of the left workgroup's last strip
(record store issued), of the right workgroup's fetcher (the look that found the record complete: sent, back) and of its first
strip (halo taken), per block, in a diagnostic build:
    make -C libstb_amd/csrc variant FILE=fill_hb NAME=tlw DEFS=-DHB_TL_FINE
usage: STB_LIB_PATH=libstb_amd/lib/libstb_amd_tlw.so python tools/hop_hb.py [N] [D] [out.txt]     (repo root, GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from libstb_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/hop_hb.txt"
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
H = words[base + JW * NB * 8: base + JW * NB * 8 + JW * NB * 2].reshape(JW, NB, 2)
os.remove(raw)
us = 0.01
lines = [f"# k_fill_hb hops (wall clock, us), N={N} D={D} C={C} P={P} R={R}: {JW} strips, {NB} blocks; medians over a hop's blocks"]
tot = []
for j in range(P, JW, P):
    b0 = (j * U * C) // R
    rows = []
    for b in range(b0 + 8, NB - 1):
        st0, st1 = F[j - 1, b, 2], F[j - 1, b, 3]          # left strip: before / after its record store
        sent, back = H[j, b]
        took = F[j, b, 4]                                   # right strip: halo taken
        if min(st0, st1, sent, back, took) <= 0:
            continue
        rows.append((st1 - st0, sent - st1, back - sent, back - st1, took - back, took - st1))
    if len(rows) < 8:
        continue
    A = np.median(np.array(rows), axis=0) * us
    tot.append(A)
    if j // P <= 6 or j // P % 6 == 0:
        lines.append(f"hop into strip {j:3d}: the stores take {A[0]:5.2f} to issue; the look that found them was sent {A[1]:+5.2f} after they were issued and was back "
                     f"{A[2]:5.2f} later (= {A[3]:5.2f} after the stores); the first strip had the halo {A[4]:5.2f} after that: {A[5]:5.2f} in all")
T_ = np.median(np.array(tot), axis=0)
lines.append(f"median over {len(tot)} hops: store issue {T_[0]:.2f}; successful look sent {T_[1]:+.2f} after the stores, its round trip {T_[2]:.2f}; seen {T_[3]:.2f} after the stores; "
             f"halo taken {T_[4]:.2f} after seen; {T_[5]:.2f} in all")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
if os.environ.get("HOP_RAW"):
    j = int(os.environ["HOP_RAW"])
    bs = (j * U * C) // R + 20
    t0 = F[j - 1, bs, 0]
    print(f"# raw stamps (us from block {bs} of strip {j - 1}): block; left strip {j - 1}: top, before / after its record store, rows start, rows end; fetcher: look sent, back; strip {j}: top, after record store, halo taken, rows start, rows end")
    for b in range(bs, bs + 12):
        L, Rr = F[j - 1, b], F[j, b]
        print(b, " ".join(f"{(x - t0) * us:7.2f}" for x in (L[0], L[2], L[3], L[5], L[6])), " | ", " ".join(f"{(x - t0) * us:7.2f}" for x in H[j, b]), " | ",
              " ".join(f"{(x - t0) * us:7.2f}" for x in (Rr[0], Rr[3], Rr[4], Rr[5], Rr[6])))

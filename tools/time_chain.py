#!/usr/bin/env python3
"""Time the chain form of the fill against the auto-selected form and check it against it.
usage: python tools/time_chain.py N D [C:RH,...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from libstb_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
combos = (sys.argv[3] if len(sys.argv) > 3 else "4:10,4:8,2:6,2:8,1:3,1:6").split(",")
M = int(sys.argv[4]) if len(sys.argv) > 4 else N
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])


def timed(T, variant, reps=5):
    T.fill(a, variant)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        T.fill(a, variant)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


T = capi.DeviceTables(N, M, D=D)
cells = T.cells * D
ms = timed(T, capi.FILL_SCALED)
print(f"N={N} M={M} D={D} auto: {ms:8.3f} ms {cells / ms / 1e6:9.2f} Gcells/s", flush=True)
ref_rows = [T.row(d, n).clone() for d in range(D) for n in (3, 4, 130, N // 3, N - 1, N)]
ref_sum = T.tables.nan_to_num(0.0, 0.0, 0.0)[:, :8].sum().item()
for cb in combos:
    parts = cb.split(":")
    C, RH = parts[0], parts[1]
    os.environ["STB_CHAIN_P"] = C
    os.environ["STB_CHAIN_NC"] = RH
    os.environ["STB_CHAIN_NF"] = parts[2] if len(parts) > 2 else "1"
    T2 = capi.DeviceTables(N, M, D=D)
    T2.tables.fill_(float("nan"))
    ms = timed(T2, capi.FILL_CHAIN)
    T2.status()
    got = [T2.row(d, n) for d in range(D) for n in (3, 4, 130, N // 3, N - 1, N)]
    err = max(((g - r).abs() / r.abs().clamp(min=1.0)).max().item() for g, r in zip(got, ref_rows))
    print(f"N={N} M={M} D={D} chain {cb}: {ms:8.3f} ms {cells / ms / 1e6:9.2f} Gcells/s "
          f"{cells * 8 / ms / 1e6:8.1f} GB/s  max rel err vs auto on probe rows {err:.2e}", flush=True)
    del T2

for P in (os.environ.get("CHAINX_P", "").split(",") if D <= 2 and os.environ.get("CHAINX_P") else []):
    os.environ["STB_CHAINX_P"] = P
    T2 = capi.DeviceTables(N, M, D=D)
    T2.tables.fill_(float("nan"))
    ms = timed(T2, capi.FILL_CHAINX)
    T2.status()
    got = [T2.row(d, n) for d in range(D) for n in (3, 4, 130, N // 3, N - 1, N)]
    err = max(((g - r).abs() / r.abs().clamp(min=1.0)).max().item() for g, r in zip(got, ref_rows))
    print(f"N={N} M={M} D={D} chainx P={P}: {ms:8.3f} ms {cells / ms / 1e6:9.2f} Gcells/s "
          f"{cells * 8 / ms / 1e6:8.1f} GB/s  max rel err vs auto on probe rows {err:.2e}", flush=True)
    del T2

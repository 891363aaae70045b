"""Time k_fill_chain strip shapes against the auto-selected and the producer/consumer form and check them against it.
usage: python tools/sweep_strip.py N D "C:P:MG:NF:RD,..." [M]      (run from the repo root)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from libstb_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
combos = (sys.argv[3] if len(sys.argv) > 3 else "1:2:3:2:4,2:1:3:2:4,4:1:2:1:4").split(",")
M = int(sys.argv[4]) if len(sys.argv) > 4 else N
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
ROWS = (3, 4, 130, N // 3, N - 1, N)


def timed(T, variant, reps=6):
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
for name, var in (("auto", capi.FILL_SCALED), ("pc", capi.FILL_PC)):
    ms = timed(T, var)
    T.status()
    print(f"N={N} M={M} D={D} {name}: {ms:8.3f} ms {cells / ms / 1e6:9.2f} Gcells/s {cells * 8 / ms / 1e6:8.1f} GB/s", flush=True)
ref_rows = [T.row(d, n).clone() for d in range(D) for n in ROWS if n <= N]
del T
for cb in combos:
    C, P, MG, NF, RD = cb.split(":")
    os.environ.update(STB_CHAIN_C=C, STB_CHAIN_P=P, STB_CHAIN_MG=MG, STB_CHAIN_NF=NF, STB_CHAIN_RD=RD)
    T2 = capi.DeviceTables(N, M, D=D)
    T2.tables.fill_(float("nan"))
    try:
        ms = timed(T2, capi.FILL_CHAIN)
        fb = capi.lib().stb_fill_fallbacks()
        T2.status()
        fell = capi.lib().stb_fill_fallbacks() != fb
    except capi.StbError as e:
        print(f"chain {cb}: {e}", flush=True)
        continue
    got = [T2.row(d, n) for d in range(D) for n in ROWS if n <= N]
    err = max(((g - r).abs() / r.abs().clamp(min=1.0)).nan_to_num(9.9).max().item() for g, r in zip(got, ref_rows))
    print(f"N={N} M={M} D={D} chain {cb}: {ms:8.3f} ms {cells / ms / 1e6:9.2f} Gcells/s "
          f"{cells * 8 / ms / 1e6:8.1f} GB/s  max rel err on probe rows {err:.2e}{'  FELL BACK' if fell else ''}", flush=True)
    del T2

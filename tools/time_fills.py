"""Fill times of the four output kinds of the halo-block form (log S double / float, V double / float) at N = M = 10^4,
1 and 8 tables: best of 6 per round, median of 3 rounds, kinds alternating in ONE process.
usage: python tools/time_fills.py [N] [D ...]      (repo root, GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from libstb_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Ds = [int(x) for x in sys.argv[2:]] or [1, 8]
L = capi.lib()
for D in Ds:
    a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
    objs = {"S double": capi.DeviceTables(N, N, D=D), "S float": capi.DeviceFloatTables(N, N, D=D),
            "V double": capi.DeviceVTables(N, N, D=D), "V float": capi.DeviceVTables(N, N, D=D, dtype="f32"),
            "V exact": capi.DeviceVTables(N, N, D=D)}
    res = {k: [] for k in objs}
    for r in range(3):
        for k, T in objs.items():
            kw = {"exact": True} if k == "V exact" else {}
            if k == "V exact" and D > 1:
                continue
            T.fill(a, **kw)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                T.fill(a, **kw)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            capi.check(L.stb_fill_status())
            res[k].append(best)
    for k, v in res.items():
        if v:
            cells = objs[k].cells * D
            bytes_ = cells * (4 if "float" in k else 8)
            print(f"N={N} D={D} {k:9s} " + " ".join(f"{x:.3f}" for x in v) + f"  median {np.median(v):.3f} ms  {bytes_ / np.median(v) / 1e6:7.0f} GB/s stored", flush=True)

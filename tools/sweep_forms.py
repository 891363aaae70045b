#!/usr/bin/env python3
"""Time every block-floating form of the fill over a grid of sizes (to set the auto-selection rule).
usage: python tools/sweep_forms.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from libstb_amd import capi, synth

FORMS = [("split", capi.FILL_SPLIT), ("pc", capi.FILL_PC), ("chain", capi.FILL_CHAIN), ("chainx", capi.FILL_CHAINX)]


def timed(T, a, variant, reps=5):
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


for N, M, D in [(200, 50, 1), (200, 50, 5), (1000, 1000, 1), (4000, 4000, 1), (4000, 4000, 8), (4000, 4000, 64),
                (10000, 10000, 1), (10000, 10000, 2), (10000, 10000, 4), (10000, 10000, 8), (10000, 10000, 16),
                (10000, 10000, 32), (10000, 10000, 64), (20000, 2000, 1), (20000, 2000, 16)]:
    a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
    T = capi.DeviceTables(N, M, D=D)
    out = []
    for name, v in FORMS:
        if name == "chainx" and D > 2:
            continue
        out.append(f"{name} {timed(T, a, v):7.3f}")
    T.status()
    print(f"N={N:6d} M={M:6d} D={D:3d}: " + "  ".join(out) + "  ms", flush=True)
    del T

#!/usr/bin/env python3
"""Chain fills under an uneven background load (big copies and fp64 matmuls on another stream):
the tables must stay bit-identical and no block may give up.  usage: soak_load.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstb_amd import capi, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
big_b = torch.empty_like(big_a)
ma = torch.randn(2048, 2048, dtype=torch.float64, device="cuda")
for N, M, D in [(10000, 10000, 1), (4000, 4000, 6), (10000, 10000, 12)]:
    a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
    T = capi.DeviceTables(N, M, D=D)
    T.tables.zero_(); T.fill(a); torch.cuda.synchronize(); T.status()
    ref = T.tables.clone()
    n = 0
    t_case = time.time() + budget / 3
    while time.time() < t_case:
        with torch.cuda.stream(side):      # bursts of unrelated work beside the fills
            if n % 3 == 0:
                big_b.copy_(big_a)
            if n % 5 == 0:
                mb = ma @ ma
        for _ in range(4):
            T.tables.zero_(); T.fill(a)
            if not torch.equal(T.tables, ref):
                print(f"MISMATCH N={N} D={D} after {n} fills", flush=True); sys.exit(1)
            n += 1
        T.status()
    torch.cuda.synchronize()
    print(f"N={N} M={M} D={D}: {n} fills identical under load", flush=True)
print("soak under load ok", flush=True)

#!/usr/bin/env python3
"""Soak test of the chain form: thousands of fills must give bit-identical tables and never give up.
usage: soak_chain.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstb_amd import capi, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cases = [(777, 500, 5), (6000, 900, 2), (10000, 10000, 1), (3000, 3000, 3), (4000, 4000, 8), (10000, 10000, 16), (2000, 2000, 40)]
t_end = time.time() + budget
total = 0
for N, M, D in cases:
    a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
    T = capi.DeviceTables(N, M, D=D)
    T.tables.zero_()   # (the row slack is only partly written: start every fill from the same bytes)
    T.fill(a); torch.cuda.synchronize(); T.status()
    ref = T.tables.clone()
    refS1 = T.S1.clone()
    n = 0
    t_case = time.time() + budget / len(cases)
    while time.time() < t_case:
        for _ in range(20):
            T.tables.zero_()
            T.fill(a)
            # bit-identical, padding included? the slack may differ: compare stored cells through a mask-free trick:
            # the stored cells are deterministic and the slack is written deterministically too
            if not torch.equal(T.tables, ref) or not torch.equal(T.S1, refS1):
                bad = (T.tables != ref).nonzero()
                print(f"MISMATCH N={N} M={M} D={D} after {n} fills: {bad.shape[0]} elements differ, first {bad[0].tolist()}", flush=True)
                sys.exit(1)
            n += 1
        T.status()
    total += n
    print(f"N={N} M={M} D={D}: {n} fills identical", flush=True)
print(f"soak ok: {total} fills", flush=True)

# V table (chain form) and the fused sampler grid: same bits every time
T = capi.DeviceVTables(5000, 5000, D=2)
av = np.array([0.3, 0.8])
T.tables.zero_(); T.fill(av); torch.cuda.synchronize()
ref = T.tables.clone()
n = 0
t_case = time.time() + 20
while time.time() < t_case:
    for _ in range(20):
        T.tables.zero_(); T.fill(av)
        if not torch.equal(T.tables, ref):
            print("MISMATCH in the V fill", flush=True); sys.exit(1)
        n += 1
capi.check(capi.lib().stb_fill_status())
print(f"V fill 5000x5000 D=2: {n} fills identical", flush=True)
L = capi.lib()
g = synth.groups(300, 300, 3000, "wide")
M = max(int(g.t.max()) + 1, 10); N = max(int(g.n.max()) + 1, M)
x = np.ascontiguousarray(synth.discount_grid(64)[:32]); out = np.zeros(32); first = None
h = L.stb_groups_create(g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p),
                        g.n.ctypes.data_as(capi.c_u32_p), g.t.ctypes.data_as(capi.c_u16_p), capi.dp(g.bpar), N, M, 32)
n = 0
t_case = time.time() + 20
while time.time() < t_case:
    capi.check(L.stb_groups_aterms(h, capi.dp(x), 32, capi.dp(out)))
    if first is None:
        first = out.copy()
    elif not np.array_equal(first, out):
        print("MISMATCH in the fused grid", flush=True); sys.exit(1)
    n += 1
L.stb_groups_free(h)
print(f"fused grid (32 discounts x {g.pairs} pairs): {n} evaluations identical", flush=True)

#!/usr/bin/env python3
"""First grid evaluation of a fresh group set (pays the fused set-up) vs the following ones."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libstb_amd import capi, synth
L = capi.lib()
g = synth.groups(1000, 1000, 4000, "wide")
M = max(int(g.t.max()) + 1, 10); N = max(int(g.n.max()) + 1, M)
x = np.ascontiguousarray(synth.discount_grid(64)); out = np.zeros(64)
for fused in ("1", "0", "1", "0"):
    os.environ["STB_ATERMS_FUSED"] = fused
    t0 = time.perf_counter()
    h = L.stb_groups_create(g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p),
                            g.n.ctypes.data_as(capi.c_u32_p), g.t.ctypes.data_as(capi.c_u16_p), capi.dp(g.bpar), N, M, 64)
    t1 = time.perf_counter()
    ts = []
    for _ in range(4):
        t = time.perf_counter(); capi.check(L.stb_groups_aterms(h, capi.dp(x), 64, capi.dp(out))); ts.append(1e3 * (time.perf_counter() - t))
    L.stb_groups_free(h)
    print(f"fused={fused}: create {1e3*(t1-t0):.2f} ms; grid evaluations " + " ".join(f"{v:.2f}" for v in ts) + " ms", flush=True)

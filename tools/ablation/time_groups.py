#!/usr/bin/env python3
"""Where does a samplea call spend its time?  create / evaluate / free of the device group set."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libstb_amd import capi, synth
L = capi.lib()
g = synth.groups(1000, 1000, 4000, "wide")
M = max(int(g.t.max()) + 1, 10); N = max(int(g.n.max()) + 1, M)
x = np.array([0.45]); out = np.zeros(1)
for rep in range(4):
    t0 = time.perf_counter()
    h = L.stb_groups_create(g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p),
                            g.n.ctypes.data_as(capi.c_u32_p), g.t.ctypes.data_as(capi.c_u16_p), capi.dp(g.bpar), N, M, 1)
    t1 = time.perf_counter()
    for _ in range(8):
        capi.check(L.stb_groups_aterms(h, capi.dp(x), 1, capi.dp(out)))
    t2 = time.perf_counter()
    L.stb_groups_free(h)
    t3 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.2f} ms, 8 evaluations {1e3*(t2-t1):.2f} ms, free {1e3*(t3-t2):.2f} ms", flush=True)

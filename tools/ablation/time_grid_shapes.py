"""Fused grid aterms (k_fill_chain as a DOT kernel): device time of the fill+sum by batch D.
usage: python tools/time_grid_shapes.py [n_max ...]     (run from the repo root)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libstb_amd import capi, synth
L = capi.lib()
for nmax in [int(v) for v in sys.argv[1:]] or [4000, 10000]:
    g = synth.groups(1000, 1000, nmax, "wide")
    M = max(int(g.t.max()) + 1, 10); N = max(int(g.n.max()) + 1, M)
    for D in (8, 64):
        x = np.ascontiguousarray(synth.discount_grid(64)[:D]); res = np.zeros(D)
        h = L.stb_groups_create(g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p), g.n.ctypes.data_as(capi.c_u32_p),
                                g.t.ctypes.data_as(capi.c_u16_p), capi.dp(g.bpar), N, M, D)
        mf, ms, mt = C.c_float(), C.c_float(), C.c_float(); best = (1e9, 0, 0)
        for _ in range(6):
            capi.check(L.stb_groups_aterms_timed(h, capi.dp(x), D, capi.dp(res), C.byref(mf), C.byref(ms), C.byref(mt)))
            best = min(best, (mf.value, ms.value, mt.value))
        print(f"n_max={nmax} N={N} M={M} fused grid D={D}: fill+sum {best[0]:.3f} ms, pairs outside the table {best[1]:.3f}, terms {best[2]:.3f}"
              f"  -> {D * g.pairs / (sum(best) * 1e-3):.3e} grid-evals/s (device time)", flush=True)
        L.stb_groups_free(h)

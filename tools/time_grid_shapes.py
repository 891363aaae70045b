import os, sys, time, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from libstb_amd import capi, synth
L = capi.lib()
for nmax in (4000, 10000):
    g = synth.groups(1000, 1000, nmax, "wide")
    M = max(int(g.t.max()) + 1, 10); N = max(int(g.n.max()) + 1, M)
    x = np.ascontiguousarray(synth.discount_grid(64)); res = np.zeros(64)
    for shape in ("4 1 3", "4 1 2", "4 1 1", "2 1 3"):
        c, p, mg = shape.split()
        os.environ.update(STB_CHAIN_C=c, STB_CHAIN_P=p, STB_CHAIN_MG=mg)
        h = L.stb_groups_create(g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p), g.n.ctypes.data_as(capi.c_u32_p), g.t.ctypes.data_as(capi.c_u16_p), capi.dp(g.bpar), N, M, 64)
        mf, ms, mt = C.c_float(), C.c_float(), C.c_float(); best = 1e9
        for _ in range(5):
            capi.check(L.stb_groups_aterms_timed(h, capi.dp(x), 64, capi.dp(res), C.byref(mf), C.byref(ms), C.byref(mt))); best = min(best, mf.value)
        print(f"n_max={nmax} fused D=64 shape {shape}: fill {best:.3f} ms  post[0]={res[0]:.6f}", flush=True)
        L.stb_groups_free(h)

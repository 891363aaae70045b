import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, collections
import orc
from libstb_amd import capi, synth
N = M = 4000; D = 8
a = np.ascontiguousarray(synth.discount_grid(64)[:D])
for dtype in ("f32", "f64"):
    V = capi.DeviceVTables(N, M, D=D, dtype=dtype)
    V.tables.fill_(float("nan"))
    V.fill(a)
    capi.check(capi.lib().stb_fill_status())
    starts = np.cumsum([0] + [min(n - 1, M - 1) for n in range(2, N + 1)])
    for d in range(D):
        got = V.packed_host(d).astype(np.float64)
        nn = np.where(~np.isfinite(got))[0]
        if len(nn):
            ns_ = np.searchsorted(starts, nn, side="right") + 1
            ms_ = nn - starts[ns_ - 2] + 2
            print(dtype, "table", d, "a", a[d], "non-finite:", len(nn), "n", ns_.min(), ns_.max(), "m", ms_.min(), ms_.max(), list(zip(ns_[:8].tolist(), ms_[:8].tolist())), got[nn[:4]])
            print("  (m-2) mod 204:", collections.Counter(((ms_ - 2) % 204).tolist()).most_common(6), " (n-2) mod 48:", collections.Counter(((ns_ - 2) % 48).tolist()).most_common(6))
        else:
            print(dtype, "table", d, "ok")

"""where does the V table taken from the S recurrence's cells differ from the oracle's?  python tools/dbg_v.py N M a"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from libstb_amd import capi
N, M, a = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
D = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dtype = sys.argv[5] if len(sys.argv) > 5 else "f64"
T = capi.DeviceVTables(N, M, D=D, dtype=dtype)
T.tables.fill_(float("nan"))
T.fill([a] * D)
capi.check(capi.lib().stb_fill_status())
got = T.packed_host(D - 1).astype(np.float64)
want = orc.fill_V(a, N, M)
err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
rel = np.abs(got - want) / np.abs(want)
print("max err (abs/max(1,|y|))", err.max(), "max rel", rel.max(), "nan", int(np.isnan(got).sum()))
# map packed index -> (n, m)
starts = np.cumsum([0] + [min(n - 1, M - 1) for n in range(2, N + 1)])
bad = np.argsort(-np.nan_to_num(rel, nan=1e9))[:25]
nn = np.where(~np.isfinite(got))[0]
if len(nn):
    ns_ = np.searchsorted(starts, nn, side="right") + 1
    ms_ = nn - starts[ns_ - 2] + 2
    print("non-finite cells:", len(nn), "n range", ns_.min(), ns_.max(), "m range", ms_.min(), ms_.max(), "first few", list(zip(ns_[:12].tolist(), ms_[:12].tolist())))
    import collections
    print("m mod 204 histogram (top):", collections.Counter(((ms_ - 2) % 204).tolist()).most_common(8))
    print("n-2 mod 48 histogram (top):", collections.Counter(((ns_ - 2) % 48).tolist()).most_common(8))
for k in bad:
    n = int(np.searchsorted(starts, k, side="right")) + 1
    m = int(k - starts[n - 2]) + 2
    print(f"n={n} m={m} got={got[k]!r} want={want[k]!r} rel={rel[k]:.3e}")
cnt = int((rel > 1e-10).sum())
print("cells over 1e-10 relative:", cnt, "of", len(rel))
if cnt:
    ks = np.where(rel > 1e-10)[0]
    ns = np.searchsorted(starts, ks, side="right") + 1
    ms = ks - starts[ns - 2] + 2
    print("n range", ns.min(), ns.max(), "m range", ms.min(), ms.max(), "n-m range", (ns - ms).min(), (ns - ms).max())

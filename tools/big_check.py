#!/usr/bin/env python3
"""One-off parity check at sizes beyond the test suite: usage big_check.py N M [a]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import orc
from libstb_amd import capi
N, M = int(sys.argv[1]), int(sys.argv[2])
a = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
T = capi.DeviceTables(N, M, D=1)
T.tables.fill_(float("nan"))
t0 = time.perf_counter(); T.fill([a]); torch.cuda.synchronize(); T.status()
print(f"GPU fill N={N} M={M} a={a}: {1e3 * (time.perf_counter() - t0):.2f} ms (first call)", flush=True)
t0 = time.perf_counter(); S1, tab = orc.fill_S(a, N, M)
print(f"oracle: {time.perf_counter() - t0:.1f} s", flush=True)
got = T.packed_host(0)
err = np.abs(got - tab) / np.maximum(1.0, np.abs(tab))
print(f"cells {got.size}  finite {np.isfinite(got).all()}  max err {err.max():.3e} at {int(err.argmax())}", flush=True)
assert err.max() <= 1e-10

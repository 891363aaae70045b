"""A/B of the fused grid evaluation between library builds, each in its own process run alternately:
python tools/ab_eval.py NMAX "D,D,.." rounds libA.so libB.so ...      (repo root, GPU box)
prints per library and D the median wall ms of stb_groups_aterms over 10^6 pairs (pairs unchanged between calls)"""
import os
import subprocess
import sys

NMAX, Ds, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
libs = sys.argv[4:]
child = r"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import ctypes as C
import numpy as np
from libstb_amd import capi, synth
NMAX = int(sys.argv[1]); Ds = [int(x) for x in sys.argv[2].split(",")]
L = capi.lib()
g = synth.groups(1000, 1000, NMAX, "wide")
N, M = int(g.n.max()), int(min(g.n.max(), g.t.max()))
u32p, u16p, i32p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), C.POINTER(C.c_int32)
res = []
for D in Ds:
    h = L.stb_groups_create(g.I, g.K.ctypes.data_as(i32p), g.T.ctypes.data_as(u32p), g.n.ctypes.data_as(u32p), g.t.ctypes.data_as(u16p), capi.dp(g.bpar), N, M, D)
    x = np.ascontiguousarray(synth.discount_grid(64)[:D] if D > 1 else np.array([0.5]))
    out = np.zeros(D)
    for _ in range(4):
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(out)))
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(out))); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    res.append(f"{ts[len(ts)//2]:.4f}")
    L.stb_groups_free(h)
print("R " + " ".join(res))
"""
acc = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, STB_LIB_PATH=os.path.abspath(l))
        out = subprocess.run([sys.executable, "-c", child, NMAX, Ds], env=env, capture_output=True, text=True, timeout=600)
        line = [x for x in out.stdout.strip().splitlines() if x.startswith("R ")]
        if not line:
            print(l, "failed:", out.stderr[-300:])
            continue
        acc[l].append([float(x) for x in line[-1].split()[1:]])
for l in libs:
    print(f"NMAX={NMAX} {os.path.basename(l):24s} D = {Ds}: median wall ms per round: " + " | ".join(" ".join(f"{x:.3f}" for x in r) for r in acc[l]), flush=True)

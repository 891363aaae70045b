"""First-light / regression driver for the checkpointed fill (k_fill_ck): small shapes first, each checked
against the CPU oracle cell by cell, then timings against the chain and producer/consumer forms.
usage: python tools/try_ck.py [quick|full|time]      (run from the repo root, on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import orc
from libstb_amd import capi, synth

mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
L = capi.lib()


def check(N, M, D, a=None):
    a = np.round(np.linspace(0.07, 0.93, D), 4) if a is None else np.asarray(a, dtype=np.float64)
    T = capi.DeviceTables(N, M, D=D)
    T.tables.fill_(float("nan"))
    fb = L.stb_fill_fallbacks()
    t0 = time.time()
    T.fill(a, capi.FILL_CK)
    try:
        T.status()
    except capi.StbError as e:
        print(f"N={N} M={M} D={D}: STATUS {e}", flush=True)
        return False
    fell = L.stb_fill_fallbacks() != fb
    worst = 0.0
    bad = None
    for d in range(D):
        S1, tab = orc.fill_S(float(a[d]), N, M)
        got = T.packed_host(d)
        fin = np.isfinite(got)
        err = np.where(fin, np.abs(got - tab) / np.maximum(1.0, np.abs(tab)), 9.9)
        k = int(np.argmax(err))
        if err[k] > worst:
            worst = float(err[k])
            bad = (d, k, got[k], tab[k])
        s1err = orc.max_err(T.S1[d].cpu().numpy(), S1)
        worst = max(worst, s1err)
    ok = worst <= 1e-10 and not fell
    where = ""
    if bad is not None and worst > 1e-10:
        d, k, g, w = bad
        # packed index -> (n, m)
        n, pos = 3, 0
        while True:
            ln = min(n - 2, M - 1)
            if k < pos + ln:
                break
            pos += ln
            n += 1
        nbad = 0
        where = f" first-worst at d={d} n={n} m={2 + k - pos} got={g!r} want={w!r}"
    print(f"N={N} M={M} D={D}: max rel err {worst:.3e} {'ok' if ok else 'FAIL'}{' FELL BACK' if fell else ''} "
          f"({time.time() - t0:.2f}s){where}", flush=True)
    return ok


def timed(T, a, variant, reps=8):
    T.fill(a, variant)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        T.fill(a, variant)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    T.status()
    return best


if mode in ("quick", "full"):
    shapes = [(3, 2, 1), (10, 10, 1), (64, 64, 1), (65, 33, 2), (130, 129, 1), (200, 50, 3), (500, 7, 1), (700, 650, 2),
              (1000, 1000, 1), (1500, 260, 2), (2600, 2600, 1)]
    if mode == "full":
        shapes += [(4000, 4000, 1), (6000, 6000, 1), (3000, 3000, 5), (10000, 10000, 1)]
    allok = True
    for N, M, D in shapes:
        allok = check(N, M, D) and allok
        if not allok and mode == "quick":
            break
    print("ALL OK" if allok else "FAILED", flush=True)
    if not allok:
        sys.exit(1)

if mode in ("time", "full"):
    for N, D in ((10000, 1), (10000, 8), (4000, 1)):
        a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
        T = capi.DeviceTables(N, N, D=D)
        for name, var in (("ck", capi.FILL_CK), ("chain", capi.FILL_CHAIN), ("pc", capi.FILL_PC)):
            ms = timed(T, a, var)
            cells = T.cells * D
            print(f"N={N} D={D} {name:6s} {ms:8.3f} ms  {cells / ms / 1e6:9.2f} Gcells/s  {cells * 8 / ms / 1e6:8.1f} GB/s", flush=True)
        del T

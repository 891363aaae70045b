#!/usr/bin/env python3
"""Where a FRESH set of (n,t) pairs spends its time: what a caller whose pairs change between calls pays
(the reference's Gibbs loop rewrites t[j][i] / T[j] every iteration, test/demo.c:405-445, then calls samplea).

  python tools/time_fresh.py [a|b|c|d ...]   a: samplea fresh against kept   b: pieces at configs[3]'s shape
                                             c: the 64-discount grid at N = 10^4   d: unsorted pairs
"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from libstb_amd import capi, synth
import orc
L = capi.lib()
which = set(sys.argv[1:]) or set("abcd")


def ms(f, *a):
    t0 = time.perf_counter(); r = f(*a); return 1e3 * (time.perf_counter() - t0), r


def create(g, N, M, D):
    return L.stb_groups_create(g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p),
                               g.n.ctypes.data_as(capi.c_u32_p), g.t.ctypes.data_as(capi.c_u16_p), capi.dp(g.bpar), N, M, D)


def ragged(g):
    NP = C.POINTER(C.c_uint32) * g.I; TP = C.POINTER(C.c_uint16) * g.I
    nn, tt = NP(), TP(); off = 0
    for i in range(g.I):
        nn[i] = C.cast(g.n.ctypes.data + 4 * off, C.POINTER(C.c_uint32))
        tt[i] = C.cast(g.t.ctypes.data + 2 * off, C.POINTER(C.c_uint16)); off += int(g.K[i])
    return nn, tt


def samplea(g, nn, tt):
    orc.seed_libc(777, 12345)
    return L.samplea(0.5, g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p), nn, tt, None,
                     capi.dp(g.bpar), None, 1, 0)


g = synth.groups(1000, 1000, 4000, "wide")
M = max(int(g.t.max()) + 1, 10); N = max(int(g.n.max()) + 1, M)

if "a" in which:
    nn, tt = ragged(g)
    for _ in range(3): samplea(g, nn, tt)
    fresh = []
    for r in range(8):
        # one pair moves between calls: a customer joins a table (n+1) -- the bounds stay
        k = 12345 + 977 * r
        if g.n[k] < N - 2: g.n[k] += 1
        fresh.append(ms(samplea, g, nn, tt))
    print("samplea, pairs change between calls   ms:", " ".join(f"{t:.2f}" for t, _ in fresh), "a =", fresh[-1][1], "evals", L.stb_sampler_trace_count(), flush=True)
    same = [ms(samplea, g, nn, tt) for _ in range(5)]
    print("samplea, same pairs (handed over anew) ms:", " ".join(f"{t:.2f}" for t, _ in same), flush=True)
    os.environ["STB_SAMPLEA_CACHE"] = "1"
    samplea(g, nn, tt)
    kept = [ms(samplea, g, nn, tt) for _ in range(5)]
    print("samplea, STB_SAMPLEA_CACHE=1, same pairs ms:", " ".join(f"{t:.2f}" for t, _ in kept), "a =", kept[-1][1], flush=True)
    miss = []
    for r in range(5):
        k = 22345 + 977 * r
        if g.n[k] < N - 2: g.n[k] += 1
        miss.append(ms(samplea, g, nn, tt))
    print("samplea, STB_SAMPLEA_CACHE=1, pairs change ms:", " ".join(f"{t:.2f}" for t, _ in miss), flush=True)
    os.environ.pop("STB_SAMPLEA_CACHE")
    L.stb_sampler_cache_clear()

if "b" in which:
    x3 = np.array([0.4, 0.5, 0.6]); y3 = np.zeros(3); x1 = np.array([0.45]); y1 = np.zeros(1)
    for rep in range(3):
        tc, h = ms(create, g, N, M, 3)
        t3, _ = ms(L.stb_groups_aterms_tables, h, capi.dp(x3), 3, capi.dp(y3))
        t1 = [ms(L.stb_groups_aterms, h, capi.dp(x1), 1, capi.dp(y1))[0] for _ in range(3)]
        L.stb_groups_update_restaurants(h, g.T.ctypes.data_as(capi.c_u32_p), capi.dp(g.bpar))
        tf = [ms(L.stb_groups_aterms, h, capi.dp(x1), 1, capi.dp(y1))[0] for _ in range(4)]
        L.stb_groups_free(h)
        print(f"N={N}: create {tc:.2f}  3 abscissae through tables {t3:.2f}  one, two-pass " + " ".join(f"{v:.2f}" for v in t1) +
              "  one, fused (first pays the lists) " + " ".join(f"{v:.2f}" for v in tf), flush=True)

if "c" in which:
    g2 = synth.groups(1000, 1000, 10000, "wide")
    M2 = max(int(g2.t.max()) + 1, 10); N2 = max(int(g2.n.max()) + 1, M2)
    for D in (64, 8):
        x = np.ascontiguousarray(synth.discount_grid(64)[:D] if D < 64 else synth.discount_grid(64)); out = np.zeros(D)
        for rep in range(2):
            tc, h = ms(create, g2, N2, M2, D)
            te = [ms(L.stb_groups_aterms, h, capi.dp(x), D, capi.dp(out))[0] for _ in range(4)]
            print(f"N={N2} D={D}: create {tc:.2f}  grid evaluations " + " ".join(f"{v:.2f}" for v in te) + f"  new set = {tc + te[0]:.2f} ms", flush=True)
            # the pairs change, the set stays: what a caller's second and later resamples pay
            for r in range(4):
                k = 12345 + 977 * r
                g2.n[k] = min(int(g2.n[k]) + 1, N2 - 2)
                tu, _ = ms(L.stb_groups_update_pairs, h, g2.n.ctypes.data_as(capi.c_u32_p), g2.t.ctypes.data_as(capi.c_u16_p))
                t1, _ = ms(L.stb_groups_aterms, h, capi.dp(x), D, capi.dp(out))
                t2, _ = ms(L.stb_groups_aterms, h, capi.dp(x), D, capi.dp(out))
                print(f"   new pairs: update {tu:.2f} + first evaluation {t1:.2f} = {tu + t1:.2f} ms   (next evaluation {t2:.2f})", flush=True)
            L.stb_groups_free(h)

if "d" in which:
    x3 = np.array([0.4, 0.5, 0.6]); y3 = np.zeros(3); x1 = np.array([0.45]); y1 = np.zeros(1)
    for srt in ("1", "0"):
        os.environ["STB_SORT_PAIRS"] = srt
        for rep in range(2):
            tc, h = ms(create, g, N, M, 3)
            mf, msw, mt = C.c_float(), C.c_float(), C.c_float()
            os.environ["STB_ATERMS_FUSED"] = "0"
            L.stb_groups_aterms_timed(h, capi.dp(x3), 3, capi.dp(y3), C.byref(mf), C.byref(msw), C.byref(mt))
            a = (mf.value, msw.value, mt.value)
            L.stb_groups_aterms_timed(h, capi.dp(x1), 1, capi.dp(y1), C.byref(mf), C.byref(msw), C.byref(mt))
            os.environ.pop("STB_ATERMS_FUSED")
            L.stb_groups_free(h)
            print(f"SORT_PAIRS={srt}: create {tc:.2f}  D=3 fill/sweep/terms {a[0]:.3f} {a[1]:.3f} {a[2]:.3f}   D=1 {mf.value:.3f} {msw.value:.3f} {mt.value:.3f}  y={y1[0]!r}", flush=True)
    os.environ.pop("STB_SORT_PAIRS")

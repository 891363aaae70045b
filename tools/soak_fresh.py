#!/usr/bin/env python3
"""Soak of the new-pairs path (round 5): a kept device set receives one of several sets of pairs over and over
(stb_groups_update_pairs: pinned staging by the library's host threads, DMA in pieces, cell lists from the count slab, no
allocation), is evaluated, and must return the bits a set created from those pairs returned; samplea call after call on
pairs that change must return what it returned for those pairs the first time; nothing may fall back.  Also checked: the
bits do not depend on the order the pairs are handed over in.
usage: python tools/soak_fresh.py [seconds]      (repo root, GPU box)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import orc
from libstb_amd import capi, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
L = capi.lib()
NV = 6                                   # variants of the pairs per case
cases = [(10000, 64), (4000, 3), (10000, 8), (4000, 1), (1500, 5), (10000, 32), (700, 2)]
total = 0
fb0, gb0 = L.stb_fill_fallbacks(), L.stb_groups_fallbacks()
rng = np.random.default_rng(5)
for Nmax, D in cases:
    g = synth.groups(1000, 1000, Nmax, "wide")
    M = max(int(g.t.max()) + 1, 10)
    N = max(int(g.n.max()) + 1, M)
    variants = []
    for v in range(NV):
        n, t = g.n.copy(), g.t.copy()
        idx = rng.integers(0, len(n), 50 * v)
        n[idx] = np.minimum(n[idx] + 1, N - 1)        # customers join tables: bounds unchanged
        if v == NV - 1:
            p = rng.permutation(len(n))              # the same pairs as variant 0, handed over in another order
            n, t = g.n[p].copy(), g.t[p].copy()
        variants.append((n, t))
    x = np.ascontiguousarray(synth.discount_grid(64)[:D] if D > 1 else np.array([0.45]))
    want = []
    for n, t in variants:
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(n), orc.u16p(t), orc.dp(g.bpar), N, M, D)
        assert h, capi.last_error()
        o = np.zeros(D)
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(o)))
        want.append(o)
        L.stb_groups_free(h)
    assert np.array_equal(want[0], want[NV - 1]), "the order of the pairs changed the bits"
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
    out = np.zeros(D)
    cnt, t_end = 0, time.time() + 0.6 * budget / len(cases)
    while time.time() < t_end:
        for _ in range(10):
            v = int(rng.integers(0, NV))
            n, t = variants[v]
            capi.check(L.stb_groups_update_pairs(h, orc.u32p(n), orc.u16p(t)))
            for rep in range(2):          # (the first builds the lists, the second finds them)
                capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(out)))
                if not np.array_equal(out, want[v]):
                    print(f"MISMATCH N={N} D={D} variant {v} after {cnt} updates: {out} != {want[v]}", flush=True)
                    sys.exit(1)
            cnt += 1
    L.stb_groups_free(h)
    total += cnt
    print(f"N={N} M={M} D={D}: {cnt} updates of 10^6 pairs + 2 evaluations each, identical bits", flush=True)

# samplea on changing pairs
g = synth.groups(1000, 1000, 4000, "wide")
NP = C.POINTER(C.c_uint32) * g.I
TP = C.POINTER(C.c_uint16) * g.I
sets = []
for v in range(5):
    n, t = g.n.copy(), g.t.copy()
    idx = rng.integers(0, len(n), 40 * v)
    n[idx] += 1
    if v == 4:
        n[7] = 4400                      # the largest count moves: new table bounds
    nn, tt = NP(), TP()
    off = 0
    for i in range(g.I):
        nn[i] = C.cast(n.ctypes.data + 4 * off, C.POINTER(C.c_uint32))
        tt[i] = C.cast(t.ctypes.data + 2 * off, C.POINTER(C.c_uint16))
        off += int(g.K[i])
    sets.append((n, t, nn, tt))


def draw(s):
    orc.seed_libc(777, 12345)
    a = L.samplea(0.5, g.I, orc.i32p(g.K), orc.u32p(g.T), s[2], s[3], None, orc.dp(g.bpar), None, 1, 0)
    return a, L.stb_sampler_trace_count()


first = [draw(s) for s in sets]
cnt, t_end = 0, time.time() + 0.4 * budget
while time.time() < t_end:
    for _ in range(10):
        v = int(rng.integers(0, len(sets)))
        got = draw(sets[v])
        if got != first[v]:
            print(f"samplea MISMATCH on set {v} after {cnt} calls: {got} != {first[v]}", flush=True)
            sys.exit(1)
        cnt += 1
L.stb_sampler_cache_clear()
print(f"samplea: {cnt} calls on pairs that change from call to call (5 sets, one with other table bounds), identical draws and evaluation counts", flush=True)
if L.stb_fill_fallbacks() != fb0 or L.stb_groups_fallbacks() != gb0:
    print(f"FELL BACK: {capi.last_error()}", flush=True)
    sys.exit(1)
print(f"soak ok: {total} updates, {cnt} samplea calls, none fell back", flush=True)

#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Runs only where oracle/_ref/libstb_ref.so exists (the build container: `make -C oracle` compiles
it from /root/reference in place).  The fixtures are data only -- inputs (or the seed that
regenerates them through libstb_amd.synth) and the reference's outputs -- and travel to the GPU
box, where /root/reference does not exist.

    python tests/golden/gen_golden.py            # rewrites every fixture

Floats inside JSON are stored as C99 hex strings (float.hex) so they round-trip bit-exactly.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import orc  # noqa: E402
from libstb_amd import synth  # noqa: E402

S_STABLE, S_UVTABLE, S_FLOAT, S_ASYMPT = 1, 2, 4, 64


def hx(x: float) -> str:
    return float(x).hex()


def dump(name: str, obj) -> None:
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


def ref_table(R, sp, N, M):
    tab = np.zeros(synth.cells(N, M), dtype=np.float64)
    buf = np.zeros(max(M, 4), dtype=np.float64)
    for n in range(3, N + 1):
        k = R.ref_copy_S_row(sp, n, orc.dp(buf))
        o = orc.row_offset(n, M)
        tab[o:o + k] = buf[:k]
    return tab


def g1_small_tables(R):
    """G1: full S table + S1 at (N=200, M=50) for five discounts (config 1 plumbing case)."""
    out = {}
    for key, a in (("a0.5", 0.5), ("a0.125", 0.125), ("a0.05", 0.05), ("a0.95", 0.95),
                   ("a2_3", 2.0 / 3.0)):
        sp = R.S_make(200, 50, 200, 50, a, S_STABLE)
        out[key + "_table"] = ref_table(R, sp, 200, 50)
        s1 = np.zeros(200)
        R.ref_copy_S1(sp, orc.dp(s1), 200)
        out[key + "_S1"] = s1
        out[key + "_a"] = np.array([a])
        out[key + "_lga"] = np.array([R.ref_lga(sp)])
        R.S_free(sp)
    np.savez_compressed(os.path.join(HERE, "stable_200x50.npz"), **out)
    print("wrote stable_200x50.npz")


def g2_big_probes(R):
    """G2: N=M=4000 and N=M=10000: sparse probes, every row's sum, a few full rows."""
    probes = []
    arrays = {}
    for N, alist in ((4000, (0.5, 0.1, 0.9)), (10000, (0.5, 0.07, 0.93))):
        for a in alist:
            sp = R.S_make(N, N, N, N, a, S_STABLE)
            buf = np.zeros(N, dtype=np.float64)
            rowsum = np.zeros(N + 1, dtype=np.float64)
            for n in range(3, N + 1):
                k = R.ref_copy_S_row(sp, n, orc.dp(buf))
                rowsum[n] = np.sum(buf[:k])  # numpy pairwise sum; recomputed identically in tests
            key = f"N{N}_a{a}"
            arrays[key + "_rowsum"] = rowsum
            for n in (N // 3, N):
                k = R.ref_copy_S_row(sp, n, orc.dp(buf))
                arrays[key + f"_row{n}"] = buf[:k].copy()
            s1 = np.zeros(N)
            R.ref_copy_S1(sp, orc.dp(s1), N)
            arrays[key + "_S1"] = s1
            rows = sorted({3, 4, 5, 10, 100, 1000, N // 2, N - 1, N})
            for n in rows:
                cols = sorted({m for m in (1, 2, 3, 4, n // 2, n - 2, n - 1, n) if 1 <= m <= n})
                for m in cols:
                    probes.append({"N": N, "a": hx(a), "n": n, "m": m, "S": hx(R.S_S(sp, n, m))})
            R.S_free(sp)
    np.savez_compressed(os.path.join(HERE, "stable_big.npz"), **arrays)
    print("wrote stable_big.npz")
    dump("stable_probes.json", probes)


def g3_asympt(R):
    out = []
    for a in (0.0, 0.01, 0.3, 0.5, 0.9):
        sp = R.S_make(20, 10, 20, 10, a, S_STABLE | S_ASYMPT)
        for n in (25, 100, 1000, 10 ** 5, 10 ** 7, 4 * 10 ** 9):
            for m in (1, 2, 5, 10, 1000):
                if m >= n:
                    continue
                out.append({"a": hx(a), "n": n, "m": m, "direct": hx(R.S_asympt(sp, n, m)),
                            # through S_S: N>maxN with S_ASYMPT set (lib/stable.c:952-953)
                            "via_S_S": hx(R.S_S(sp, n, m))})
        R.S_free(sp)
    dump("asympt.json", out)


def g4_extend(R):
    """G4: integer trace of the growth policy, driven through S_S / S_V like a caller would."""
    traces = []
    specs = [
        dict(flags=S_STABLE, init=(20, 10, 600, 600), a=0.5,
             probes=[(30, 5), (120, 40), (121, 119), (300, 100), (599, 300), (599, 598)]),
        dict(flags=S_STABLE, init=(10, 10, 5000, 100), a=0.3,
             probes=[(11, 2), (12, 11), (70, 60), (200, 99), (1000, 100), (4999, 7), (5000, 100)]),
        dict(flags=S_STABLE | S_UVTABLE, init=(50, 20, 400, 200), a=0.7,
             probes=[(48, 18), (49, 19), (60, 19), (100, 100), (399, 150), (400, 200)]),
        dict(flags=S_STABLE, init=(100, 10, 60, 20), a=0.5, probes=[(15, 5), (59, 19)]),
        dict(flags=S_STABLE, init=(5, 3, 8, 4), a=0.5, probes=[(9, 3), (10, 9)]),
    ]
    for s in specs:
        # lazy S1 growth on a fresh table, before any S_extend (lib/stable.c:822-873)
        sp = R.S_make(*s["init"], s["a"], s["flags"])
        fresh = []
        u0, mx = R.ref_usedN(sp), R.ref_maxN(sp)
        for n in (u0, u0 + 5, u0 + 1, u0 + 80, u0 + 79, mx, mx + 1, 0, 1):
            fresh.append({"n": n, "S1": hx(R.S_S1(sp, n)), "usedN1": R.ref_usedN1(sp),
                          "usedN": R.ref_usedN(sp)})
        R.S_free(sp)
        sp = R.S_make(*s["init"], s["a"], s["flags"])
        tr = {"S1_lazy_fresh": fresh,
              "flags": s["flags"], "init": list(s["init"]), "a": hx(s["a"]),
              "made": [R.ref_usedN(sp), R.ref_usedM(sp), R.ref_maxN(sp), R.ref_maxM(sp),
                       R.ref_usedN1(sp), R.ref_startM(sp)],
              "steps": []}
        for (n, m) in s["probes"]:
            val = R.S_S(sp, n, m)
            step = {"n": n, "m": m, "S": hx(val), "usedN": R.ref_usedN(sp),
                    "usedM": R.ref_usedM(sp)}
            if s["flags"] & S_UVTABLE:
                step["V"] = hx(R.S_V(sp, n, m)) if m >= 2 else None
                step["usedN_afterV"] = R.ref_usedN(sp)
                step["usedM_afterV"] = R.ref_usedM(sp)
            tr["steps"].append(step)
        # lazy S1 growth beyond usedN (lib/stable.c:822-873)
        s1 = []
        for n in (R.ref_usedN(sp) + 1, R.ref_usedN(sp) + 7, R.ref_maxN(sp), R.ref_maxN(sp) + 1):
            s1.append({"n": n, "S1": hx(R.S_S1(sp, n)), "usedN1": R.ref_usedN1(sp)})
        tr["S1_lazy"] = s1
        R.S_free(sp)
        traces.append(tr)
    dump("extend_trace.json", traces)


def group_hash(g: synth.Groups) -> str:
    h = hashlib.sha256()
    for arr in (g.K, g.n, g.t, g.T, g.N, g.bpar):
        h.update(np.ascontiguousarray(arr).tobytes())
    return h.hexdigest()


GROUP_SETS = {
    # name: (I, K, n_max, profile)
    "small_wide": (20, 30, 300, "wide"),
    "small_real": (20, 30, 300, "realistic"),
    "mid_wide": (100, 100, 1000, "wide"),
    "big_wide": (1000, 1000, 4000, "wide"),      # SURVEY section 6 / config 4 shape A
    "big_real": (1000, 1000, 4000, "realistic"),
}


def g5_aterms(R):
    out = {}
    for name, (I, K, nmax, prof) in GROUP_SETS.items():
        g = synth.groups(I, K, nmax, prof)
        h = R.ref_aterms_open(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t),
                              orc.dp(g.bpar))
        xs = (0.05, 0.2, 0.4, 0.45, 0.5, 0.6, 0.8, 0.95)
        if name.startswith("big"):
            xs = (0.2, 0.45, 0.5, 0.8)
        vals = [hx(R.ref_aterms_eval(h, x)) for x in xs]
        out[name] = {"I": I, "K": K, "n_max": nmax, "profile": prof, "sha256": group_hash(g),
                     "maxn": R.ref_aterms_maxn(h), "maxt": R.ref_aterms_maxt(h),
                     "x": [hx(x) for x in xs], "aterms": vals}
        R.ref_aterms_close(h)
    dump("aterms.json", out)


def g6_bterms(R):
    out = {}
    for name in ("small_wide", "mid_wide", "big_wide"):
        I, K, nmax, prof = GROUP_SETS[name]
        g = synth.groups(I, K, nmax, prof)
        rows = []
        for apar in (0.1, 0.5, 0.9):
            for Q in (0.05, 3.7):
                for x in (0.01, 0.5, 10.0, 56.6, 2000.0):
                    rows.append({"apar": hx(apar), "Q": hx(Q), "shape": hx(g.shape), "x": hx(x),
                                 "bterms": hx(R.ref_bterms_eval(x, Q, g.shape, g.I, orc.u32p(g.T),
                                                                apar))})
        out[name] = {"sha256": group_hash(g), "rows": rows}
    # shape B: 10^6 restaurants x 1 pair
    g = synth.groups(1000000, 1, 4000, "realistic")
    rows = []
    for x in (1.0, 10.0, 100.0):
        rows.append({"apar": hx(0.5), "Q": hx(0.05), "shape": hx(g.shape), "x": hx(x),
                     "bterms": hx(R.ref_bterms_eval(x, 0.05, g.shape, g.I, orc.u32p(g.T), 0.5))})
    out["shapeB"] = {"I": 1000000, "K": 1, "n_max": 4000, "profile": "realistic",
                     "sha256": group_hash(g), "rows": rows}
    dump("bterms.json", out)


def trace(R):
    n = R.ref_trace_count()
    return {"count": n, "code": R.ref_trace_code(), "xl": hx(R.ref_trace_xl()),
            "xr": hx(R.ref_trace_xr()), "x": [hx(R.ref_trace_x(i)) for i in range(min(n, 1024))],
            "y": [hx(R.ref_trace_y(i)) for i in range(min(n, 1024))]}


def g7_samplers(R):
    out = {"seed_rand": 777, "seed_rand48": 12345, "samplea": [], "sampleb": []}
    for name, (I, K, nmax, prof) in GROUP_SETS.items():
        g = synth.groups(I, K, nmax, prof)
        for a0 in (0.5, 0.1, 0.98):
            if name.startswith("big") and a0 != 0.5:
                continue
            orc.seed_libc(777, 12345)
            r = R.ref_samplea_flat(a0, g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n),
                                   orc.u16p(g.t), orc.dp(g.bpar), 1, 0)
            out["samplea"].append({"set": name, "a_in": hx(a0), "a_out": hx(r),
                                   "trace": trace(R)})
        for (b0, apar) in ((10.0, 0.5), (0.5, 0.2), (2000.0, 0.9), (10.0, 0.0)):
            orc.seed_libc(777, 12345)
            r = R.sampleb(b0, g.I, g.shape, g.scale, orc.u32p(g.N), orc.u32p(g.T), apar, None, 1, 0)
            rec = {"set": name, "b_in": hx(b0), "apar": hx(apar), "b_out": hx(r)}
            if apar != 0.0:
                rec["trace"] = trace(R)
            out["sampleb"].append(rec)
    dump("samplers.json", out)


def g8_arms(R):
    out = []
    cases = [
        (0, 0.3, 0.1, 0.0, -1.0, 2.0), (0, 5.0, 2.0, 0.0, -10.0, 30.0),
        (1, 3.0, 2.0, 0.0, 0.01, 40.0), (1, 1.5, 0.2, 0.0, 0.01, 2000.0),
        (2, 2.0, 5.0, 0.0, 0.01, 0.98), (2, 30.0, 8.0, 0.0, 0.3, 0.7),
        (3, 0.4, 2.0, 500.0, 0.01, 0.98), (3, 0.6, -3.0, 1e5, 0.4, 0.8),
        (4, -2.0, 2.5, 0.0, -6.0, 6.0),
    ]
    for (kind, p0, p1, p2, xl, xr) in cases:
        for seed in (1, 777, 424242):
            for metro in (0, 1):
                if kind == 4 and metro == 0 and False:
                    continue
                orc.seed_libc(seed, 12345)
                xs = np.zeros(256)
                xsamp = C.c_double(0.0)
                ncalls = C.c_int(0)
                xprev = xl + 0.37 * (xr - xl)
                code = R.ref_arms_probe(kind, p0, p1, p2, xl, xr, metro, xprev, C.byref(xsamp),
                                        C.byref(ncalls), orc.dp(xs), 256)
                out.append({"kind": kind, "p": [hx(p0), hx(p1), hx(p2)], "xl": hx(xl),
                            "xr": hx(xr), "seed": seed, "dometrop": metro, "xprev": hx(xprev),
                            "code": code, "xsamp": hx(xsamp.value), "ncalls": ncalls.value,
                            "xs": [hx(v) for v in xs[:min(ncalls.value, 256)]]})
    dump("arms.json", out)


def g9_slice(R):
    out = []
    for (kind, p0, p1, p2, lo, hi, x0) in [(0, 0.3, 0.1, 0.0, -1.0, 2.0, 0.25),
                                           (1, 3.0, 2.0, 0.0, 0.01, 40.0, 1.0),
                                           (2, 2.0, 5.0, 0.0, 0.01, 0.98, 0.2),
                                           (0, 0.3, 0.1, 0.0, -1.0, 2.0, 5.0)]:
        for seed in (12345, 99):
            for loops in (1, 5):
                orc.seed_libc(777, seed)
                x = C.c_double(x0)
                nc = C.c_int(0)
                err = R.ref_slice_probe(kind, p0, p1, p2, lo, hi, C.byref(x), loops, C.byref(nc))
                out.append({"kind": kind, "p": [hx(p0), hx(p1), hx(p2)], "lo": hx(lo), "hi": hx(hi),
                            "x0": hx(x0), "seed48": seed, "loops": loops, "err": err,
                            "x": hx(x.value), "ncalls": nc.value})
    dump("slice.json", out)


def g10_sapprox(R):
    out = []
    for a in (1 / 16, 1 / 8, 3 / 16, 7 / 32, 0.3, 0.5):
        for m in (1, 2, 3, 4, 5):
            for n in (m, m + 1, 10, 100, 2000):
                if n < m:
                    continue
                out.append({"a": hx(np.float32(a)), "n": n, "m": m, "S_approx": hx(R.S_approx(n, m, a)),
                            "S_approx_da": hx(R.S_approx_da(n, m, a))})
    dig = [{"x": hx(x), "digamma": hx(R.digammaRN(x))} for x in (0.1, 0.5, 1.0, 4.9, 5.0, 5.1, 37.5, 1e4)]
    dump("sapprox.json", {"rows": out, "digamma": dig})


def g11_uv(R):
    """Next-row fixture: V table, float storage, U/UV accessors at (200,50)."""
    arrays = {}
    acc = []
    for key, a in (("a0.5", 0.5), ("a0.05", 0.05), ("a0.95", 0.95)):
        sp = R.S_make(200, 50, 200, 50, a, S_STABLE | S_UVTABLE)
        buf = np.zeros(64)
        rows = []
        for n in range(2, 201):
            k = R.ref_copy_V_row(sp, n, orc.dp(buf))
            rows.append(buf[:k].copy())
        arrays[key + "_V"] = np.concatenate(rows)
        for (n, m) in ((5, 2), (5, 5), (5, 6), (100, 1), (100, 30), (150, 48), (198, 48)):
            acc.append({"a": hx(a), "n": n, "m": m, "U": hx(R.S_U(sp, n, m)) if m >= 1 else None,
                        "V": hx(R.S_V(sp, n, m)) if m >= 2 else None,
                        "UV": hx(R.S_UV(sp, n, m))})
        R.S_free(sp)
        sp = R.S_make(200, 50, 200, 50, a, S_STABLE | S_UVTABLE | S_FLOAT)
        fbuf = np.zeros(64, dtype=np.float32)
        srows, vrows = [], []
        for n in range(2, 201):
            if n >= 3:
                k = R.ref_copy_Sf_row(sp, n, fbuf.ctypes.data_as(orc.c_float_p))
                srows.append(fbuf[:k].copy())
            k = R.ref_copy_Vf_row(sp, n, fbuf.ctypes.data_as(orc.c_float_p))
            vrows.append(fbuf[:k].copy())
        arrays[key + "_Sf"] = np.concatenate(srows)
        arrays[key + "_Vf"] = np.concatenate(vrows)
        R.S_free(sp)
    np.savez_compressed(os.path.join(HERE, "uv_200x50.npz"), **arrays)
    print("wrote uv_200x50.npz")
    dump("uv_access.json", acc)


def g12_rng(R):
    out = {"seed48": 12345}
    orc.seed_libc(777, 12345)
    out["gaussian"] = [hx(R.gsl_rng_gaussian_ziggurat(1.0)) for _ in range(64)]
    for a in (0.3, 1.0, 2.5, 100.0):
        orc.seed_libc(777, 12345)
        out[f"gamma_{a}"] = [hx(R.gsl_rng_gamma(a)) for _ in range(32)]
    orc.seed_libc(777, 12345)
    out["beta_10_500"] = [hx(R.gsl_rng_beta(10.0, 500.0)) for _ in range(32)]
    dump("rng.json", out)


GRID_PROBE_D = (0, 7, 31, 63)  # members of the 64-point grid a_d = 0.05 + 0.9 (d + 1/2) / 64


def g13_grid_tables(R):
    """configs[2]: the 64-discount batch at N=M=10000.  For four members of the grid (first and last
    of rank 0's share of eight, last of rank 3's, last of rank 7's): every row's sum, two full rows,
    S1 and sparse probes."""
    N = 10000
    grid = synth.discount_grid(64)
    arrays, probes = {}, []
    for d in GRID_PROBE_D:
        a = float(grid[d])
        sp = R.S_make(N, N, N, N, a, S_STABLE)
        buf = np.zeros(N, dtype=np.float64)
        rowsum = np.zeros(N + 1, dtype=np.float64)
        for n in range(3, N + 1):
            k = R.ref_copy_S_row(sp, n, orc.dp(buf))
            rowsum[n] = np.sum(buf[:k])
        key = f"d{d}"
        arrays[key + "_a"] = np.array([a])
        arrays[key + "_rowsum"] = rowsum
        for n in (N // 3, N):
            k = R.ref_copy_S_row(sp, n, orc.dp(buf))
            arrays[key + f"_row{n}"] = buf[:k].copy()
        s1 = np.zeros(N)
        R.ref_copy_S1(sp, orc.dp(s1), N)
        arrays[key + "_S1"] = s1
        for n in sorted({3, 4, 5, 10, 100, 1000, N // 2, N - 1, N}):
            for m in sorted({m for m in (1, 2, 3, 4, n // 2, n - 2, n - 1, n) if 1 <= m <= n}):
                probes.append({"d": d, "a": hx(a), "n": n, "m": m, "S": hx(R.S_S(sp, n, m))})
        R.S_free(sp)
    np.savez_compressed(os.path.join(HERE, "stable_grid10k.npz"), **arrays)
    print("wrote stable_grid10k.npz")
    dump("stable_grid10k_probes.json", probes)


def g14_grid_aterms(R):
    """configs[4]: aterms at ALL 64 grid discounts for the 10^6-pair set of config 3/4 (n < 4000, wide
    t), and at four of them for the same shape with n < 10000 (the bench's N=M=10000 grid)."""
    grid = synth.discount_grid(64)
    out = {}
    for name, (I, K, nmax, prof), ds in (("big_wide", GROUP_SETS["big_wide"], range(64)),
                                        ("big10k_wide", (1000, 1000, 10000, "wide"), GRID_PROBE_D)):
        g = synth.groups(I, K, nmax, prof)
        h = R.ref_aterms_open(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar))
        ds = list(ds)
        vals = [hx(R.ref_aterms_eval(h, float(grid[d]))) for d in ds]
        out[name] = {"I": I, "K": K, "n_max": nmax, "profile": prof, "sha256": group_hash(g),
                     "maxn": R.ref_aterms_maxn(h), "maxt": R.ref_aterms_maxt(h), "d": ds,
                     "x": [hx(float(grid[d])) for d in ds], "aterms": vals}
        R.ref_aterms_close(h)
    dump("aterms_grid64.json", out)


def g15_samplea2(R):
    """8f-4: the S-free discount sampler.  From the reference built with -DSAMPLEA_M
    (oracle/_ref/libstb_ref_m.so): for three group sets and two starting discounts the table sizes
    samplea2 sampled (drand48 stream 12345), ARMS' abscissae and values (rand stream 777), the draw,
    and aterms2 at fixed abscissae for that partition."""
    if not orc.have_ref_m():
        sys.exit("oracle/_ref/libstb_ref_m.so missing: run `make -C oracle`")
    RM = orc.ref_m()
    out = {"seed_rand": 777, "seed_rand48": 12345, "runs": []}
    for name in ("small_wide", "small_real", "mid_wide"):
        I, K, nmax, prof = GROUP_SETS[name]
        g = synth.groups(I, K, nmax, prof)
        maxn, maxt = int(g.n.max()) + 1, int(g.t.max()) + 1
        for a0 in (0.5, 0.15):
            sp = RM.S_make(maxn, maxt, maxn, maxt, a0, S_STABLE)
            orc.seed_libc(777, 12345)
            r = RM.ref_samplea2_flat(a0, sp, g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t),
                                     orc.dp(g.bpar), 1, 0)
            m = np.array([RM.ref_m_get(i) for i in range(RM.ref_m_size())], dtype=np.uint16)
            n = RM.ref_trace_count()
            rec = {"set": name, "a_in": hx(a0), "a_out": hx(r), "maxn": maxn, "maxt": maxt,
                   "m_sha256": hashlib.sha256(m.tobytes()).hexdigest(), "m_count": int(m.shape[0]),
                   "m_head": [int(v) for v in m[:64]],
                   "trace": {"count": n, "code": RM.ref_trace_code(), "xl": hx(RM.ref_trace_xl()), "xr": hx(RM.ref_trace_xr()),
                             "x": [hx(RM.ref_trace_x(i)) for i in range(n)], "y": [hx(RM.ref_trace_y(i)) for i in range(n)]},
                   "aterms2": [{"x": hx(x), "y": hx(RM.ref_aterms2_eval(x, g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n),
                                                                       orc.u16p(g.t), orc.dp(g.bpar), orc.u16p(m)))}
                               for x in (0.05, 0.3, 0.5, 0.7, 0.95)]}
            if m.shape[0] <= 5000:
                rec["m"] = [int(v) for v in m]
            out["runs"].append(rec)
            RM.S_free(sp)
    dump("samplea2.json", out)


def g16_samplea_slice(R):
    """samplea's OTHER sampler: the slice branch (lib/samplea.c:216-221, lib/sslice.c:33-80), from the reference built with
    PSAMPLE_ARS off for lib/samplea.c (oracle/_ref/libstb_ref_slice.so): the draw, the number of posterior evaluations,
    every abscissa and value, the bracket handed to SliceSimple, under srand48(12345)"""
    if not orc.have_ref_slice():
        sys.exit("oracle/_ref/libstb_ref_slice.so missing: run `make -C oracle`")
    RS = orc.ref_slice()
    out = {"seed_rand": 777, "seed_rand48": 12345, "runs": []}
    for name, a0, loops in (("small_wide", 0.5, 1), ("small_wide", 0.1, 3), ("small_real", 0.3, 2), ("small_real", 0.9, 1),
                            ("mid_wide", 0.5, 2), ("mid_wide", 0.25, 1), ("mid_wide", 0.97, 3), ("big_real", 0.5, 1)):
        I, K, nmax, prof = GROUP_SETS[name]
        g = synth.groups(I, K, nmax, prof)
        orc.seed_libc(777, 12345)
        r = RS.ref_samplea_flat(a0, g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), loops, 0)
        n = RS.ref_trace_count()
        out["runs"].append({"set": name, "a_in": hx(a0), "loops": loops, "a_out": hx(r),
                            "trace": {"count": n, "code": RS.ref_trace_code(), "xl": hx(RS.ref_trace_xl()), "xr": hx(RS.ref_trace_xr()),
                                      "x": [hx(RS.ref_trace_x(i)) for i in range(n)], "y": [hx(RS.ref_trace_y(i)) for i in range(n)]}})
    dump("samplers_slice.json", out)


def main():
    if not orc.have_ref():
        sys.exit("oracle/_ref/libstb_ref.so missing: run `make -C oracle` where /root/reference exists")
    R = orc.ref()
    gens = [g1_small_tables, g2_big_probes, g3_asympt, g4_extend, g5_aterms, g6_bterms,
            g7_samplers, g8_arms, g9_slice, g10_sapprox, g11_uv, g12_rng, g13_grid_tables, g14_grid_aterms, g15_samplea2, g16_samplea_slice]
    want = sys.argv[1:]
    for g in gens:
        if not want or g.__name__.split("_")[0] in want:
            g(R)


if __name__ == "__main__":
    main()

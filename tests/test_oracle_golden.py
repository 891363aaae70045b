"""The CPU oracle (oracle/stb_oracle.c) against the committed golden fixtures, which were dumped
from the real reference by tests/golden/gen_golden.py.  Pins the checker itself; no GPU.

Same-libm machines reproduce the fixtures bit-for-bit; the assertion uses the path's parity
metric |x-y| <= 1e-10*max(1,|y|) so a different glibc still passes, and reports how
many values matched exactly.
"""
import json
import os

import numpy as np
import pytest

import orc
from libstb_amd import synth

fh = float.fromhex


def load_json(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("key", ["a0.5", "a0.125", "a0.05", "a0.95", "a2_3"])
def test_small_table_config1(golden_dir, key):
    """configs[0]: S_make N=200 M=50 -- the reference's own CPU-runnable case."""
    z = np.load(os.path.join(golden_dir, "stable_200x50.npz"))
    a = float(z[key + "_a"][0])
    S1, tab = orc.fill_S(a, 200, 50)
    assert tab.shape[0] == 8526 == synth.cells(200, 50)
    assert orc.close(tab, z[key + "_table"], 1e-13)
    assert orc.close(S1, z[key + "_S1"], 1e-13)


def test_config1_known_values(golden_dir):
    """values printed by the reference's test/list.c at a=0.5, N=200."""
    L = orc.oracle()
    S1, tab = orc.fill_S(0.5, 200, 50)
    S = lambda n, m: L.orc_S_S(orc.dp(tab), orc.dp(S1), 200, 50, n, m)
    assert abs(S(200, 2) - 855.40717151108788) < 1e-9
    assert abs(S(200, 25) - 815.82736174046067) < 1e-9
    assert abs(S(200, 49) - 744.08298941125418) < 1e-9
    assert abs(L.orc_S_asympt(0.5, 200, 25) - 814.80290541542593) < 1e-9


@pytest.mark.parametrize("N,a", [(4000, 0.5), (4000, 0.1), (4000, 0.9)])
def test_big_table_rows(golden_dir, N, a):
    z = np.load(os.path.join(golden_dir, "stable_big.npz"))
    key = f"N{N}_a{a}"
    S1, tab = orc.fill_S(a, N, N)
    rowsum = np.zeros(N + 1)
    for n in range(3, N + 1):
        o = orc.row_offset(n, N)
        rowsum[n] = np.sum(tab[o:o + orc.row_len(n, N)])
    assert orc.close(rowsum, z[key + "_rowsum"], 1e-12)
    for n in (N // 3, N):
        o = orc.row_offset(n, N)
        assert orc.close(tab[o:o + orc.row_len(n, N)], z[key + f"_row{n}"], 1e-13)
    assert orc.close(S1, z[key + "_S1"], 1e-13)


def test_big_table_probes_4000(golden_dir):
    L = orc.oracle()
    probes = [p for p in load_json(golden_dir, "stable_probes.json") if p["N"] == 4000]
    cache = {}
    for p in probes:
        a = fh(p["a"])
        if a not in cache:
            cache[a] = orc.fill_S(a, 4000, 4000)
        S1, tab = cache[a]
        got = L.orc_S_S(orc.dp(tab), orc.dp(S1), 4000, 4000, p["n"], p["m"])
        assert orc.close(got, fh(p["S"]), 1e-13), p


def test_asympt(golden_dir):
    L = orc.oracle()
    for r in load_json(golden_dir, "asympt.json"):
        got = L.orc_S_asympt(fh(r["a"]), r["n"], r["m"])
        assert orc.close(got, fh(r["direct"]), 1e-13), r


def test_extend_policy_trace(golden_dir):
    L = orc.oracle()
    import ctypes as C
    for tr in load_json(golden_dir, "extend_trace.json"):
        iN, iM, mN, mM = (C.c_uint(v) for v in tr["init"])
        L.orc_make_clamp(C.byref(iN), C.byref(iM), C.byref(mN), C.byref(mM))
        assert [iN.value, iM.value, mN.value, mM.value] == tr["made"][:4]
        usedN, usedM = iN.value, iM.value
        S_UV = bool(tr["flags"] & 2)
        for st in tr["steps"]:
            n, m = st["n"], st["m"]
            # S_S grows only when T>usedM || N>usedN and inside the max bounds (lib/stable.c:950-965)
            if n != m and m != 1 and not (n < m or m == 0) and (m > usedM or n > usedN) \
                    and not (n > mN.value or m > mM.value):
                nN, nM = C.c_uint(), C.c_uint()
                L.orc_extend_policy(usedN, usedM, mN.value, mM.value, n + 1, m + 1, C.byref(nN),
                                    C.byref(nM))
                usedN, usedM = nN.value, nM.value
            assert (usedN, usedM) == (st["usedN"], st["usedM"]), (tr["init"], st)
            if S_UV and m >= 2:
                # S_V grows when m>=usedM-1 || n>=usedN-1 (lib/stable.c:903)
                if (m >= usedM - 1 or n >= usedN - 1) and not (n > mN.value or m > mM.value):
                    nN, nM = C.c_uint(), C.c_uint()
                    L.orc_extend_policy(usedN, usedM, mN.value, mM.value, n + 1, m + 1,
                                        C.byref(nN), C.byref(nM))
                    usedN, usedM = nN.value, nM.value
                assert (usedN, usedM) == (st["usedN_afterV"], st["usedM_afterV"]), st


def _groups(spec):
    return synth.groups(spec["I"], spec["K"], spec["n_max"], spec["profile"])


@pytest.mark.parametrize("name", ["small_wide", "small_real", "mid_wide"])
def test_aterms(golden_dir, name):
    L = orc.oracle()
    spec = load_json(golden_dir, "aterms.json")[name]
    g = _groups(spec)
    import ctypes as C
    mn, mt = C.c_int(), C.c_int()
    L.orc_scan_bounds(g.I, orc.i32p(g.K), orc.u32p(g.n), orc.u16p(g.t), C.byref(mn), C.byref(mt))
    assert (mn.value, mt.value) == (spec["maxn"], spec["maxt"])
    # S_make(maxn,maxt,maxn,maxt) clamps (lib/samplea.c:60, lib/stable.c:118-129)
    M = max(mt.value, 10)
    N = max(mn.value, M)
    scratch = np.zeros(synth.cells(N, M) + N)
    for x, want in zip(spec["x"], spec["aterms"]):
        got = L.orc_aterms(fh(x), g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t),
                           orc.dp(g.bpar), N, M, orc.dp(scratch))
        assert orc.close(got, fh(want), 1e-13), (name, x)


def test_aterms_big_config4(golden_dir):
    """config 4 shape A: 10^6 pairs against the N=M=4000 table (one abscissa; ~0.3 s)."""
    L = orc.oracle()
    spec = load_json(golden_dir, "aterms.json")["big_wide"]
    g = _groups(spec)
    N, M = spec["maxn"], spec["maxt"]
    assert N == 4000 and 3900 < M <= 4000
    scratch = np.zeros(synth.cells(N, M) + N)
    got = L.orc_aterms(fh(spec["x"][1]), g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n),
                       orc.u16p(g.t), orc.dp(g.bpar), N, M, orc.dp(scratch))
    assert orc.close(got, fh(spec["aterms"][1]), 1e-13)


def test_bterms(golden_dir):
    L = orc.oracle()
    d = load_json(golden_dir, "bterms.json")
    sets = {"small_wide": (20, 30, 300, "wide"), "mid_wide": (100, 100, 1000, "wide"),
            "big_wide": (1000, 1000, 4000, "wide"), "shapeB": (1000000, 1, 4000, "realistic")}
    for name, (I, K, nmax, prof) in sets.items():
        g = synth.groups(I, K, nmax, prof)
        for r in d[name]["rows"]:
            got = L.orc_bterms(fh(r["x"]), fh(r["Q"]), fh(r["shape"]), g.I, orc.u32p(g.T),
                               fh(r["apar"]))
            assert orc.close(got, fh(r["bterms"]), 1e-13), (name, r)


def test_sapprox_known_answer(golden_dir):
    """closed form for m<=4 agrees with the recurrence for dyadic a with m*a<1."""
    L = orc.oracle()
    rows = load_json(golden_dir, "sapprox.json")["rows"]
    for r in rows:
        a = fh(r["a"])
        got = L.orc_S_approx(r["n"], r["m"], a)
        want = fh(r["S_approx"])
        if np.isnan(want):
            assert np.isnan(got)
        else:
            assert orc.close(got, want, 1e-13), r
    for a in (1 / 16, 1 / 8, 3 / 16, 7 / 32):
        S1, tab = orc.fill_S(a, 2000, 10)
        for m in (2, 3, 4):
            if m * a >= 1:
                continue
            for n in (10, 100, 2000):
                tv = L.orc_S_S(orc.dp(tab), orc.dp(S1), 2000, 10, n, m)
                assert abs(L.orc_S_approx(n, m, a) - tv) <= 1e-11 * max(1, abs(tv))


def test_uv_table(golden_dir):
    z = np.load(os.path.join(golden_dir, "uv_200x50.npz"))
    for key, a in (("a0.5", 0.5), ("a0.05", 0.05), ("a0.95", 0.95)):
        v = orc.fill_V(a, 200, 50)
        assert orc.close(v, z[key + "_V"], 1e-13)


def test_identities():
    """S_S(n,n)=0; S_S(n,1)=lgamma(n-a)-lgamma(1-a); S^3_2=log(3-3a)."""
    from math import lgamma, log
    L = orc.oracle()
    for a in (0.37, 2.0 / 3.0):
        S1, tab = orc.fill_S(a, 600, 40)
        S = lambda n, m: L.orc_S_S(orc.dp(tab), orc.dp(S1), 600, 40, n, m)
        assert S(17, 17) == 0.0
        assert abs(S(600, 1) - (lgamma(600 - a) - lgamma(1 - a))) < 1e-9
        assert abs(S(3, 2) - log(3 - 3 * a)) < 1e-14
        assert S(5, 7) == -np.inf and S(5, 0) == -np.inf


@pytest.mark.parametrize("d", [0, 63])
def test_grid_table_config3(golden_dir, d):
    """configs[2]: members of the 64-discount grid at N=M=10000 (two of the four fixtures here; the
    GPU suite checks all four)."""
    N = 10000
    z = np.load(os.path.join(golden_dir, "stable_grid10k.npz"))
    a = float(z[f"d{d}_a"][0])
    assert a == float(synth.discount_grid(64)[d])
    S1, tab = orc.fill_S(a, N, N)
    rowsum = np.zeros(N + 1)
    for n in range(3, N + 1):
        o = orc.row_offset(n, N)
        rowsum[n] = np.sum(tab[o:o + orc.row_len(n, N)])
    assert orc.close(rowsum, z[f"d{d}_rowsum"], 1e-12)
    for n in (N // 3, N):
        o = orc.row_offset(n, N)
        assert orc.close(tab[o:o + orc.row_len(n, N)], z[f"d{d}_row{n}"], 1e-13)
    assert orc.close(S1, z[f"d{d}_S1"], 1e-13)
    L = orc.oracle()
    probes = [p for p in load_json(golden_dir, "stable_grid10k_probes.json") if p["d"] == d]
    assert len(probes) > 40
    for p in probes:
        got = L.orc_S_S(orc.dp(tab), orc.dp(S1), N, N, p["n"], p["m"])
        assert orc.close(got, fh(p["S"]), 1e-13), p


def test_grid_aterms_config5(golden_dir):
    """configs[4]: aterms on the 10^6-pair set at members of the 64-discount grid (every eighth here;
    the GPU suite checks all 64)."""
    L = orc.oracle()
    spec = load_json(golden_dir, "aterms_grid64.json")["big_wide"]
    assert spec["d"] == list(range(64))
    g = _groups(spec)
    N, M = spec["maxn"], spec["maxt"]
    scratch = np.zeros(synth.cells(N, M) + N)
    grid = synth.discount_grid(64)
    for d in range(0, 64, 8):
        assert fh(spec["x"][d]) == float(grid[d])
        got = L.orc_aterms(float(grid[d]), g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t),
                           orc.dp(g.bpar), N, M, orc.dp(scratch))
        assert orc.close(got, fh(spec["aterms"][d]), 1e-13), d


def _uniforms(count):
    """the drand48 stream samplea2 draws its one uniform per pair from (seed 12345)"""
    import ctypes as C

    libc = C.CDLL(None)
    libc.drand48.restype = C.c_double
    orc.seed_libc(777, 12345)
    return np.array([libc.drand48() for _ in range(count)])


@pytest.mark.parametrize("run", range(6))
def test_samplea2_partition_and_aterms2(golden_dir, run):
    """8f-4: the oracle's restatement of samplea2's table-size sampling (lib/samplea.c:295-320) and of
    aterms2 (:85-150) against the reference built with -DSAMPLEA_M"""
    import hashlib

    L = orc.oracle()
    rec = load_json(golden_dir, "samplea2.json")["runs"][run]
    g = synth.groups(*{"small_wide": (20, 30, 300, "wide"), "small_real": (20, 30, 300, "realistic"),
                       "mid_wide": (100, 100, 1000, "wide")}[rec["set"]])
    a0 = fh(rec["a_in"])
    N, M = rec["maxn"], max(rec["maxt"], 10)
    N = max(N, M)
    S1, tab = orc.fill_S(a0, N, M)
    npairs = int(np.sum((g.t > 1) & (g.t < g.n)))
    u = _uniforms(npairs)
    m = np.zeros(rec["m_count"] + 1, dtype=np.uint16)
    cnt = L.orc_partition(a0, orc.dp(tab), orc.dp(S1), N, M, g.I, orc.i32p(g.K), orc.u32p(g.n), orc.u16p(g.t), orc.dp(u),
                          orc.u16p(m))
    assert cnt == rec["m_count"]
    m = np.ascontiguousarray(m[:cnt])
    assert [int(v) for v in m[:64]] == rec["m_head"]
    assert hashlib.sha256(m.tobytes()).hexdigest() == rec["m_sha256"]
    if "m" in rec:
        assert [int(v) for v in m] == rec["m"]
    for p in rec["aterms2"]:
        got = L.orc_aterms2(fh(p["x"]), g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar),
                            orc.u16p(m))
        assert orc.close(got, fh(p["y"]), 1e-13), p


def test_streamed_rows_are_the_stored_tables_rows():
    """orc_rows_stream (rows of tables too large to keep: the scale tests' checker) against orc_fill_S, bit for bit, in
    wide, narrow and one-thread shapes"""
    import orc as _orc

    for a, N, M, thr in ((0.5, 700, 700, 5), (0.23, 900, 120, 8), (0.0, 300, 300, 1), (0.9, 400, 50, 3)):
        S1, tab = _orc.fill_S(a, N, M)
        want_rows = [3, 4, 5, N // 3, N // 2, N - 1, N]
        got = _orc.rows_stream(a, N, M, want_rows, threads=thr)
        for n in want_rows:
            ln = min(n - 1, M) - 1
            o = _orc.row_offset(n, M)
            assert np.array_equal(got[n], tab[o:o + ln]), (a, N, M, n)

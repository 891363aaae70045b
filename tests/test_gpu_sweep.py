"""GPU parity of the posterior pieces: S_S gather-sum (K3), restaurant terms and bterms (K4), and
whole aterms evaluations, against the CPU oracle and the golden values from the reference."""
import json
import os

import numpy as np
import pytest

import orc
from libstb_amd import capi, synth

pytestmark = pytest.mark.gpu
fh = float.fromhex
TOL = 1e-10


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def groups_of(spec):
    return synth.groups(spec["I"], spec["K"], spec["n_max"], spec["profile"])


def bounds(spec):
    M = max(spec["maxt"], 10)
    return max(spec["maxn"], M), M


@pytest.mark.parametrize("name", ["small_wide", "small_real", "mid_wide", "big_wide", "big_real"])
def test_aterms_vs_reference_golden(golden_dir, name):
    """whole evaluations: device table build + sweep + restaurant terms == reference's aterms(x)"""
    L = capi.lib()
    spec = load(golden_dir, "aterms.json")[name]
    g = groups_of(spec)
    N, M = bounds(spec)
    x = np.array([fh(v) for v in spec["x"]])
    want = np.array([fh(v) for v in spec["aterms"]])
    D = len(x)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar),
                            N, M, D)
    assert h, capi.last_error()
    try:
        out = np.zeros(D)
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(out)))   # batched: all abscissae at once
        assert orc.close(out, want, TOL), (out, want)
        again = np.zeros(D)
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(again)))
        assert np.array_equal(again, out)                                 # deterministic run to run
        if D > 2:                                                         # ... and independent of the batch
            capi.check(L.stb_groups_aterms(h, capi.dp(x[1:].copy()), D - 1, capi.dp(again)))
            assert np.array_equal(again[:D - 1], out[1:])
        one = np.zeros(1)
        for d in range(D):   # one at a time: the stored-table + gather path (a grid is summed inside the fill)
            capi.check(L.stb_groups_aterms(h, capi.dp(x[d:d + 1].copy()), 1, capi.dp(one)))
            assert orc.close(one[0], want[d], TOL)
            assert orc.close(one[0], out[d], 1e-13)
    finally:
        L.stb_groups_free(h)


def test_config5_grid64_vs_reference(golden_dir):
    """configs[4]: the 64-discount grid x 10^6 pairs in ONE call (the fused evaluation: the chain fill
    sums count * log S itself) against the reference's aterms at all 64 discounts; then the same grid
    through stored tables + gather (two-pass), which must agree with both."""
    L = capi.lib()
    spec = load(golden_dir, "aterms_grid64.json")["big_wide"]
    g = groups_of(spec)
    N, M = bounds(spec)
    x = np.ascontiguousarray(synth.discount_grid(64))
    assert [fh(v) for v in spec["x"]] == list(x)
    want = np.array([fh(v) for v in spec["aterms"]])
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, 64)
    assert h, capi.last_error()
    try:
        before = L.stb_fill_fallbacks()
        out = np.zeros(64)
        capi.check(L.stb_groups_aterms(h, capi.dp(x), 64, capi.dp(out)))
        assert L.stb_fill_fallbacks() == before
        assert orc.close(out, want, TOL), np.max(np.abs(out - want) / np.abs(want))
    finally:
        L.stb_groups_free(h)
    os.environ["STB_ATERMS_FUSED"] = "0"
    try:
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, 64)
        assert h, capi.last_error()
        two = np.zeros(64)
        capi.check(L.stb_groups_aterms(h, capi.dp(x), 64, capi.dp(two)))
        L.stb_groups_free(h)
    finally:
        os.environ.pop("STB_ATERMS_FUSED", None)
    assert orc.close(two, want, TOL)
    assert orc.close(two, out, 1e-12)


def test_grid_at_10000_vs_reference(golden_dir):
    """the bench's second metric at N=M=10000: 10^6 pairs with n < 10000 against four members of the
    64-discount grid, evaluated as part of the full 64-point grid call"""
    L = capi.lib()
    spec = load(golden_dir, "aterms_grid64.json")["big10k_wide"]
    g = groups_of(spec)
    N, M = bounds(spec)
    assert N == 10000
    x = np.ascontiguousarray(synth.discount_grid(64))
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, 64)
    assert h, capi.last_error()
    try:
        out = np.zeros(64)
        capi.check(L.stb_groups_aterms(h, capi.dp(x), 64, capi.dp(out)))
        for d, xs, want in zip(spec["d"], spec["x"], spec["aterms"]):
            assert fh(xs) == x[d]
            assert orc.close(out[d], fh(want), TOL), (d, out[d], fh(want))
        one = np.zeros(1)
        capi.check(L.stb_groups_aterms(h, capi.dp(x[31:32].copy()), 1, capi.dp(one)))   # stored table + gather
        assert orc.close(one[0], fh(spec["aterms"][spec["d"].index(31)]), TOL)
    finally:
        L.stb_groups_free(h)


def test_sweep_and_terms_separately_vs_oracle():
    O = orc.oracle()
    g = synth.groups(60, 50, 600, "wide")
    # inject edge pairs: n=1 (skipped), n=t (0), t=1 (S1), t=0 and t>n (-inf is checked separately)
    g.n[:4] = [1, 9, 30, 2]
    g.t[:4] = [1, 9, 1, 1]
    N, M = 600, 600
    a = np.array([0.2, 0.5, 0.8])
    T = capi.DeviceTables(N, M, D=3)
    T.fill(a)
    dg = capi.DeviceGroups(g)
    got = capi.sweep(T, dg).cpu().numpy()
    terms = capi.restaurant_terms(a, dg).cpu().numpy()
    for d in range(3):
        S1, tab = orc.fill_S(a[d], N, M)
        zero_T = np.zeros_like(g.T)
        # oracle sum of the pair terms only: restaurant terms vanish when T=0 and bpar cancels
        pair_sum = 0.0
        vals = []
        off = 0
        for i in range(g.I):
            for k in range(g.K[i]):
                n, t = int(g.n[off + k]), int(g.t[off + k])
                if n > 1:
                    vals.append(O.orc_S_S(orc.dp(tab), orc.dp(S1), N, M, n, t))
            off += g.K[i]
        pair_sum = float(np.sum(np.array(vals)))
        assert orc.close(got[d], pair_sum, 1e-12)
        full = O.orc_aterms_sum(a[d], g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t),
                                orc.dp(g.bpar), orc.dp(tab), orc.dp(S1), N, M)
        assert orc.close(got[d] + terms[d], full, TOL)


def test_sweep_invalid_pairs_give_minus_infinity():
    g = synth.groups(4, 8, 50, "wide")
    g.n[3], g.t[3] = 5, 9     # t > n  -> log 0 (lib/stable.c:948-949)
    T = capi.DeviceTables(60, 60, D=2)
    T.fill([0.3, 0.6])
    got = capi.sweep(T, capi.DeviceGroups(g)).cpu().numpy()
    assert np.all(np.isneginf(got))
    g.t[3] = 0                # t == 0 -> log 0 as well
    got = capi.sweep(T, capi.DeviceGroups(g)).cpu().numpy()
    assert np.all(np.isneginf(got))


def test_sweep_empty_and_ragged():
    T = capi.DeviceTables(40, 40, D=1)
    T.fill([0.5])
    g = synth.groups(3, 5, 30, "wide")
    g.K[:] = [0, 15, 0]       # ragged: all pairs in the middle restaurant
    dg = capi.DeviceGroups(g)
    S1, tab = orc.fill_S(0.5, 40, 40)
    O = orc.oracle()
    want = O.orc_aterms_sum(0.5, 3, orc.i32p(g.K), orc.u32p(np.zeros(3, dtype=np.uint32)), orc.u32p(g.n),
                            orc.u16p(g.t), orc.dp(g.bpar), orc.dp(tab), orc.dp(S1), 40, 40)
    got = capi.sweep(T, dg).cpu().numpy()[0]
    # T=0: restaurant terms are lgamma(b/x)-lgamma(b/x)=0, so the oracle total is the pair sum
    assert orc.close(got, want, 1e-12)


def test_bterms_vs_reference_golden(golden_dir):
    d = load(golden_dir, "bterms.json")
    sets = {"small_wide": (20, 30, 300, "wide"), "mid_wide": (100, 100, 1000, "wide"),
            "big_wide": (1000, 1000, 4000, "wide"), "shapeB": (1000000, 1, 4000, "realistic")}
    for name, (I, K, nmax, prof) in sets.items():
        g = synth.groups(I, K, nmax, prof)
        dg = capi.DeviceGroups(g)
        rows = d[name]["rows"]
        # group rows sharing (apar, Q): one batched call per group
        keyed = {}
        for r in rows:
            keyed.setdefault((r["apar"], r["Q"], r["shape"]), []).append(r)
        for (apar, Q, shape), rs in keyed.items():
            x = np.array([fh(r["x"]) for r in rs])
            got = capi.bterms(x, fh(Q), fh(shape), fh(apar), dg).cpu().numpy()
            want = np.array([fh(r["bterms"]) for r in rs])
            assert orc.close(got, want, TOL), (name, got, want)


def test_config4_shape_sweep_is_deterministic():
    """10^6 pairs against the N=M=4000 table: two runs agree to the last bit"""
    g = synth.groups(1000, 1000, 4000, "wide")
    T = capi.DeviceTables(4000, 4000, D=1)
    T.fill([0.45])
    dg = capi.DeviceGroups(g)
    a = capi.sweep(T, dg).cpu().numpy()
    b = capi.sweep(T, dg).cpu().numpy()
    assert a[0] == b[0] and np.isfinite(a[0])


def test_fused_aterms_equals_table_then_sweep(monkeypatch):
    """stb_groups_aterms with the chain form never stores the tables (count * log S summed inside the
    fill, over the occurring cells only or over a count slab); both must agree with the stored-table +
    gather path, edge pairs included"""
    L = capi.lib()
    g = synth.groups(80, 60, 900, "wide")
    n, t = g.n.copy(), g.t.copy()
    n[0], t[0] = 1, 1          # skipped
    n[1], t[1] = 77, 77        # t = n: contributes 0
    n[2], t[2] = 500, 1        # t = 1: S1
    n[3], t[3] = 3, 2          # first cell of the table
    n[4], t[4] = 900, 2
    n[5], t[5] = 900, 899      # next to the diagonal
    n[6:40], t[6:40] = 400, 123  # one cell many times
    N, M = 900, 900
    x = np.array([0.11, 0.5, 0.83])
    outs = []
    for fused, sparse in (("1", "1"), ("0", "1"), ("1", "0")):   # sparse DOT, two-pass, dense DOT
        monkeypatch.setenv("STB_ATERMS_FUSED", fused)
        monkeypatch.setenv("STB_ATERMS_SPARSE", sparse)
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(n), orc.u16p(t), orc.dp(g.bpar),
                                N, M, len(x))
        assert h, capi.last_error()
        try:
            out = np.zeros(len(x))
            capi.check(L.stb_groups_aterms(h, capi.dp(x), len(x), capi.dp(out)))
            outs.append(out)
        finally:
            L.stb_groups_free(h)
    assert np.all(np.isfinite(outs[0]))
    assert orc.close(outs[0], outs[1], 1e-12), (outs[0], outs[1])
    assert orc.close(outs[2], outs[1], 1e-12), (outs[2], outs[1])
    monkeypatch.setenv("STB_ATERMS_SPARSE", "1")
    # an out-of-bounds pair makes the whole sum -inf on both paths
    n[7], t[7] = 50, 60
    for fused in ("1", "0"):
        monkeypatch.setenv("STB_ATERMS_FUSED", fused)
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(n), orc.u16p(t), orc.dp(g.bpar),
                                N, M, len(x))
        try:
            out = np.zeros(len(x))
            capi.check(L.stb_groups_aterms(h, capi.dp(x), len(x), capi.dp(out)))
            assert np.all(np.isneginf(out))
        finally:
            L.stb_groups_free(h)


@pytest.mark.parametrize("sum_C", ["2", "3", "4"])
def test_fused_aterms_in_the_halo_block_form(monkeypatch, golden_dir, sum_C):
    """the default for a grid of discounts: the summing fill as k_fill_hb<C, DOT> (a spine that walks blocks of rows
    alone + tile workers that sum their tiles' listed cells; cell lists keyed by (tile, group of 8 rows)), with strips of
    2 columns per lane (a set of few discounts: the faster walk), 3 or 4.  Same
    sums as the chain form to rounding -- against the reference's aterms golden values at 1e-10, against the
    chain form, run to run bit for bit -- with edge pairs, several tables, and a set whose chain-form lists are
    built later on the same object (both layouts live side by side)."""
    L = capi.lib()
    monkeypatch.setenv("STB_HB_DOT_C", sum_C)
    UC = {"2": 80, "3": 144, "4": 208}[sum_C]
    for name in ("mid_wide", "small_realistic"):
        spec = load(golden_dir, "aterms.json").get(name)
        if spec is None:
            continue
        g = groups_of(spec)
        N, M = bounds(spec)
        xs = np.array([fh(v) for v in spec["x"]])
        want = np.array([fh(v) for v in spec["aterms"]])
        D = min(len(xs), 8)
        if D < 2:
            continue
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
        assert h, capi.last_error()
        try:
            x = np.ascontiguousarray(xs[:D])
            hb1, hb2, ch = np.zeros(D), np.zeros(D), np.zeros(D)
            monkeypatch.setenv("STB_ATERMS_GRID", "0")   # (beyond 24 discounts sparse pairs take the grid form since round 4: tests below)
            monkeypatch.setenv("STB_ATERMS_HB", "1")
            fb = L.stb_fill_fallbacks()
            capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(hb1)))
            capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(hb2)))
            assert L.stb_fill_fallbacks() == fb
            monkeypatch.setenv("STB_ATERMS_HB", "0")
            capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(ch)))      # the other layout, same object
            assert np.array_equal(hb1, hb2)
            assert orc.close(hb1, want[:D], 1e-10), orc.max_err(hb1, want[:D])
            assert orc.close(hb1, ch, 1e-12), (hb1, ch)
        finally:
            L.stb_groups_free(h)
    # edge pairs (t = 1, t = n, n = 1, many pairs on one cell, next to the diagonal, first and last rows and
    # columns of tiles), 3 tables of 900 x 900
    g = synth.groups(80, 60, 900, "wide")
    n, t = g.n.copy(), g.t.copy()
    n[0], t[0] = 1, 1
    n[1], t[1] = 77, 77
    n[2], t[2] = 500, 1
    n[3], t[3] = 3, 2
    n[4], t[4] = 900, 2
    n[5], t[5] = 900, 899
    n[6:40], t[6:40] = 400, 123
    n[40], t[40] = 49, 2       # last row of block 0
    n[41], t[41] = 50, 2       # first row of block 1
    n[42], t[42] = 300, UC + 1    # last own column of strip 0 (UC own columns from column 2)
    n[43], t[43] = 300, UC + 2    # first own column of strip 1
    n[44], t[44] = UC + 3, UC + 2  # next to the diagonal in strip 1's first block
    n[45], t[45] = 301, UC + 2    # ... and in an odd row of a group (one step below the staged row)
    x = np.array([0.11, 0.5, 0.83])
    outs = []
    monkeypatch.setenv("STB_ATERMS_GRID", "0")
    for hb in ("1", "0"):
        monkeypatch.setenv("STB_ATERMS_HB", hb)
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(n), orc.u16p(t), orc.dp(g.bpar), 900, 900, 3)
        assert h, capi.last_error()
        try:
            out = np.zeros(3)
            capi.check(L.stb_groups_aterms(h, capi.dp(x), 3, capi.dp(out)))
            outs.append(out)
        finally:
            L.stb_groups_free(h)
    assert np.all(np.isfinite(outs[0])) and orc.close(outs[0], outs[1], 1e-12), outs


def test_fused_aterms_dense_pairs_take_the_count_slab(monkeypatch):
    """more pairs than a third of the table's cells: the fused evaluation uses the count slab (every
    cell's log) instead of the per-item cell lists; same sums as the stored-table path"""
    L = capi.lib()
    g = synth.groups(90, 80, 70, "wide")      # 7200 pairs on a 70 x 70 table
    N, M = 70, 70
    x = np.array([0.2, 0.6])
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("STB_ATERMS_FUSED", fused)
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar),
                                N, M, len(x))
        assert h, capi.last_error()
        try:
            out = np.zeros(len(x))
            capi.check(L.stb_groups_aterms(h, capi.dp(x), len(x), capi.dp(out)))
            outs.append(out)
        finally:
            L.stb_groups_free(h)
    assert np.all(np.isfinite(outs[0]))
    assert orc.close(outs[0], outs[1], 1e-12), (outs[0], outs[1])


def _edge_pairs(C):
    """pairs on every kind of edge of the self-summing form's strips (C columns per lane: 80 or 208 own columns)"""
    UC = (64 - 48 // C) * C
    g = synth.groups(80, 60, 900, "wide")
    n, t = g.n.copy(), g.t.copy()
    n[0], t[0] = 1, 1          # n = 1: skipped (lib/samplea.c:78)
    n[1], t[1] = 77, 77        # t = n: log 1
    n[2], t[2] = 500, 1        # t = 1: column 1, which strip 0's halo computes along
    n[3], t[3] = 3, 2          # the table's first cell
    n[4], t[4] = 900, 2
    n[5], t[5] = 900, 899      # last row, next to the diagonal
    n[6:40], t[6:40] = 400, 123    # many pairs on one cell
    n[40], t[40] = 49, 2       # last row of block 0
    n[41], t[41] = 50, 2       # first row of block 1
    nr = 300 if UC + 2 < 300 else UC + 120   # (a row below the diagonal at the strip edge: 8 columns per lane make strips of 464)
    n[42], t[42] = nr, UC + 1      # last own column of strip 0
    n[43], t[43] = nr, UC + 2      # first own column of strip 1
    n[44], t[44] = UC + 3, UC + 2  # next to the diagonal in strip 1's first block
    n[60], t[60] = nr + 1, UC + 2  # the same column in an odd row of a group (taken from the staged row above it)
    n[61], t[61] = nr + 2, UC + 2
    n[64], t[64] = nr + 3, UC + 2  # ... three rows below a staged one (every 4th row staged)
    n[65], t[65] = nr + 3, UC + 4
    n[62], t[62] = 52, 3
    n[63], t[63] = 53, 3
    n[45], t[45] = 2, 1        # S^2_1 = 1 - a: a negative log
    n[46], t[46] = 13, 1
    n[47:60], t[47:60] = 900, 1    # column 1 in the last row, several times
    return g, n, t


HELP_ALL = {"STB_GRID_HELP_NW": "1"}                           # every tile with a listed cell is a helper job
HELP_FEW = {"STB_GRID_HELP_NW": "1", "STB_GRID_JOBS": "3"}     # ... but room for three: the threshold moves up
HELP_OFF = {"STB_GRID_HELP": "0"}


@pytest.mark.parametrize("C,P,G,PH,env", [(2, 4, 24, 1, {}), (2, 7, 12, 3, {}), (2, 2, 8, 1, {}), (4, 4, 12, 1, {}), (4, 7, 8, 4, {}),
                                          (4, 3, 16, 2, {}), (4, 4, 24, 5, {}), (2, 4, 24, 1, HELP_ALL), (4, 4, 24, 1, HELP_ALL),
                                          (4, 3, 12, 1, HELP_ALL), (4, 4, 24, 1, HELP_FEW), (4, 4, 24, 1, HELP_OFF), (4, 4, 8, 3, HELP_ALL),
                                          (8, 2, 16, 1, {}), (8, 3, 24, 1, {}), (8, 2, 8, 2, {}), (8, 4, 12, 1, HELP_ALL), (8, 2, 16, 1, HELP_ALL),
                                          (8, 2, 16, 1, {"STB_GRID_K": "2"}), (8, 1, 24, 3, HELP_OFF)])
def test_fused_aterms_the_spine_sums(monkeypatch, C, P, G, PH, env):
    """the default for a grid since round 4: k_grid_hb<C, G> (grid_hb.hip), whose walking waves stage every other
    row of a group of G in LDS and sum their own strip's listed cells (column 1 included; no tile workers, no S1
    vector, no gather pass), in PH launches over bands of blocks -- against stored tables + gather at 1e-12, bit
    for bit run to run, at the ends of samplea's bracket (A_MIN = 0.01, A_MAX = 0.98, lib/psample.h:89-94), in
    every strip shape; and with tiles left to helper waves (a strip's own wave only walks them and leaves a record;
    waves whose strips have ended sum them): all tiles, a few, none"""
    L = capi.lib()
    monkeypatch.setenv("STB_ATERMS_GRID", "1")
    for k, v in (("C", C), ("P", P), ("G", G), ("PHASES", PH)):
        monkeypatch.setenv("STB_GRID_" + k, str(v))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g, n, t = _edge_pairs(C)
    x = np.array([0.01, 0.11, 0.5, 0.83, 0.98])
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("STB_ATERMS_FUSED", fused)
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(n), orc.u16p(t), orc.dp(g.bpar), 900, 900, 5)
        assert h, capi.last_error()
        try:
            fb = L.stb_fill_fallbacks()
            out, again = np.zeros(5), np.zeros(5)
            capi.check(L.stb_groups_aterms(h, capi.dp(x), 5, capi.dp(out)))
            capi.check(L.stb_groups_aterms(h, capi.dp(x), 5, capi.dp(again)))
            assert L.stb_fill_fallbacks() == fb
            assert np.array_equal(out, again)
            outs.append(out)
        finally:
            L.stb_groups_free(h)
    assert np.all(np.isfinite(outs[0])) and orc.close(outs[0], outs[1], 1e-12), outs
    # ... and against the oracle's aterms, pair by pair in the reference's order
    O = orc.oracle()
    for d in (0, 2, 4):
        S1, tab = orc.fill_S(float(x[d]), 900, 900)
        want = O.orc_aterms_sum(float(x[d]), g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(n), orc.u16p(t), orc.dp(g.bpar),
                                orc.dp(tab), orc.dp(S1), 900, 900)
        assert orc.close(outs[0][d], want, TOL), (d, outs[0][d], want)


@pytest.mark.parametrize("order", [(8, 4, 2, 8), (3, 8, 5)])
def test_fused_aterms_helper_jobs_one_set_many_grid_sizes(monkeypatch, order):
    """the helper jobs of the grid form are chosen when a set's cell lists are built (at its first evaluation) and sized
    for Dmax tables: later evaluations of the same set with fewer or more discounts use the same jobs -- every one equals
    stored tables + gather"""
    L = capi.lib()
    monkeypatch.setenv("STB_ATERMS_GRID", "1")
    monkeypatch.setenv("STB_GRID_C", "4")
    monkeypatch.setenv("STB_GRID_HELP_NW", "1")
    g, n, t = _edge_pairs(4)
    grid = np.ascontiguousarray(synth.discount_grid(64)[::8])
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(n), orc.u16p(t), orc.dp(g.bpar), 900, 900, 8)
    assert h, capi.last_error()
    try:
        fb = L.stb_fill_fallbacks()
        for D in order:
            x = np.ascontiguousarray(grid[:D])
            got, want = np.zeros(D), np.zeros(D)
            capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(got)))
            capi.check(L.stb_groups_aterms_tables(h, capi.dp(x), D, capi.dp(want)))
            assert orc.close(got, want, 1e-12), (D, got, want)
        assert L.stb_fill_fallbacks() == fb
    finally:
        L.stb_groups_free(h)


def test_fused_aterms_the_spine_sums_log_zero_pairs(monkeypatch):
    """a pair outside the table's support has S_S = log 0 (lib/stable.c:948-949): the sum is -inf, as the
    reference's is, in the form that never gathers"""
    L = capi.lib()
    monkeypatch.setenv("STB_ATERMS_GRID", "1")
    g = synth.groups(20, 30, 300, "wide")
    n, t = g.n.copy(), g.t.copy()
    n[7], t[7] = 5, 9
    x = np.array([0.3, 0.6])
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(n), orc.u16p(t), orc.dp(g.bpar), 300, 300, 2)
    assert h, capi.last_error()
    try:
        out = np.zeros(2)
        capi.check(L.stb_groups_aterms(h, capi.dp(x), 2, capi.dp(out)))
        assert np.all(np.isneginf(out)), out
    finally:
        L.stb_groups_free(h)


def test_fused_aterms_the_spine_sums_vs_reference(golden_dir, monkeypatch):
    """... against the reference's own aterms values (golden), small and mid-sized sets (dense ones included:
    groups of more than 63 listed cells are taken from the lists in CSR form)"""
    L = capi.lib()
    monkeypatch.setenv("STB_ATERMS_GRID", "1")
    for name in ("small_wide", "small_real", "mid_wide"):
        spec = load(golden_dir, "aterms.json")[name]
        g = groups_of(spec)
        N, M = bounds(spec)
        x = np.array([fh(v) for v in spec["x"]])
        want = np.array([fh(v) for v in spec["aterms"]])
        D = len(x)
        if D < 2:
            continue
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
        assert h, capi.last_error()
        try:
            out = np.zeros(D)
            assert L.stb_fill_tuning(N, M, D, None, None, None) in (3, 6)
            capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(out)))
            assert orc.close(out, want, TOL), (name, orc.max_err(out, want))
        finally:
            L.stb_groups_free(h)

"""GPU parity of the checkpointed fill form (k_fill_ck: recurrence-only spine + tile workers, tools/ablation/fill_ck.hip;
round 3's first design, superseded by the halo-block form and no longer in the product library) through the C ABI.
Build and run on a GPU box from the repo root:

    make -C tools/ablation
    STB_LIB_PATH=$PWD/libstb_amd/lib/libstb_amd_ablation.so python -m pytest tools/ablation/test_fill_ck.py -q

Parity metric |x-y| <= 1e-10*max(1,|y|)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p_ in (ROOT, os.path.join(ROOT, "tests")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import numpy as np
import pytest

import orc
from libstb_amd import capi, synth

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not capi.lib().stb_has_ablation(), reason="load libstb_amd_ablation.so through STB_LIB_PATH")]
fh = float.fromhex


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


@pytest.fixture
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def groups_of(spec):
    return synth.groups(spec["I"], spec["K"], spec["n_max"], spec["profile"])


def bounds(spec):
    M = max(spec["maxt"], 10)
    return max(spec["maxn"], M), M
TOL = 1e-10


def _check_tables(T, a, N, M):
    for d in range(T.D):
        S1, tab = orc.fill_S(float(a[d]), N, M)
        got = T.packed_host(d)
        assert np.all(np.isfinite(got)), (N, M, d)
        assert orc.close(got, tab, TOL), (N, M, d, orc.max_err(got, tab))
        assert orc.close(T.S1[d].cpu().numpy(), S1, TOL)


@pytest.mark.parametrize("C,P", [(1, 1), (1, 4), (2, 1), (2, 2), (2, 4), (4, 1), (4, 2), (4, 4)])
def test_ck_geometries_agree(monkeypatch, C, P):
    """every strip shape (columns per lane, spine waves per workgroup) computes the same tables, no
    wave gives up waiting (several wave strips, several workgroups per table, several tables)"""
    monkeypatch.setenv("STB_CK_C", str(C))
    monkeypatch.setenv("STB_CK_P", str(P))
    L = capi.lib()
    a = np.array([0.05, 0.5, 0.93])
    T = capi.DeviceTables(900, 700, D=3)
    T.tables.fill_(float("nan"))
    before = L.stb_fill_fallbacks()
    T.fill(a, capi.FILL_CK)
    T.status()
    assert L.stb_fill_fallbacks() == before
    _check_tables(T, a, 900, 700)


@pytest.mark.parametrize("rows,period", [(48, 0), (96, 24), (128, 0), (64, 16)])
def test_ck_block_and_period_lengths(monkeypatch, rows, period):
    """tiles of other heights and shorter renormalisation periods: checkpoints fall on period
    boundaries whatever the two are"""
    monkeypatch.setenv("STB_CK_BLOCK_ROWS", str(rows))
    if period:
        monkeypatch.setenv("STB_FILL_P", str(period))
    a = np.array([0.2, 0.8])
    T = capi.DeviceTables(1300, 1100, D=2)
    T.tables.fill_(float("nan"))
    T.fill(a, capi.FILL_CK)
    T.status()
    _check_tables(T, a, 1300, 1100)


def test_ck_random_shapes_vs_oracle():
    rng = np.random.default_rng(20261004)
    L = capi.lib()
    before = L.stb_fill_fallbacks()
    for _ in range(14):
        N = int(rng.integers(3, 2600))
        M = int(rng.integers(2, N + 1))
        D = int(rng.integers(1, 5))
        a = np.round(rng.uniform(0.0, 0.99, size=D), 6)
        T = capi.DeviceTables(N, M, D=D)
        T.tables.fill_(float("nan"))
        T.fill(a, capi.FILL_CK)
        T.status()
        _check_tables(T, a, N, M)
    assert L.stb_fill_fallbacks() == before


def test_ck_more_spine_workgroups_than_compute_units(monkeypatch):
    """one-wave spine workgroups of 64 columns, 40 tables of 3000 columns: 1880 spine workgroups for a
    grid that holds a few hundred at once -- a strip only ever waits for a strip with a smaller ticket"""
    monkeypatch.setenv("STB_CK_C", "1")
    monkeypatch.setenv("STB_CK_P", "1")
    D, N = 40, 3000
    L = capi.lib()
    a = synth.discount_grid(64)[:D]
    T = capi.DeviceTables(N, N, D=D)
    T.tables.fill_(float("nan"))
    before = L.stb_fill_fallbacks()
    T.fill(a, capi.FILL_CK)
    T.status()
    assert L.stb_fill_fallbacks() == before       # (finished, not rescued by the other form after a time-out)
    T2 = capi.DeviceTables(N, N, D=D)
    T2.fill(a, capi.FILL_PC)
    for d in (0, 7, D - 1):
        got = T.packed_host(d)
        assert np.all(np.isfinite(got))
        assert orc.max_err(got, T2.packed_host(d)) <= TOL


def test_ck_gives_up_instead_of_hanging(monkeypatch):
    """every wait of the checkpointed form is bounded: with the bound at zero whoever has to wait
    records an error and everybody runs to the end; stb_fill_status then repeats the fill with the
    producer/consumer form -- or reports the failure when that is switched off"""
    L = capi.lib()
    S1, tab = orc.fill_S(0.5, 3000, 3000)
    monkeypatch.setenv("STB_CHAIN_TIMEOUT_MS", "0")
    monkeypatch.setenv("STB_CHAIN_NO_FALLBACK", "1")
    T = capi.DeviceTables(3000, 3000, D=1)
    T.fill([0.5], capi.FILL_CK)
    with pytest.raises(capi.StbError):
        T.status()
    monkeypatch.delenv("STB_CHAIN_NO_FALLBACK")
    before = L.stb_fill_fallbacks()
    T.tables.fill_(float("nan"))
    T.fill([0.5], capi.FILL_CK)
    T.status()
    assert L.stb_fill_fallbacks() == before + 1
    assert orc.max_err(T.packed_host(0), tab) <= TOL
    monkeypatch.delenv("STB_CHAIN_TIMEOUT_MS")
    T.tables.fill_(float("nan"))
    T.fill([0.5], capi.FILL_CK)
    T.status()
    assert L.stb_fill_fallbacks() == before + 1
    assert orc.max_err(T.packed_host(0), tab) <= TOL


@pytest.mark.parametrize("variant", [capi.FILL_CK])
def test_10000_full_table_vs_oracle(variant):
    """configs[1] cell by cell: all 49 985 001 cells of the N=M=10000, a=0.5 table against the oracle's
    (reference recurrence lib/stable.c:380-388), in the form stb_fill_S picks (halo blocks) and in the checkpointed one"""
    N, a = 10000, 0.5
    T = capi.DeviceTables(N, N, D=1)
    T.tables.fill_(float("nan"))
    T.fill([a], variant)
    T.status()
    S1, tab = orc.fill_S(a, N, N)
    got = T.packed_host(0)
    assert got.shape[0] == 49985001
    err = np.abs(got - tab) / np.maximum(1.0, np.abs(tab))
    assert np.all(np.isfinite(got))
    assert float(err.max()) <= TOL, float(err.max())
    assert orc.close(T.S1[0].cpu().numpy(), S1, TOL)




def test_fused_aterms_in_the_checkpointed_form(monkeypatch, golden_dir):
    """STB_ATERMS_CK=1: the summing fill as recurrence-only spine + tile workers (k_fill_ck<.., DOT>, cell lists
    keyed from column 2).  Same sums as the chain form to rounding -- against the reference's aterms golden values
    at 1e-10, against the chain form, run to run bit for bit -- with edge pairs, several tables and a set whose
    chain-form lists are built later on the same object (both layouts live side by side)."""
    L = capi.lib()
    monkeypatch.setenv("STB_ATERMS_HB", "0")       # (the halo-block form is the default for a grid: its own test below)
    spec = load(golden_dir, "aterms.json")["mid_wide"]
    g = groups_of(spec)
    N, M = bounds(spec)
    xs = np.array([fh(v) for v in spec["x"]])
    want = np.array([fh(v) for v in spec["aterms"]])
    D = min(len(xs), 8)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
    assert h, capi.last_error()
    try:
        x = np.ascontiguousarray(xs[:D])
        ck1, ck2, ch = np.zeros(D), np.zeros(D), np.zeros(D)
        monkeypatch.setenv("STB_ATERMS_CK", "1")
        fb = L.stb_fill_fallbacks()
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(ck1)))
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(ck2)))
        assert L.stb_fill_fallbacks() == fb
        monkeypatch.setenv("STB_ATERMS_CK", "0")
        capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(ch)))      # the other layout, same object
        assert np.array_equal(ck1, ck2)
        assert orc.close(ck1, want[:D], 1e-10), orc.max_err(ck1, want[:D])
        assert orc.close(ck1, ch, 1e-12), (ck1, ch)
    finally:
        L.stb_groups_free(h)
    # edge pairs (t = 1, t = n, n = 1, many pairs on one cell, next to the diagonal), 3 tables of 900 x 900
    g = synth.groups(80, 60, 900, "wide")
    n, t = g.n.copy(), g.t.copy()
    n[0], t[0] = 1, 1
    n[1], t[1] = 77, 77
    n[2], t[2] = 500, 1
    n[3], t[3] = 3, 2
    n[4], t[4] = 900, 2
    n[5], t[5] = 900, 899
    n[6:40], t[6:40] = 400, 123
    x = np.array([0.11, 0.5, 0.83])
    outs = []
    monkeypatch.setenv("STB_ATERMS_GRID", "0")
    for ck in ("1", "0"):
        monkeypatch.setenv("STB_ATERMS_CK", ck)
        h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(n), orc.u16p(t), orc.dp(g.bpar), 900, 900, 3)
        assert h, capi.last_error()
        try:
            out = np.zeros(3)
            capi.check(L.stb_groups_aterms(h, capi.dp(x), 3, capi.dp(out)))
            outs.append(out)
        finally:
            L.stb_groups_free(h)
    assert np.all(np.isfinite(outs[0])) and orc.close(outs[0], outs[1], 1e-12), outs



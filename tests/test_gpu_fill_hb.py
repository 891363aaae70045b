"""GPU parity of the halo-block fill form (k_fill_hb: a spine that walks blocks of rows behind a halo,
alone, + tile workers; libstb_amd/csrc/fill_hb.hip) through the C ABI, against the oracle's table
(reference recurrence lib/stable.c:380-388).  Parity metric |x-y| <= 1e-10*max(1,|y|)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import orc
from libstb_amd import capi, synth

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _check_tables(T, a, N, M):
    for d in range(T.D):
        S1, tab = orc.fill_S(float(a[d]), N, M)
        got = T.packed_host(d)
        assert np.all(np.isfinite(got)), (N, M, d)
        assert orc.close(got, tab, TOL), (N, M, d, orc.max_err(got, tab))
        assert orc.close(T.S1[d].cpu().numpy(), S1, TOL)


@pytest.mark.parametrize("C,P,R", [(1, 6, 32), (1, 3, 16), (2, 7, 48), (2, 1, 32), (2, 4, 16), (4, 7, 48), (4, 2, 24), (4, 5, 8)])
def test_hb_geometries_agree(monkeypatch, C, P, R):
    """every strip shape (columns per lane, spine waves per workgroup, rows per block) computes the same
    tables and no wave gives up waiting (several strips, several workgroups per table, several tables)"""
    monkeypatch.setenv("STB_HB_C", str(C))
    monkeypatch.setenv("STB_HB_P", str(P))
    monkeypatch.setenv("STB_HB_ROWS", str(R))
    L = capi.lib()
    a = np.array([0.05, 0.5, 0.93])
    T = capi.DeviceTables(900, 700, D=3)
    T.tables.fill_(float("nan"))
    before = L.stb_fill_fallbacks()
    T.fill(a, capi.FILL_HB)
    T.status()
    assert L.stb_fill_fallbacks() == before
    _check_tables(T, a, 900, 700)


def test_hb_random_shapes_vs_oracle():
    rng = np.random.default_rng(20261005)
    L = capi.lib()
    before = L.stb_fill_fallbacks()
    for _ in range(14):
        N = int(rng.integers(3, 2600))
        M = int(rng.integers(2, N + 1))
        D = int(rng.integers(1, 5))
        a = np.round(rng.uniform(0.0, 0.99, size=D), 6)
        T = capi.DeviceTables(N, M, D=D)
        T.tables.fill_(float("nan"))
        T.fill(a, capi.FILL_HB)
        T.status()
        _check_tables(T, a, N, M)
    assert L.stb_fill_fallbacks() == before


@pytest.mark.parametrize("N,M,D", [(30000, 60, 2), (16500, 500, 1), (70000, 130, 1)])
def test_hb_tall_narrow_tables(N, M, D):
    """tables much taller than wide -- one or two strips, thousands of blocks -- and row counts beyond 2^14 and 2^16,
    where the renormalisation period (hence the block) is shorter than 48 rows"""
    a = np.array([0.37, 0.81])[:D]
    T = capi.DeviceTables(N, M, D=D)
    T.tables.fill_(float("nan"))
    L = capi.lib()
    before = L.stb_fill_fallbacks()
    T.fill(a, capi.FILL_HB)
    T.status()
    assert L.stb_fill_fallbacks() == before
    _check_tables(T, a, N, M)


@pytest.mark.parametrize("D,parts", [(70, 0), (3, 5), (9, 2)])
def test_hb_ticket_counters(monkeypatch, D, parts):
    """tiles are handed out by a counter per table and per part of the order list; more tables than counters
    (tables share them), odd numbers of parts, and every table still complete and right"""
    if parts:
        monkeypatch.setenv("STB_HB_TICKET_PARTS", str(parts))
    N, M = 700, 640
    L = capi.lib()
    a = np.resize(synth.discount_grid(64), D)
    T = capi.DeviceTables(N, M, D=D)
    T.tables.fill_(float("nan"))
    before = L.stb_fill_fallbacks()
    T.fill(a, capi.FILL_HB)
    T.status()
    assert L.stb_fill_fallbacks() == before
    for d in sorted(set([0, 1, D // 2, D - 2, D - 1])):
        S1, tab = orc.fill_S(float(a[d]), N, M)
        got = T.packed_host(d)
        assert np.all(np.isfinite(got)), d
        assert orc.close(got, tab, TOL), (d, orc.max_err(got, tab))


def test_hb_short_periods(monkeypatch):
    """a renormalisation period shorter than the block asked for shortens the block"""
    monkeypatch.setenv("STB_FILL_P", "20")
    a = np.array([0.2, 0.8])
    T = capi.DeviceTables(1300, 1100, D=2)
    T.tables.fill_(float("nan"))
    T.fill(a, capi.FILL_HB)
    T.status()
    _check_tables(T, a, 1300, 1100)


def test_hb_more_spine_workgroups_than_compute_units(monkeypatch):
    """one-wave spine workgroups of 32 columns, 40 tables of 3000 columns: thousands of spine workgroups
    for a grid that holds a few hundred at once -- a strip only ever waits for a strip with a smaller ticket"""
    monkeypatch.setenv("STB_HB_C", "1")
    monkeypatch.setenv("STB_HB_P", "1")
    D, N = 40, 3000
    L = capi.lib()
    a = synth.discount_grid(64)[:D]
    T = capi.DeviceTables(N, N, D=D)
    T.tables.fill_(float("nan"))
    before = L.stb_fill_fallbacks()
    T.fill(a, capi.FILL_HB)
    T.status()
    assert L.stb_fill_fallbacks() == before       # (finished, not rescued by the other form after a time-out)
    T2 = capi.DeviceTables(N, N, D=D)
    T2.fill(a, capi.FILL_PC)
    for d in (0, 7, D - 1):
        got = T.packed_host(d)
        assert np.all(np.isfinite(got))
        assert orc.max_err(got, T2.packed_host(d)) <= TOL


def test_hb_gives_up_instead_of_hanging(monkeypatch):
    """every wait of the halo-block form is bounded: with the bound at zero whoever has to wait records an
    error and everybody runs to the end; stb_fill_status then repeats the fill with the producer/consumer
    form -- or reports the failure when that is switched off"""
    L = capi.lib()
    S1, tab = orc.fill_S(0.5, 3000, 3000)
    monkeypatch.setenv("STB_CHAIN_TIMEOUT_MS", "0")
    monkeypatch.setenv("STB_CHAIN_NO_FALLBACK", "1")
    T = capi.DeviceTables(3000, 3000, D=1)
    T.fill([0.5], capi.FILL_HB)
    with pytest.raises(capi.StbError):
        T.status()
    monkeypatch.delenv("STB_CHAIN_NO_FALLBACK")
    before = L.stb_fill_fallbacks()
    T.tables.fill_(float("nan"))
    T.fill([0.5], capi.FILL_HB)
    T.status()
    assert L.stb_fill_fallbacks() == before + 1
    assert orc.max_err(T.packed_host(0), tab) <= TOL
    monkeypatch.delenv("STB_CHAIN_TIMEOUT_MS")
    T.tables.fill_(float("nan"))
    T.fill([0.5], capi.FILL_HB)
    T.status()
    assert L.stb_fill_fallbacks() == before + 1
    assert orc.max_err(T.packed_host(0), tab) <= TOL


@pytest.mark.parametrize("variant", [capi.FILL_SCALED, capi.FILL_HB])
def test_hb_10000_full_table_vs_oracle(variant):
    """configs[1] cell by cell: all 49 985 001 cells of the N=M=10000, a=0.5 table against the oracle's (reference
    recurrence lib/stable.c:380-388), in the form stb_fill_S picks by itself and forced into the halo blocks"""
    N, a = 10000, 0.5
    T = capi.DeviceTables(N, N, D=1)
    T.tables.fill_(float("nan"))
    T.fill([a], variant)
    T.status()
    S1, tab = orc.fill_S(a, N, N)
    got = T.packed_host(0)
    err = np.abs(got - tab) / np.maximum(1.0, np.abs(tab))
    assert np.all(np.isfinite(got))
    assert float(err.max()) <= TOL, float(err.max())


@pytest.mark.parametrize("C", [2, 4])
def test_hb_discount_zero_and_the_samplers_bounds(monkeypatch, C):
    """a = 0 (unsigned Stirling numbers of the first kind: the reference takes it, lib/stable.c:1058-1065,
    lib/sampleb.c:101-118) and the ends of samplea's bracket, A_MIN = 0.01 and A_MAX = 0.98
    (lib/psample.h:89-94), through the halo-block form at N = M = 4000, every cell against the oracle"""
    monkeypatch.setenv("STB_HB_C", str(C))
    L = capi.lib()
    a = np.array([0.0, 0.01, 0.98])
    T = capi.DeviceTables(4000, 4000, D=3)
    T.tables.fill_(float("nan"))
    before = L.stb_fill_fallbacks()
    T.fill(a, capi.FILL_HB)
    T.status()
    assert L.stb_fill_fallbacks() == before
    _check_tables(T, a, 4000, 4000)


def test_float_table_after_a_fill_that_gave_up(monkeypatch):
    """S_FLOAT through S_make when the one-launch fill gives up (bound on its waits set to zero): the
    float slab must be narrowed from the table the producer/consumer form rebuilt, not from what the
    aborted fill left behind (the narrowing is queued only after the status check)"""
    L = capi.lib()
    N, M, a = 1500, 1400, 0.45
    S1, tab = orc.fill_S(a, N, M)
    monkeypatch.setenv("STB_CHAIN_TIMEOUT_MS", "0")
    before = L.stb_fill_fallbacks()
    t = capi.Table(N, M, N, M, a, capi.S_STABLE | capi.S_FLOAT)
    assert L.stb_fill_fallbacks() == before + 1          # the chain form did give up and was replaced
    for n, m in ((3, 2), (100, 57), (700, 699), (1499, 1000), (1500, 1399), (1500, 2)):
        want = tab[orc.row_offset(n, M) + m - 2]
        got = t.S(n, m)
        assert got == float(np.float32(got))             # stored as float
        assert abs(got - want) <= 2e-7 * max(1.0, abs(want)), (n, m, got, want)
    t.free()


@pytest.mark.parametrize("D", [8, 64])
def test_10000_batch_every_table_vs_oracle(D):
    """configs[2]: EVERY table of the 8-per-GPU share (halo-block form, the default there) and of the whole
    64-discount batch on one GPU (producer/consumer form) against the oracle: the last row, an interior row and
    the row where the table turns rectangular-free (n = N/3), per-row maximum relative error"""
    N = 10000
    grid = synth.discount_grid(64)
    a = np.ascontiguousarray(grid[:D])
    T = capi.DeviceTables(N, N, D=D)
    T.fill(a)
    T.status()
    rows = (N, 6311, N // 3)

    def oracle_rows(ad):
        S1, tab = orc.fill_S(float(ad), N, N)
        return [tab[orc.row_offset(n, N):orc.row_offset(n, N) + n - 2].copy() for n in rows], S1[-1]

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        want = list(ex.map(oracle_rows, a))
    for d in range(D):
        for n, w in zip(rows, want[d][0]):
            got = T.row(d, n).cpu().numpy()
            err = np.abs(got - w) / np.maximum(1.0, np.abs(w))
            assert np.all(np.isfinite(got)) and float(err.max()) <= TOL, (d, n, float(err.max()))
        assert abs(float(T.S1[d, N - 1]) - want[d][1]) <= TOL * abs(want[d][1])


@pytest.mark.parametrize("split", [0, 1, 6, 1000])
@pytest.mark.parametrize("C", [2, 4])
def test_quartered_last_tiles(monkeypatch, C, split):
    """the tiles of every strip's last `split` blocks are handed out as four tickets of 12 rows each (STB_HB_SPLIT; the
    default takes 6 where strips have 2 columns per lane): the same cells whatever the number, for log S in double and
    in float and for the V table, several strips and workgroups, tables ending in the middle of a block"""
    monkeypatch.setenv("STB_HB_C", str(C))
    monkeypatch.setenv("STB_HB_SPLIT", str(split))
    L = capi.lib()
    before = L.stb_fill_fallbacks()
    for N, M in ((1531, 1400), (1000, 333)):
        a = np.array([0.11, 0.77])
        T = capi.DeviceTables(N, M, D=2)
        T.tables.fill_(float("nan"))
        T.fill(a, capi.FILL_HB)
        T.status()
        _check_tables(T, a, N, M)
        F = capi.DeviceFloatTables(N, M, D=2)
        F.tables.fill_(float("nan"))
        F.fill(a)
        F.status()
        for d in range(2):
            assert np.array_equal(F.packed_host(d), T.packed_host(d).astype(np.float32))
        V = capi.DeviceVTables(N, M, D=2)
        V.tables.fill_(float("nan"))
        V.fill(a)
        capi.check(L.stb_fill_status())
        for d in range(2):
            got = V.packed_host(d)
            assert np.all(np.isfinite(got)) and orc.max_err(got, orc.fill_V(float(a[d]), N, M)) <= TOL
    assert L.stb_fill_fallbacks() == before


@pytest.mark.parametrize("prep,by_value", [("0", "0"), ("0", "1"), ("1", "0"), ("1", "1")])
def test_prep_launch_and_discounts_by_value(monkeypatch, prep, by_value):
    """k_prep (the workspace zeroed, the S1 vector and the discounts written in one launch) against a memset, k_s1 and a
    host-to-device copy: the same tables and S1 vectors bit for bit; 64 tables still travel by value, 65 by copy"""
    ref = {}
    for mode in (("1", "1"), (prep, by_value)):
        monkeypatch.setenv("STB_HB_PREP", mode[0])
        monkeypatch.setenv("STB_A_BY_VALUE", mode[1])
        outs = []
        for N, M, D in ((1200, 900, 3), (700, 80, 64), (700, 80, 65)):
            a = np.linspace(0.02, 0.97, D)
            T = capi.DeviceTables(N, M, D=D)
            T.tables.fill_(float("nan"))
            T.S1.fill_(float("nan"))
            T.fill(a, capi.FILL_HB)
            T.status()
            outs.append((T.tables[:, :T.elems].cpu().numpy().copy(), T.S1.cpu().numpy().copy()))
        ref[mode] = outs
    for (t0, s0), (t1, s1) in zip(ref[("1", "1")], ref[(prep, by_value)]):
        assert np.all(np.isfinite(s0)) and np.array_equal(s0, s1)
        assert np.array_equal(t0, t1, equal_nan=True)
    S1, tab = orc.fill_S(0.02, 1200, 900)
    assert orc.close(ref[(prep, by_value)][0][1][0], S1, TOL)

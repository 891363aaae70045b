"""GPU parity: the HIP table fill (K1/K2) through the C ABI against the CPU oracle and the golden
fixtures dumped from the reference.  Parity metric |x-y| <= 1e-10*max(1,|y|)."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

import orc
from libstb_amd import capi, synth

pytestmark = pytest.mark.gpu
fh = float.fromhex
TOL = 1e-10


def test_library_sees_gpu():
    L = capi.lib()
    assert L.stb_device_count() >= 1, capi.last_error()


@pytest.mark.parametrize("variant", [capi.FILL_SCALED, capi.FILL_LOGDOMAIN, capi.FILL_PC, capi.FILL_CHAIN, capi.FILL_HB])
def test_small_tables_batched_vs_golden(golden_dir, variant):
    """configs[0] shape (N=200, M=50), five discounts in ONE batched launch sequence."""
    z = np.load(os.path.join(golden_dir, "stable_200x50.npz"))
    keys = ["a0.5", "a0.125", "a0.05", "a0.95", "a2_3"]
    a = np.array([float(z[k + "_a"][0]) for k in keys])
    T = capi.DeviceTables(200, 50, D=len(keys))
    T.fill(a, variant)
    for d, k in enumerate(keys):
        got = T.packed_host(d)
        assert got.shape[0] == 8526
        assert orc.close(got, z[k + "_table"], TOL), (k, orc.max_err(got, z[k + "_table"]))
        assert orc.close(T.S1[d].cpu().numpy(), z[k + "_S1"], TOL)


@pytest.mark.parametrize("variant", [capi.FILL_SCALED, capi.FILL_LOGDOMAIN, capi.FILL_PC, capi.FILL_CHAIN, capi.FILL_HB])
@pytest.mark.parametrize("N,M", [(2, 2), (3, 2), (3, 3), (10, 10), (64, 64), (65, 33), (97, 96), (130, 129), (500, 7),
                                 (1000, 1000), (1500, 260)])
def test_ragged_shapes_vs_oracle(N, M, variant):
    """edge shapes: tiny, one strip / several strips, diagonal inside a strip, M << N"""
    a = np.array([0.31, 0.77])
    T = capi.DeviceTables(N, M, D=2)
    T.fill(a, variant)
    for d in range(2):
        S1, tab = orc.fill_S(a[d], N, M)
        got = T.packed_host(d)
        assert orc.close(got, tab, TOL), (N, M, orc.max_err(got, tab))
        assert orc.close(T.S1[d].cpu().numpy(), S1, TOL)


@pytest.mark.parametrize("C,R", [(1, 16), (1, 60), (2, 32), (2, 7), (4, 32), (4, 128)])
def test_tunings_agree(monkeypatch, C, R):
    """every (columns-per-lane, rows-per-launch) tuning computes the same table"""
    monkeypatch.setenv("STB_FILL_C", str(C))
    monkeypatch.setenv("STB_FILL_R", str(R))
    T = capi.DeviceTables(700, 650, D=1)
    T.fill([0.5])
    S1, tab = orc.fill_S(0.5, 700, 650)
    assert orc.close(T.packed_host(0), tab, TOL)
    assert orc.close(T.S1[0].cpu().numpy(), S1, TOL)


@pytest.mark.parametrize("variant", [capi.FILL_SCALED, capi.FILL_LOGDOMAIN, capi.FILL_PC, capi.FILL_CHAIN, capi.FILL_HB])
@pytest.mark.parametrize("a", [0.5, 0.1, 0.9])
def test_4000_full_table_vs_oracle(golden_dir, a, variant):
    N = 4000
    T = capi.DeviceTables(N, N, D=1)
    T.fill([a], variant)
    S1, tab = orc.fill_S(a, N, N)
    got = T.packed_host(0)
    err = orc.max_err(got, tab)
    assert err <= TOL, err
    assert orc.close(T.S1[0].cpu().numpy(), S1, TOL)
    z = np.load(os.path.join(golden_dir, "stable_big.npz"))
    key = f"N{N}_a{a}"
    o = orc.row_offset(N, N)
    assert orc.close(got[o:o + N - 2], z[key + f"_row{N}"], TOL)


@pytest.mark.parametrize("a", [0.5, 0.07, 0.93])
def test_10000_config2_vs_golden(golden_dir, a):
    """configs[1]: single-discount N=M=10000 -- every row's sum, two full rows, sparse probes."""
    N = 10000
    T = capi.DeviceTables(N, N, D=1)
    T.fill([a])
    z = np.load(os.path.join(golden_dir, "stable_big.npz"))
    key = f"N{N}_a{a}"
    t = T.tables[0].cpu().numpy()
    rowsum = np.zeros(N + 1)
    absmax = np.zeros(N + 1)
    for n in range(3, N + 1):
        o = T.rowoff(n)
        r = t[o:o + n - 2]
        rowsum[n] = np.sum(r)
        absmax[n] = (n - 2) * max(1.0, np.max(np.abs(r)))
    want = z[key + "_rowsum"]
    # (every cell is also compared at 1e-10 in tests/test_gpu_fill_hb.py::test_hb_10000_full_table_vs_oracle;
    # here the bar on a row's sum is 1e-12 of n * max|row|, which a single cell off by 1e-7 relative breaks)
    assert np.all(np.abs(rowsum - want) <= 1e-12 * np.maximum(absmax, 1.0))
    for n in (N // 3, N):
        o = T.rowoff(n)
        assert orc.close(t[o:o + n - 2], z[key + f"_row{n}"], TOL)
    assert orc.close(T.S1[0].cpu().numpy(), z[key + "_S1"], TOL)
    with open(os.path.join(golden_dir, "stable_probes.json")) as f:
        probes = [p for p in json.load(f) if p["N"] == N and fh(p["a"]) == a]
    assert len(probes) > 40
    got = T.lookup([p["n"] for p in probes], [p["m"] for p in probes])
    assert orc.close(got, [fh(p["S"]) for p in probes], TOL)


def _check_against_grid_fixture(T, dl, d, z, probes):
    """table dl of T is member d of the 64-discount grid: row sums, two rows, S1, probes"""
    N = 10000
    t = T.tables[dl].cpu().numpy()
    rowsum = np.zeros(N + 1)
    absmax = np.zeros(N + 1)
    for n in range(3, N + 1):
        o = T.rowoff(n)
        r = t[o:o + n - 2]
        rowsum[n] = np.sum(r)
        absmax[n] = (n - 2) * max(1.0, np.max(np.abs(r)))
    assert np.all(np.isfinite(rowsum))
    assert np.all(np.abs(rowsum - z[f"d{d}_rowsum"]) <= 1e-12 * np.maximum(absmax, 1.0)), d
    for n in (N // 3, N):
        o = T.rowoff(n)
        assert orc.close(t[o:o + n - 2], z[f"d{d}_row{n}"], TOL), (d, n)
    assert orc.close(T.S1[dl].cpu().numpy(), z[f"d{d}_S1"], TOL)
    mine = [p for p in probes if p["d"] == d]
    assert len(mine) > 40
    got = T.lookup([p["n"] for p in mine], [p["m"] for p in mine], d=dl)
    assert orc.close(got, [fh(p["S"]) for p in mine], TOL), d


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_10000_batched_config3_eight_per_gpu(golden_dir, rank):
    """configs[2] as one GPU of eight sees it: its 8 consecutive members of the 64-discount grid at
    N=M=10000 in one batched fill (the form stb_fill_S picks by itself: the halo blocks), status checked,
    no fallback taken, against the reference's tables for the grid members 0, 7, 31, 63."""
    L = capi.lib()
    z = np.load(os.path.join(golden_dir, "stable_grid10k.npz"))
    with open(os.path.join(golden_dir, "stable_grid10k_probes.json")) as f:
        probes = json.load(f)
    grid = synth.discount_grid(64)
    mine = np.ascontiguousarray(grid[8 * rank:8 * rank + 8])
    T = capi.DeviceTables(10000, 10000, D=8)
    T.tables.fill_(float("nan"))
    before = L.stb_fill_fallbacks()
    assert L.stb_fill_tuning(10000, 10000, 8, None, None, None) == 6      # halo-block form (spine + tile workers)
    T.fill(mine)
    T.status()
    assert L.stb_fill_fallbacks() == before
    for d in (0, 7, 31, 63):
        if 8 * rank <= d < 8 * rank + 8:
            assert float(z[f"d{d}_a"][0]) == mine[d - 8 * rank]
            _check_against_grid_fixture(T, d - 8 * rank, d, z, probes)


def test_10000_batched_config3_all_64_on_one_gpu(golden_dir):
    """configs[2] on ONE GPU: all 64 discounts in one batched fill at N=M=10000 (27 GB of tables),
    in the form stb_fill_S picks for that many columns in flight"""
    L = capi.lib()
    z = np.load(os.path.join(golden_dir, "stable_grid10k.npz"))
    with open(os.path.join(golden_dir, "stable_grid10k_probes.json")) as f:
        probes = json.load(f)
    grid = synth.discount_grid(64)
    T = capi.DeviceTables(10000, 10000, D=64)
    before = L.stb_fill_fallbacks()
    T.fill(grid)
    T.status()
    assert L.stb_fill_fallbacks() == before
    for d in (0, 7, 31, 63):
        _check_against_grid_fixture(T, d, d, z, probes)
    # the other 60: finite everywhere and monotone in the discount at a probe cell
    probe = T.tables[:, T.rowoff(10000) + 4998].cpu().numpy()       # log S^10000_5000 per discount
    assert np.all(np.isfinite(probe)) and np.all(np.diff(probe) < 0)


def test_10000_round_trip_property():
    """size-independent property at full size: the last row satisfies the recurrence
    S^n_m = (n-1-m a) S^{n-1}_m + S^{n-1}_{m-1} in the log domain, row by row sampled."""
    N, a = 10000, 0.5
    T = capi.DeviceTables(N, N, D=1)
    T.fill([a])
    t = T.tables[0].cpu().numpy()
    for n in (4, 57, 1000, 5001, 10000):
        cur = t[T.rowoff(n):T.rowoff(n) + n - 2]          # m = 2..n-1
        prv = t[T.rowoff(n - 1):T.rowoff(n - 1) + n - 3]  # m = 2..n-2
        m = np.arange(3, n - 1)                            # interior columns
        up = prv[m - 2]
        left = prv[m - 3]
        want = np.logaddexp(np.log(n - 1 - m * a) + up, left)
        assert orc.close(cur[m - 2], want, 1e-12)


def test_v_table_bit_exact(golden_dir):
    """V ratios are plain mul/div in the reference's order: expect equality to the last bit."""
    z = np.load(os.path.join(golden_dir, "uv_200x50.npz"))
    T = capi.DeviceVTables(200, 50, D=3)
    T.fill([0.5, 0.05, 0.95])
    for d, key in enumerate(("a0.5", "a0.05", "a0.95")):
        got = T.packed_host(d)
        assert np.array_equal(got, z[key + "_V"]), orc.max_err(got, z[key + "_V"])


def test_v_table_big_vs_oracle():
    """3000 x 1200: on the reference's own recurrence (two dependent divisions a row) the table is the reference's
    bit for bit; the default for a table this size takes every V^n_m = S^n_m / S^n_{m-1} from the S recurrence's
    block-floating cells instead -- one division per cell, off the serial path -- and is within 1e-10"""
    want = orc.fill_V(0.4, 3000, 1200)
    T = capi.DeviceVTables(3000, 1200, D=1)
    T.tables.fill_(float("nan"))
    T.fill([0.4], exact=True)
    assert np.array_equal(T.packed_host(0), want)
    T.tables.fill_(float("nan"))
    assert capi.lib().stb_fill_takes_kind(3000, 1200, 1, 2) == 1
    T.fill([0.4])
    T.status = capi.DeviceTables.status.__get__(T)
    T.status()
    got = T.packed_host(0)
    assert np.all(np.isfinite(got))
    assert orc.max_err(got, want) <= TOL, orc.max_err(got, want)


@pytest.mark.parametrize("N,M,D,a", [(10000, 10000, 1, [0.5]), (4000, 4000, 3, [0.0, 0.01, 0.98]), (2500, 700, 2, [0.3, 0.9]),
                                     (777, 777, 1, [0.66])])
def test_v_table_from_the_s_recurrence_vs_oracle(N, M, D, a):
    """the V table (reference lib/stable.c:451-482) taken from the halo-block fill's cells, every cell against the
    oracle's V recurrence at 1e-10: config sizes, the ends of samplea's bracket and a = 0, a table wider than tall is
    not possible (M <= N) but a narrow one is, and the diagonal column V^n_n = 1 / S^n_{n-1} is part of every row"""
    L = capi.lib()
    assert L.stb_fill_takes_kind(N, M, D, 2) == 1
    T = capi.DeviceVTables(N, M, D=D)
    T.tables.fill_(float("nan"))
    before = L.stb_fill_fallbacks()
    T.fill(a)
    capi.check(L.stb_fill_status())
    assert L.stb_fill_fallbacks() == before
    for d in range(D):
        want = orc.fill_V(float(a[d]), N, M)
        got = T.packed_host(d)
        assert np.all(np.isfinite(got)), d
        assert orc.max_err(got, want) <= TOL, (d, orc.max_err(got, want))


@pytest.mark.parametrize("N,M,D,a", [(10000, 10000, 1, [0.5]), (3000, 1200, 2, [0.05, 0.95]), (4000, 4000, 8, None)])
def test_float_tables_written_once(N, M, D, a):
    """S_FLOAT (reference lib/stable.c:389-449, :483-537): the recurrence in double, the stored value a float -- here
    narrowed by the kernel that computed it, no double slab anywhere.  Every cell equals the double table's value
    rounded to float (S) / is within a float ulp of the oracle's V."""
    L = capi.lib()
    a = np.asarray(a if a is not None else synth.discount_grid(64)[:D], dtype=np.float64)
    assert L.stb_fill_takes_kind(N, M, D, 1) == 1 and L.stb_fill_takes_kind(N, M, D, 3) == 1
    F = capi.DeviceFloatTables(N, M, D=D)
    F.tables.fill_(float("nan"))
    F.fill(a)
    F.status()
    T = capi.DeviceTables(N, M, D=D)
    T.fill(a, capi.FILL_HB)
    T.status()
    for d in range(D):
        got, dbl = F.packed_host(d), T.packed_host(d)
        assert np.array_equal(got, dbl.astype(np.float32)), d      # the same double, narrowed once
        assert np.array_equal(F.S1[d].cpu().numpy(), T.S1[d].cpu().numpy())
    V = capi.DeviceVTables(N, M, D=D, dtype="f32")
    V.tables.fill_(float("nan"))
    V.fill(a)
    capi.check(L.stb_fill_status())
    for d in (0, D - 1):
        want = orc.fill_V(float(a[d]), N, M)
        got = V.packed_host(d)
        assert np.all(np.isfinite(got))
        assert np.max(np.abs(got.astype(np.float64) - want) / np.abs(want)) <= 2.0 ** -23, d


@pytest.mark.parametrize("variant", [capi.FILL_PC, capi.FILL_CHAIN, capi.FILL_HB, capi.FILL_SCALED])
@pytest.mark.parametrize("a", [0.0, 0.01, 0.07, 0.5, 0.98])
def test_growth_next_to_the_diagonal(a, variant, monkeypatch):
    """cells next to the diagonal grow by ~n^2/2 per row (S^n_{n-1} = n(n-1)(1-a)/2): the
    renormalisation period must be sized for that, at every discount, with the longest launches"""
    monkeypatch.setenv("STB_FILL_R", "120")
    monkeypatch.setenv("STB_FILL_C", "2")
    N = 6000
    T = capi.DeviceTables(N, N, D=1)
    T.fill([a], variant)
    S1, tab = orc.fill_S(a, N, N)
    got = T.packed_host(0)
    assert np.all(np.isfinite(got))
    assert orc.max_err(got, tab) <= TOL


@pytest.mark.parametrize("C,P,MG,NF,RD", [(2, 1, 3, 1, 4), (4, 1, 3, 1, 4), (2, 1, 3, 2, 4), (2, 1, 3, 1, 8), (4, 1, 2, 1, 4),
                                          (4, 1, 1, 1, 4), (1, 1, 3, 1, 4), (1, 2, 3, 2, 8), (1, 4, 2, 2, 8), (2, 2, 2, 1, 8)])
def test_chain_geometries_agree(monkeypatch, C, P, MG, NF, RD):
    """chain form: every strip shape (columns per producer lane, producer waves, consumer groups,
    fetcher waves, ring depth) computes the same tables and no strip gives up waiting for its
    neighbour (several strips, several tables)"""
    for k, v in (("C", C), ("P", P), ("MG", MG), ("NF", NF), ("RD", RD)):
        monkeypatch.setenv("STB_CHAIN_" + k, str(v))
    L = capi.lib()
    a = np.array([0.05, 0.5, 0.93])
    T = capi.DeviceTables(900, 700, D=3)
    T.tables.fill_(float("nan"))
    before = L.stb_fill_fallbacks()
    T.fill(a, capi.FILL_CHAIN)
    T.status()
    assert L.stb_fill_fallbacks() == before
    for d in range(3):
        S1, tab = orc.fill_S(a[d], 900, 700)
        got = T.packed_host(d)
        assert np.all(np.isfinite(got))
        assert orc.close(got, tab, TOL), (d, orc.max_err(got, tab))
        assert orc.close(T.S1[d].cpu().numpy(), S1, TOL)


def test_chain_more_blocks_than_fit_at_once():
    """chain form with more column blocks than the chip holds at once: blocks take tickets in
    column-major order, so late blocks only ever wait for blocks that are running or done"""
    D = 40
    a = synth.discount_grid(64)[:D]
    N = 3000
    T = capi.DeviceTables(N, N, D=D)
    T.fill(a, capi.FILL_CHAIN)
    T.status()
    T2 = capi.DeviceTables(N, N, D=D)
    T2.fill(a, capi.FILL_PC)
    for d in (0, 7, D - 1):
        assert orc.max_err(T.packed_host(d), T2.packed_host(d)) <= TOL


def test_chain_random_shapes_vs_oracle():
    """chain form (the default) on random shapes, batches and discounts"""
    rng = np.random.default_rng(20261003)
    for _ in range(14):
        N = int(rng.integers(3, 2600))
        M = int(rng.integers(2, N + 1))
        D = int(rng.integers(1, 5))
        a = np.round(rng.uniform(0.0, 0.99, size=D), 6)
        T = capi.DeviceTables(N, M, D=D)
        T.tables.fill_(float("nan"))
        T.fill(a, capi.FILL_CHAIN)
        T.status()
        for d in range(D):
            S1, tab = orc.fill_S(float(a[d]), N, M)
            got = T.packed_host(d)
            assert np.all(np.isfinite(got)), (N, M, D, a)
            assert orc.close(got, tab, TOL), (N, M, D, a, orc.max_err(got, tab))
            assert orc.close(T.S1[d].cpu().numpy(), S1, TOL)


def test_chain_gives_up_instead_of_hanging(monkeypatch):
    """every wait of the chain form is bounded: with the bound set to zero a block that has to wait
    records an error and runs to its end; the fill returns.  stb_fill_status then repeats the fill
    with the producer/consumer form (no waits between workgroups) -- or, with the repeat switched
    off, reports the failure."""
    L = capi.lib()
    S1, tab = orc.fill_S(0.5, 3000, 3000)
    monkeypatch.setenv("STB_CHAIN_TIMEOUT_MS", "0")
    monkeypatch.setenv("STB_CHAIN_NO_FALLBACK", "1")
    T = capi.DeviceTables(3000, 3000, D=1)
    T.fill([0.5], capi.FILL_CHAIN)
    with pytest.raises(capi.StbError):
        T.status()
    monkeypatch.delenv("STB_CHAIN_NO_FALLBACK")
    before = L.stb_fill_fallbacks()
    T.tables.fill_(float("nan"))
    T.fill([0.5], capi.FILL_CHAIN)
    T.status()                          # gave up -> refilled by k_fill_pc
    assert L.stb_fill_fallbacks() == before + 1
    assert orc.max_err(T.packed_host(0), tab) <= TOL
    monkeypatch.delenv("STB_CHAIN_TIMEOUT_MS")
    T.tables.fill_(float("nan"))
    T.fill([0.5], capi.FILL_CHAIN)      # and the next fill is fine again
    T.status()
    assert L.stb_fill_fallbacks() == before + 1
    assert orc.max_err(T.packed_host(0), tab) <= TOL


def test_status_after_a_non_chain_fill_is_clean():
    """stb_fill_status refers to the LAST fill of the thread: a chain fill that gave up, followed by
    a fill of another form in the same workspace, must not report (or read) the old header"""
    T = capi.DeviceTables(3000, 3000, D=1)
    T.fill([0.5], capi.FILL_CHAIN)
    T.fill([0.5], capi.FILL_PC)
    T.status()
    T.fill([0.5], capi.FILL_LOGDOMAIN)
    T.status()


def test_set_device_is_recorded_by_tables():
    """stb_set_device / stb_get_device: objects are created on the chosen GPU and keep using it"""
    L = capi.lib()
    n = L.stb_device_count()
    assert L.stb_set_device(n) != 0 and b"stb_set_device" in L.stb_last_error()
    assert L.stb_set_device(0) == 0
    assert L.stb_get_device() == 0
    t = capi.Table(300, 50, 300, 50, 0.5, capi.S_STABLE)
    assert abs(t.S(300, 25) - orc.fill_S(0.5, 300, 50)[1][orc.row_offset(300, 50) + 23]) < 1e-9
    prev = L.stb_device_enter(0)
    L.stb_device_leave(prev)
    t.free()


def test_pc_sub_batches_on_forked_streams(monkeypatch):
    """from 24 tables on k_fill_pc runs two halves of the batch on two internal streams forked from
    and joined to the caller's: same tables as one stream gives, for a batch that does not split
    evenly, and work queued behind the fill on the caller's stream sees the finished tables"""
    N, M, D = 700, 500, 27
    a = synth.discount_grid(64)[:D]
    monkeypatch.setenv("STB_PC_STREAMS", "1")
    T1 = capi.DeviceTables(N, M, D=D)
    T1.fill(a, capi.FILL_PC)
    T1.status()
    for ns in ("2", "3"):
        monkeypatch.setenv("STB_PC_STREAMS", ns)
        T2 = capi.DeviceTables(N, M, D=D)
        T2.tables.fill_(float("nan"))
        T2.fill(a, capi.FILL_PC)
        after = T2.tables.clone()        # queued on the same stream behind the fill, nothing synchronised in between
        T2.status()
        for d in (0, D // 2 - 1, D // 2, D - 1):
            for n in (3, 129, N):
                o = T2.rowoff(n)
                ln = T2.row(d, n).numel()
                assert torch.equal(T2.row(d, n), T1.row(d, n)), (ns, d, n)
                assert torch.equal(after[d, o:o + ln], T1.row(d, n)), (ns, d, n)
    monkeypatch.delenv("STB_PC_STREAMS")
    tab = orc.fill_S(float(a[D - 1]), N, M)[1]
    assert orc.max_err(T1.packed_host(D - 1), tab) <= TOL


def test_profile_span_of_overlapping_launches():
    """stb_fill_profile_end sums the launches' device times, stb_fill_profile_span is first start to
    last end: equal-ish for one stream, the span clearly shorter when two sub-batches overlap"""
    L = capi.lib()
    T = capi.DeviceTables(4000, 4000, D=32)
    a = synth.discount_grid(64)[:32]
    T.fill(a, capi.FILL_PC)
    torch.cuda.synchronize()
    L.stb_fill_profile_begin()
    T.fill(a, capi.FILL_PC)
    torch.cuda.synchronize()
    ms, n = C.c_double(0.0), C.c_int(0)
    capi.check(L.stb_fill_profile_end(C.byref(ms), C.byref(n)))
    span = L.stb_fill_profile_span()
    T.status()
    assert n.value > 2 and 0.0 < span <= ms.value * 1.001
    assert span < 0.8 * ms.value        # two streams ran side by side

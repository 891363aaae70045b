"""Past the configs (round 6).  `stcnt_int` is uint16 (lib/psample.h:68): t -- and with it the table's M -- goes up to
65 535, N as far as memory; the configs stop at 10^4.  Here: N = M = 45 000 (8 GB; with 2-column strips 563 strips, so the
kernels' record offsets no longer fit their LDS copy), N = 66 000 with M = 65 535 (17 GB, the last column a uint16 can
name), and a fused evaluation whose pairs sit in those last columns.  The checker streams the wanted rows (the oracle's
loop over two row buffers, tests/orc.py rows_stream): bars as everywhere, |x - y| <= 1e-10 max(1, |y|)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import orc
from libstb_amd import capi, synth

pytestmark = pytest.mark.gpu
TOL = 1e-10
THREADS = max(1, min(16, len(os.sched_getaffinity(0))))


def check_rows(T, d, want):
    worst = 0.0
    for n, w in want.items():
        o = T.rowoff(n)
        got = T.tables[d, o:o + len(w)].cpu().numpy()
        assert np.all(np.isfinite(got)), n
        worst = max(worst, float(np.max(np.abs(got - w) / np.maximum(1.0, np.abs(w)))))
    return worst


@pytest.mark.parametrize("env", [{}, {"STB_HB_C": "2"}], ids=["default", "two_column_strips"])
def test_a_table_of_45000_rows_and_columns(monkeypatch, env):
    """STB_HB_C=2: 563 strips -- the strips' record offsets are read from global memory (JW + 2 > 512), the tile order
    too; default: what the dispatch takes by itself.  Rows N, N/2, N/3 and two near the top against the oracle."""
    L = capi.lib()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N = M = 45000
    want = orc.rows_stream(0.5, N, M, [3, 700, N // 3, N // 2, N], threads=THREADS)
    T = capi.DeviceTables(N, M, D=1)
    fb = L.stb_fill_fallbacks()
    T.fill([0.5], capi.FILL_HB)
    torch.cuda.synchronize()
    T.status()
    assert L.stb_fill_fallbacks() == fb
    assert check_rows(T, 0, want) <= TOL
    if not env:                                  # ... and whatever form the library picks by itself
        T.tables.zero_()
        T.fill([0.5])
        T.status()
        assert check_rows(T, 0, want) <= TOL
    del T
    torch.cuda.empty_cache()


def test_the_last_column_a_uint16_can_name():
    """N = 66 000, M = 65 535: the default dispatch and the halo-block form; the last rows, with m = 65 535 among them"""
    L = capi.lib()
    N, M = 66000, 65535
    rows = [65536, 65537, N - 1, N]
    want = orc.rows_stream(0.37, N, M, rows, threads=THREADS)
    assert len(want[N]) == M - 1 and len(want[65536]) == M - 1
    T = capi.DeviceTables(N, M, D=1)
    for variant in (capi.FILL_SCALED, capi.FILL_HB):
        T.tables.zero_()
        fb = L.stb_fill_fallbacks()
        T.fill([0.37], variant)
        torch.cuda.synchronize()
        T.status()
        assert L.stb_fill_fallbacks() == fb
        assert check_rows(T, 0, want) <= TOL, variant
    # look-ups with S_S semantics in the last columns
    n = torch.tensor([N, N, N - 1, 65536, 65536], dtype=torch.int32, device="cuda")
    m = torch.tensor([65535, 65534, 65535, 65535, 2], dtype=torch.int32, device="cuda")
    out = torch.empty(5, dtype=torch.float64, device="cuda")
    capi.check(L.stb_lookup_S(T.tables.data_ptr(), T.S1.data_ptr(), N, M, n.data_ptr(), m.data_ptr(), 5, out.data_ptr(), None))
    got = out.cpu().numpy()
    w = np.array([want[N][65533], want[N][65532], want[N - 1][65533], want[65536][65533], want[65536][0]])
    assert np.all(np.abs(got - w) <= TOL * np.maximum(1.0, np.abs(w)))
    del T
    torch.cuda.empty_cache()


def test_a_fused_evaluation_with_t_near_65535():
    """pairs in the table's last columns (t up to 65 535) and last rows: the sum of S_S over them -- the evaluation of the
    set minus that of the same restaurants without pairs -- against the oracle's streamed rows, fused and through stored
    tables"""
    L = capi.lib()
    N, M = 66000, 65535
    a = 0.43
    rows = [65537, 65800, N]
    want = orc.rows_stream(a, N, M, rows, threads=THREADS)
    rng = np.random.default_rng(7)
    G = 3000
    n = rng.choice(rows, G).astype(np.uint32)
    t = np.where(rng.random(G) < 0.7, rng.integers(65000, 65536, G), rng.integers(2, 3000, G)).astype(np.uint16)
    t[:4] = 65535
    n[:2] = N
    I = 3
    K = np.array([1000, 1500, 500], dtype=np.int32)
    Tt = np.array([int(t[:1000].sum()), int(t[1000:2500].sum()), int(t[2500:].sum())], dtype=np.uint32)
    bpar = np.array([3.0, 10.0, 40.0])
    pair_sum = float(np.sum([want[int(nn)][int(tt) - 2] for nn, tt in zip(n, t)]))
    x = np.array([a, 0.2])
    h = L.stb_groups_create(I, orc.i32p(K), orc.u32p(Tt), orc.u32p(n), orc.u16p(t), orc.dp(bpar), N, M, 2)
    assert h, capi.last_error()
    none = np.ones(G, dtype=np.uint32)                       # n = 1: a pair that contributes nothing (lib/samplea.c:78)
    h0 = L.stb_groups_create(I, orc.i32p(K), orc.u32p(Tt), orc.u32p(none), orc.u16p(np.ones(G, dtype=np.uint16)), orc.dp(bpar), N, M, 2)
    assert h0, capi.last_error()
    try:
        fb = L.stb_groups_fallbacks()
        fused, base, tables = np.zeros(2), np.zeros(2), np.zeros(2)
        capi.check(L.stb_groups_aterms(h, capi.dp(x), 2, capi.dp(fused)))
        capi.check(L.stb_groups_aterms(h0, capi.dp(x), 2, capi.dp(base)))
        capi.check(L.stb_groups_aterms_tables(h, capi.dp(x), 2, capi.dp(tables)))
        assert L.stb_groups_fallbacks() == fb
        assert abs((fused[0] - base[0]) - pair_sum) <= TOL * abs(pair_sum)
        assert np.all(np.abs(fused - tables) <= TOL * np.abs(tables))
    finally:
        L.stb_groups_free(h)
        L.stb_groups_free(h0)

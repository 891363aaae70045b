"""A caller whose (n,t) pairs change between calls -- every real one: the reference's Gibbs loop rewrites t[j][i] and
T[j] in each iteration (test/demo.c:405-445) and then calls samplea (:478-480).  The pairs' way into an existing device
set (stb_groups_pairs_begin / _put / _commit, stb_groups_update_pairs) and the cell lists built from the count slab
(csrc/lists.hip) against a set made from scratch and against the sort-based lists of rounds 2-4: identical bits."""
import ctypes as C

import numpy as np
import pytest

import orc
from libstb_amd import capi, synth

pytestmark = pytest.mark.gpu


def create(L, g, n, t, N, M, D):
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), None if n is None else orc.u32p(n), None if t is None else orc.u16p(t),
                            orc.dp(g.bpar), N, M, D)
    assert h, capi.last_error()
    return h


def aterms(L, h, x):
    out = np.zeros(len(x))
    capi.check(L.stb_groups_aterms(h, capi.dp(np.ascontiguousarray(x)), len(x), capi.dp(out)))
    return out


def edgy(seed, I=60, K=50, n_max=1500):
    """synthetic pairs with every kind of edge among them"""
    g = synth.groups(I, K, n_max, "wide", seed=seed)
    n, t = g.n.copy(), g.t.copy()
    n[0], t[0] = 1, 1              # skipped (lib/samplea.c:78)
    n[1], t[1] = 77, 77            # t = n: log 1
    n[2], t[2] = 500, 1            # column 1
    n[3], t[3] = 3, 2              # the table's first cell
    n[4], t[4] = n_max, 2
    n[5], t[5] = n_max, n_max - 1  # next to the diagonal in the last row
    n[6:40], t[6:40] = 400, 123    # many pairs on one cell
    n[40], t[40] = 2, 1            # S^2_1 = 1 - a
    n[41:50], t[41:50] = n_max, 1
    return g, n, t


FORMS = [("hb2", {"STB_HB_DOT_C": "2"}, (1, 3)), ("hb3", {"STB_HB_DOT_C": "3"}, (1, 4, 8)), ("hb4", {"STB_HB_DOT_C": "4"}, (1, 3, 8)),
         ("grid2", {"STB_ATERMS_GRID": "1", "STB_GRID_C": "2"}, (2, 5)), ("grid4", {"STB_ATERMS_GRID": "1", "STB_GRID_C": "4"}, (3, 8)),
         ("grid4jobs", {"STB_ATERMS_GRID": "1", "STB_GRID_C": "4", "STB_GRID_HELP_NW": "1"}, (8,)),
         ("grid8", {"STB_ATERMS_GRID": "1", "STB_GRID_C": "8"}, (2, 8)),
         ("grid8jobs", {"STB_ATERMS_GRID": "1", "STB_GRID_C": "8", "STB_GRID_HELP_NW": "1"}, (8,))]


@pytest.mark.parametrize("name,env,Ds", FORMS, ids=[f[0] for f in FORMS])
def test_lists_from_the_count_slab_equal_the_sorted_lists(monkeypatch, name, env, Ds):
    """the CSR lists k_count_cells / k_emit_cells build are the lists the radix sort + run-length encoding built: the
    fused sums (a fixed order over the list entries) have the same bits; the helper-job choice of the grid form too"""
    L = capi.lib()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g, n, t = edgy(11)
    N = M = 1500
    grid = synth.discount_grid(64)[::7]
    outs = {}
    fb = L.stb_groups_fallbacks()
    for slab in ("1", "0"):
        monkeypatch.setenv("STB_LISTS_SLAB", slab)
        h = create(L, g, n, t, N, M, max(Ds))
        try:
            outs[slab] = [aterms(L, h, grid[:D]) for D in Ds]
        finally:
            L.stb_groups_free(h)
    assert L.stb_groups_fallbacks() == fb            # (the fused kernels ran: nothing was redone through stored tables)
    for a, b in zip(outs["1"], outs["0"]):
        assert np.all(np.isfinite(a)) and np.array_equal(a, b), (a, b)
    # ... and against stored tables + gather
    monkeypatch.setenv("STB_LISTS_SLAB", "1")
    h = create(L, g, n, t, N, M, max(Ds))
    try:
        want = np.zeros(max(Ds))
        capi.check(L.stb_groups_aterms_tables(h, capi.dp(np.ascontiguousarray(grid[:max(Ds)])), max(Ds), capi.dp(want)))
        assert orc.close(outs["1"][-1], want, 1e-12)
    finally:
        L.stb_groups_free(h)


@pytest.mark.parametrize("name,env,Ds", FORMS, ids=[f[0] for f in FORMS])
def test_new_pairs_in_a_kept_set_equal_a_new_set(monkeypatch, name, env, Ds):
    """stb_groups_update_pairs: the set's buffers, slab and stream stay, the pairs change -- every evaluation has the bits
    of a set created from those pairs; back to the first pairs gives the first values"""
    L = capi.lib()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N = M = 1500
    g, n1, t1 = edgy(11)
    _, n2, t2 = edgy(12)
    n3, t3 = n1.copy(), t1.copy()
    n3[100] += 1                                    # one customer joins one table
    D = max(Ds)
    x = synth.discount_grid(64)[::7][:D]
    fresh = []
    for n, t in ((n1, t1), (n2, t2), (n3, t3)):
        h = create(L, g, n, t, N, M, D)
        fresh.append([aterms(L, h, x[:d]) for d in Ds])
        L.stb_groups_free(h)
    assert not np.array_equal(fresh[0][-1], fresh[2][-1])
    h = create(L, g, n1, t1, N, M, D)
    try:
        for k, (n, t) in ((0, (n1, t1)), (1, (n2, t2)), (2, (n3, t3)), (0, (n1, t1)), (2, (n3, t3))):
            capi.check(L.stb_groups_update_pairs(h, orc.u32p(n), orc.u16p(t)))
            for d, want in zip(Ds, fresh[k]):
                assert np.array_equal(aterms(L, h, x[:d]), want), (k, d)
    finally:
        L.stb_groups_free(h)


def test_pairs_in_pieces_new_bounds_and_an_empty_set(monkeypatch):
    """a set created EMPTY (no pairs, no bounds), filled restaurant by restaurant; new bounds at commit re-size what depends
    on them; the maxima fall out of the puts; errors are errors"""
    L = capi.lib()
    g, n, t = edgy(5, I=40, K=30, n_max=800)
    x = np.array([0.2, 0.5, 0.7])
    h0 = create(L, g, n, t, 800, 800, 3)
    want800 = aterms(L, h0, x)
    L.stb_groups_free(h0)
    h0 = create(L, g, n, t, 1024, 896, 3)
    want1024 = aterms(L, h0, x)
    L.stb_groups_free(h0)
    assert orc.close(want800, want1024, 1e-12)       # (the cells do not depend on the bounds; their rounding may)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), None, None, None, None, 0, 0, 3)
    assert h, capi.last_error()
    try:
        out = np.zeros(3)
        assert L.stb_groups_aterms(h, capi.dp(x), 3, capi.dp(out)) != 0            # no pairs yet
        assert L.stb_groups_pairs_put(h, orc.u32p(n), orc.u16p(t), 10, None, None) != 0   # begin comes first
        mn, mt = C.c_uint(), C.c_uint()
        for N, M, want in ((800, 800, want800), (1024, 896, want1024), (800, 800, want800)):
            capi.check(L.stb_groups_pairs_begin(h))
            off = 0
            for i in range(g.I):
                k = int(g.K[i])
                capi.check(L.stb_groups_pairs_put(h, orc.u32p(n[off:off + k].copy()), orc.u16p(t[off:off + k].copy()), k, C.byref(mn), C.byref(mt)))
                off += k
            assert (mn.value, mt.value) == (int(n.max()), int(t.max()))
            assert L.stb_groups_aterms(h, capi.dp(x), 3, capi.dp(out)) != 0        # not committed
            capi.check(L.stb_groups_pairs_commit(h, orc.u32p(g.T), orc.dp(g.bpar), N, M))
            assert np.array_equal(aterms(L, h, x), want)
        # too many pairs, too few
        capi.check(L.stb_groups_pairs_begin(h))
        assert L.stb_groups_pairs_put(h, orc.u32p(n), orc.u16p(t), len(n) + 1, None, None) != 0
        capi.check(L.stb_groups_pairs_put(h, orc.u32p(n), orc.u16p(t), len(n) - 1, None, None))
        assert L.stb_groups_pairs_commit(h, None, None, 0, 0) != 0
        # (the failed commit ends the hand-over: begin again)
        capi.check(L.stb_groups_update_pairs(h, orc.u32p(n), orc.u16p(t)))
        assert np.array_equal(aterms(L, h, x), want800)
    finally:
        L.stb_groups_free(h)


def test_new_pairs_with_log_zero_and_the_order_of_the_pairs(monkeypatch):
    """a pair outside the support makes the sum -inf (lib/stable.c:948-949) -- counted on the device now -- and the next
    set of pairs without one is finite again; the caller's order of the pairs does not matter (integer counts)"""
    L = capi.lib()
    g, n, t = edgy(3, I=30, K=40, n_max=700)
    x = np.array([0.3, 0.6])
    h = create(L, g, n, t, 700, 700, 2)
    try:
        base = aterms(L, h, x)
        assert np.all(np.isfinite(base))
        nb, tb = n.copy(), t.copy()
        nb[7], tb[7] = 5, 9
        capi.check(L.stb_groups_update_pairs(h, orc.u32p(nb), orc.u16p(tb)))
        assert np.all(np.isneginf(aterms(L, h, x)))
        perm = np.random.default_rng(1).permutation(len(n))
        capi.check(L.stb_groups_update_pairs(h, orc.u32p(n[perm].copy()), orc.u16p(t[perm].copy())))
        assert np.array_equal(aterms(L, h, x), base)
        one = aterms(L, h, x[:1])                      # one discount on a fresh set: fused too (lists are cheap now)
        assert one[0] == base[0]
    finally:
        L.stb_groups_free(h)


def ragged(g, n, t):
    NP = C.POINTER(C.c_uint32) * g.I
    TP = C.POINTER(C.c_uint16) * g.I
    nn, tt = NP(), TP()
    off = 0
    for i in range(g.I):
        nn[i] = C.cast(n.ctypes.data + 4 * off, C.POINTER(C.c_uint32))
        tt[i] = C.cast(t.ctypes.data + 2 * off, C.POINTER(C.c_uint16))
        off += int(g.K[i])
    return nn, tt


def test_samplea_on_changing_pairs_equals_samplea_from_scratch(monkeypatch):
    """samplea call after call on pairs that change (one count, then the largest count -- the table bounds move --, then
    everything): every draw, abscissa and evaluation equals the one of a thread that has never sampled before, and the one
    with STB_SAMPLEA_CACHE=1"""
    L = capi.lib()
    g = synth.groups(100, 100, 1000, "wide")
    sets = [(g.n.copy(), g.t.copy())]
    n2, t2 = g.n.copy(), g.t.copy()
    n2[int(np.argmin(n2))] += 1
    sets.append((n2, t2))
    n3, t3 = g.n.copy(), g.t.copy()
    n3[17] = 1203                                   # the largest n moves past a multiple of 128
    sets.append((n3, t3))
    g4 = synth.groups(100, 100, 1000, "wide", seed=9)
    sets.append((g4.n.copy(), g4.t.copy()))
    sets.append(sets[0])

    def draw(n, t):
        nn, tt = ragged(g, n, t)
        orc.seed_libc(777, 12345)
        a = L.samplea(0.5, g.I, orc.i32p(g.K), orc.u32p(g.T), nn, tt, None, orc.dp(g.bpar), None, 1, 0)
        xs = []
        for i in range(L.stb_sampler_trace_count()):
            x, y = C.c_double(), C.c_double()
            L.stb_sampler_trace_get(i, C.byref(x), C.byref(y))
            xs.append((x.value, y.value))
        return a, xs

    scratch = []
    for n, t in sets:
        L.stb_sampler_cache_clear()
        scratch.append(draw(n, t))
    L.stb_sampler_cache_clear()
    fb = L.stb_groups_fallbacks()
    assert [draw(n, t) for n, t in sets] == scratch
    monkeypatch.setenv("STB_SAMPLEA_CACHE", "1")
    assert [draw(n, t) for n, t in sets] == scratch
    assert [draw(*sets[0]), draw(*sets[0])] == [scratch[0], scratch[0]]     # (the second of these is served by the kept pairs)
    assert L.stb_groups_fallbacks() == fb
    L.stb_sampler_cache_clear()


def test_a_fused_evaluation_that_gives_up_is_counted_and_redone(monkeypatch):
    """STB_CHAIN_TIMEOUT_MS=0 makes every wait of the grid form give up at once: the evaluation is redone through stored
    tables (the same values as stb_groups_aterms_tables) and stb_groups_fallbacks counts it -- for the blocking call and
    for stb_groups_aterms_device + stb_groups_wait, whose device buffer is rewritten"""
    import torch
    L = capi.lib()
    monkeypatch.setenv("STB_ATERMS_GRID", "1")
    g, n, t = edgy(11)
    x = np.ascontiguousarray(synth.discount_grid(64)[::13])
    D = len(x)
    h = create(L, g, n, t, 1500, 1500, D)
    try:
        good = aterms(L, h, x)
        want = np.zeros(D)
        capi.check(L.stb_groups_aterms_tables(h, capi.dp(x), D, capi.dp(want)))
        fb = L.stb_groups_fallbacks()
        assert orc.close(good, want, 1e-12) and fb == L.stb_groups_fallbacks()
        monkeypatch.setenv("STB_CHAIN_TIMEOUT_MS", "0")
        redo = aterms(L, h, x)
        assert L.stb_groups_fallbacks() == fb + 1
        assert np.array_equal(redo, want)
        dev = torch.full((D,), -1.0, dtype=torch.float64, device="cuda")
        capi.check(L.stb_groups_aterms_device(h, capi.dp(x), D, C.c_void_p(dev.data_ptr()), None))
        capi.check(L.stb_groups_wait(h))
        torch.cuda.synchronize()
        assert L.stb_groups_fallbacks() == fb + 2
        assert np.array_equal(dev.cpu().numpy(), want)
        monkeypatch.delenv("STB_CHAIN_TIMEOUT_MS")
        assert np.array_equal(aterms(L, h, x), good)
        assert L.stb_groups_fallbacks() == fb + 2
    finally:
        L.stb_groups_free(h)


def test_a_cell_with_a_huge_count_and_an_empty_set(monkeypatch):
    """600 000 pairs on ONE cell (its count does not fit a dense word: the tile goes through the CSR lists) beside ordinary
    pairs, in the grid form and in the halo-block form, lists from the count slab: equal to stored tables + gather; then a
    set without any pair (every K = 0): the restaurant terms alone, as the reference's aterms gives"""
    L = capi.lib()
    I, K = 700, 1000
    g = synth.groups(I, K, 1200, "wide", seed=21)
    n, t = g.n.copy(), g.t.copy()
    n[:600000], t[:600000] = 1000, 37
    x = np.ascontiguousarray(synth.discount_grid(64)[::9])
    D = len(x)
    for env in ({"STB_ATERMS_GRID": "1", "STB_GRID_C": "4"}, {"STB_ATERMS_GRID": "1", "STB_GRID_C": "2"}, {"STB_ATERMS_GRID": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = create(L, g, n, t, 1200, 1200, D)
        try:
            got = aterms(L, h, x)
            want = np.zeros(D)
            capi.check(L.stb_groups_aterms_tables(h, capi.dp(x), D, capi.dp(want)))
            assert np.all(np.isfinite(got)) and orc.close(got, want, 1e-12), (env, got, want)
            capi.check(L.stb_groups_update_pairs(h, orc.u32p(n), orc.u16p(t)))
            assert np.array_equal(aterms(L, h, x), got)
        finally:
            L.stb_groups_free(h)
        for k in env:
            monkeypatch.delenv(k)
    # no pairs at all
    K0 = np.zeros(5, dtype=np.int32)
    T0 = np.array([3, 1, 4, 1, 5], dtype=np.uint32)
    b0 = np.full(5, 10.0)
    h = L.stb_groups_create(5, orc.i32p(K0), orc.u32p(T0), None, None, orc.dp(b0), 10, 10, 2)
    assert h, capi.last_error()
    try:
        capi.check(L.stb_groups_pairs_begin(h))
        capi.check(L.stb_groups_pairs_commit(h, orc.u32p(T0), orc.dp(b0), 10, 10))
        xs = np.array([0.3, 0.6])
        got = aterms(L, h, xs)
        import math
        for d, xv in enumerate(xs):
            want = sum(float(Ti) * math.log(xv) + math.lgamma(float(Ti) + 10.0 / xv) - math.lgamma(10.0 / xv) for Ti in T0)
            assert abs(got[d] - want) <= 1e-10 * max(1.0, abs(want)), (got[d], want)
    finally:
        L.stb_groups_free(h)
    # ... and samplea on it (lib/samplea.c:184-208: maxn = maxt = 1, a 10 x 10 table nobody reads)
    NP = C.POINTER(C.c_uint32) * 5
    TP = C.POINTER(C.c_uint16) * 5
    orc.seed_libc(777, 12345)
    a = L.samplea(0.5, 5, orc.i32p(K0), orc.u32p(T0), NP(), TP(), None, orc.dp(b0), None, 1, 0)
    assert 0.3 <= a <= 0.7 and L.stb_sampler_trace_count() >= 3     # (ARMS may accept by the squeeze test: no fourth evaluation)
    if orc.have_ref():
        R = orc.ref()
        orc.seed_libc(777, 12345)
        empty_n, empty_t = np.zeros(1, dtype=np.uint32), np.zeros(1, dtype=np.uint16)
        want = R.ref_samplea_flat(0.5, 5, orc.i32p(K0), orc.u32p(T0), orc.u32p(empty_n), orc.u16p(empty_t), orc.dp(b0), 1, 0)
        assert R.ref_trace_count() == L.stb_sampler_trace_count()
        assert abs(a - want) <= 1e-12
    L.stb_sampler_cache_clear()


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_lists_from_the_slab_equal_the_sorted_lists(monkeypatch, seed):
    """random table bounds (N from 520, M anywhere up to N), random ragged restaurants, pairs anywhere -- on the diagonal,
    beyond the bounds, t = 0, n = 1 among them --, a random number of discounts, whatever form is picked: the lists from the
    count slab give the bits of the sorted lists; a hand-over of the same pairs in another order too; stored tables agree"""
    L = capi.lib()
    rng = np.random.default_rng(100 + seed)
    N = int(rng.integers(520, 3200))
    M = int(rng.integers(10, N + 1))
    I = int(rng.integers(1, 40))
    K = rng.integers(0, 400, I).astype(np.int32)
    G = int(K.sum())
    if G == 0:
        K[0] = 5
        G = 5
    n = rng.integers(1, N + 3, G).astype(np.uint32)
    t = np.minimum(rng.integers(0, M + 3, G), 65535).astype(np.uint16)
    ok = rng.random(G) < 0.9                       # most pairs inside the support: 1 <= t <= n <= N, t <= M
    n[ok] = np.clip(n[ok], 2, N)
    t[ok] = np.minimum(np.maximum(t[ok], 1), np.minimum(n[ok], M)).astype(np.uint16)
    finite = bool(np.all((n <= 1) | ((t >= 1) & (t <= n) & (t <= M) & (n <= N))))
    T = np.array([int(t[K[:i].sum():K[:i + 1].sum()].sum()) for i in range(I)], dtype=np.uint32)
    bpar = rng.uniform(0.5, 50.0, I)
    D = int(rng.integers(1, 9))
    x = np.sort(rng.uniform(0.02, 0.97, D))
    if seed % 3 == 1:
        monkeypatch.setenv("STB_ATERMS_GRID", "1")

    def make(nn, tt):
        h = L.stb_groups_create(I, orc.i32p(K), orc.u32p(T), orc.u32p(nn), orc.u16p(tt), orc.dp(bpar), N, M, D)
        assert h, capi.last_error()
        return h

    outs = {}
    for slab in ("1", "0"):
        monkeypatch.setenv("STB_LISTS_SLAB", slab)
        h = make(n, t)
        try:
            outs[slab] = aterms(L, h, x)
            if slab == "1":
                want = np.zeros(D)
                capi.check(L.stb_groups_aterms_tables(h, capi.dp(np.ascontiguousarray(x)), D, capi.dp(want)))
                p = rng.permutation(G)
                capi.check(L.stb_groups_update_pairs(h, orc.u32p(n[p].copy()), orc.u16p(t[p].copy())))
                again = aterms(L, h, x)
        finally:
            L.stb_groups_free(h)
    assert np.array_equal(outs["1"], outs["0"], equal_nan=True), (N, M, G, D, outs)
    assert np.array_equal(again, outs["1"], equal_nan=True)
    if finite:
        assert np.all(np.isfinite(outs["1"])) and orc.close(outs["1"], want, 1e-12), (outs["1"], want)
    else:
        assert np.all(np.isneginf(outs["1"])) and np.all(np.isneginf(want))

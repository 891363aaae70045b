"""GPU parity of samplea / sampleb (include/psample.h) against the reference's end-to-end runs
under the same libc streams (srand(777), srand48(12345)).

The host control code is bit-identical to the reference (tests/test_host_logic.py); what can
differ is the log-posterior value handed to ARMS: the reference accumulates 10^3..10^6 terms
left to right in one double (its own rounding noise is ~1e-13 relative, i.e. up to ~1e-3 absolute
at |y| ~ 1e10), the device sums in double-double.  ARMS only looks at DIFFERENCES of these values,
so the abscissae it visits agree to many digits but not to the last bit; the integer outputs
(number of evaluations, return code) and the bracket are exact.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import orc
from libstb_amd import capi, synth

pytestmark = pytest.mark.gpu
fh = float.fromhex

SETS = {"small_wide": (20, 30, 300, "wide"), "small_real": (20, 30, 300, "realistic"),
        "mid_wide": (100, 100, 1000, "wide"), "big_wide": (1000, 1000, 4000, "wide"),
        "big_real": (1000, 1000, 4000, "realistic")}


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def ragged(g):
    """n[i][k], t[i][k] pointer vectors over the flat arrays, as the reference's callers pass them"""
    NP = C.POINTER(C.c_uint32) * g.I
    TP = C.POINTER(C.c_uint16) * g.I
    n, t = NP(), TP()
    off = 0
    for i in range(g.I):
        n[i] = C.cast(g.n.ctypes.data + 4 * off, C.POINTER(C.c_uint32))
        t[i] = C.cast(g.t.ctypes.data + 2 * off, C.POINTER(C.c_uint16))
        off += int(g.K[i])
    return n, t


def trace(L):
    xs, ys = [], []
    for i in range(L.stb_sampler_trace_count()):
        x, y = C.c_double(), C.c_double()
        L.stb_sampler_trace_get(i, C.byref(x), C.byref(y))
        xs.append(x.value)
        ys.append(y.value)
    return np.array(xs), np.array(ys), L.stb_sampler_trace_code()


# End-to-end bars per recorded run: 10 x the worst deviation observed on MI355X (round 5, gpurun_out/r05/t_new.txt) of the
# adaptive abscissae (relative) and of the draw (absolute), so that a regression of one run is caught by that run's own
# bar -- the reference sums up to 10^6 terms left to right in one double, the device in double-double, and ARMS
# exponentiates differences of these sums: the big sets deviate most.
OBS_A_RELX = [0, 0, 4.8e-13, 4.43e-15, 8e-11, 3.41e-14, 7.23e-14, 1.11e-15, 3.98e-9, 4.76e-15, 2.76e-5]
OBS_A_DA = [3.5e-14, 1.8e-14, 4.29e-13, 3.06e-13, 2.51e-12, 2.65e-14, 5.06e-14, 3.33e-16, 2.36e-10, 3.33e-15, 2.12e-8]
OBS_B_RELX = [4.09e-12, 4.87e-12, 1.27e-11, 0, 4.7e-13, 2.58e-13, 1.4e-12, 0, 4.63e-11, 1.29e-10, 5.01e-9, 0, 5.22e-7, 8.73e-14, 2.62e-6, 0,
              1.53e-9, 1.79e-10, 4.69e-9, 0]
OBS_B_DB = [1.43e-10, 8.03e-11, 1.42e-8, 0, 3.46e-14, 3.31e-12, 1.82e-11, 0, 9.7e-9, 1.05e-8, 2.95e-7, 0, 1.13e-5, 1.75e-10, 1.35e-4, 0,
            7.04e-8, 1.94e-8, 3.89e-8, 0]


def bars(obs_relx, obs_d):
    relx = max(10 * obs_relx, 1e-11)
    return relx, max(10 * obs_d, 1e-12), max(1e-10, 100 * relx)


@pytest.mark.parametrize("rec_index", range(11))
def test_samplea_vs_reference(golden_dir, rec_index):
    L = capi.lib()
    rec = load(golden_dir, "samplers.json")["samplea"][rec_index]
    g = synth.groups(*SETS[rec["set"]])
    n, t = ragged(g)
    orc.seed_libc(777, 12345)
    got = L.samplea(fh(rec["a_in"]), g.I, orc.i32p(g.K), orc.u32p(g.T), n, t, None, orc.dp(g.bpar), None, 1, 0)
    xs, ys, code = trace(L)
    want_x = np.array([fh(v) for v in rec["trace"]["x"]])
    want_y = np.array([fh(v) for v in rec["trace"]["y"]])
    assert code == rec["trace"]["code"]
    assert len(xs) == rec["trace"]["count"], (len(xs), rec["trace"]["count"])   # same number of aterms calls
    assert np.array_equal(xs[:3], want_x[:3])                    # the three starting abscissae are exact
    assert orc.close(ys[:3], want_y[:3], 1e-10)                  # log-posterior parity where x is identical
    relx = np.max(np.abs(xs - want_x) / np.abs(want_x))
    print(f"{rec['set']}: {len(xs)} evals, max rel dx {relx:.2e}, da {abs(got - fh(rec['a_out'])):.2e}")
    bar_x, bar_a, bar_y = bars(OBS_A_RELX[rec_index], OBS_A_DA[rec_index])
    assert relx <= bar_x, (relx, bar_x)                           # adaptive abscissae: see module docstring
    assert orc.close(ys, want_y, bar_y)                           # y at (slightly) different x
    assert abs(got - fh(rec["a_out"])) <= bar_a, (abs(got - fh(rec["a_out"])), bar_a)


@pytest.mark.parametrize("rec_index", range(11))
def test_device_aterms_at_every_recorded_abscissa(golden_dir, rec_index):
    """the device posterior aterms (lib/samplea.c:46-83) at EVERY abscissa the reference's ARMS visited in
    the recorded run -- not only the three starting points that both runs share -- against the value the
    reference computed there, at the parity bar 1e-10 (one discount at a time through a stored table, as
    samplea does, and all of them in one grid call through the fused fill)"""
    L = capi.lib()
    rec = load(golden_dir, "samplers.json")["samplea"][rec_index]
    g = synth.groups(*SETS[rec["set"]])
    want_x = np.array([fh(v) for v in rec["trace"]["x"]])
    want_y = np.array([fh(v) for v in rec["trace"]["y"]])
    M = max(int(g.t.max()) + 1, 10)             # the table samplea builds (lib/samplea.c:60, lib/stable.c:118-129)
    N = max(int(g.n.max()) + 1, M)
    D = len(want_x)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
    assert h, capi.last_error()
    try:
        one = np.zeros(1)
        single = np.zeros(D)
        for d in range(D):
            capi.check(L.stb_groups_aterms(h, capi.dp(want_x[d:d + 1].copy()), 1, capi.dp(one)))
            single[d] = one[0]
        assert orc.close(single, want_y, 1e-10), orc.max_err(single, want_y)
        if D >= 2:
            grid = np.zeros(D)
            capi.check(L.stb_groups_aterms(h, capi.dp(np.ascontiguousarray(want_x)), D, capi.dp(grid)))
            assert orc.close(grid, want_y, 1e-10), orc.max_err(grid, want_y)
    finally:
        L.stb_groups_free(h)


@pytest.mark.parametrize("rec_index", range(20))
def test_device_bterms_at_every_recorded_abscissa(golden_dir, rec_index):
    """the device posterior bterms (lib/sampleb.c:33-41) at every abscissa of the recorded run against the
    reference's value, 1e-10.  Q = 1/scale - sum log Beta(b, N_i) (lib/sampleb.c:90-100) is drawn here as
    sampleb draws it, from the same rand48 stream through the library's own (bit-identical) generator."""
    L = capi.lib()
    rec = load(golden_dir, "samplers.json")["sampleb"][rec_index]
    apar = fh(rec["apar"])
    if apar == 0.0:
        pytest.skip("a = 0: sampleb draws b from a Gamma directly, no posterior is evaluated")
    g = synth.groups(*SETS[rec["set"]])
    want_x = np.array([fh(v) for v in rec["trace"]["x"]])
    want_y = np.array([fh(v) for v in rec["trace"]["y"]])
    orc.seed_libc(777, 12345)
    Q = 1.0 / g.scale
    b_in = fh(rec["b_in"])
    for i in range(g.I):
        if g.N[i] > 0:
            Q -= float(np.log(L.gsl_rng_beta(b_in, float(int(g.N[i])))))
    dg = capi.DeviceGroups(g)
    got = capi.bterms(np.ascontiguousarray(want_x), Q, g.shape, apar, dg).cpu().numpy()
    assert orc.close(got, want_y, 1e-10), orc.max_err(got, want_y)


def test_samplea_keeps_the_pairs_between_calls(monkeypatch):
    """samplea keeps the sorted device copy of the (n,t) pairs while the next call brings the same
    pairs, refreshing T and bpar; a changed pair, shape or STB_SAMPLEA_CACHE=0 builds a new set.  The
    draw never depends on whether the set was reused."""
    L = capi.lib()
    L.stb_sampler_cache_clear()
    g = synth.groups(100, 100, 1000, "wide")
    n, t = ragged(g)

    def draw(a_in, bpar, T=None, nn=n, tt=t):
        orc.seed_libc(777, 12345)
        return L.samplea(a_in, g.I, orc.i32p(g.K), orc.u32p(g.T if T is None else T), nn, tt, None, orc.dp(bpar), None, 1, 0)

    b10, b3 = g.bpar.copy(), np.full(g.I, 3.0)
    monkeypatch.setenv("STB_SAMPLEA_CACHE", "0")
    ref = [draw(0.5, b10), draw(0.5, b3), draw(0.3, b10)]
    monkeypatch.delenv("STB_SAMPLEA_CACHE")
    got = [draw(0.5, b10), draw(0.5, b3), draw(0.3, b10)]     # second and third call reuse the set
    assert got == ref
    assert ref[0] != ref[1]                                    # (bpar matters: the refresh is what made them agree)
    # a different pair set must not be served from the kept one
    g2 = synth.groups(100, 100, 1000, "wide", seed=7)
    n2, t2 = ragged(g2)
    orc.seed_libc(777, 12345)
    a2 = L.samplea(0.5, g2.I, orc.i32p(g2.K), orc.u32p(g2.T), n2, t2, None, orc.dp(g2.bpar), None, 1, 0)
    monkeypatch.setenv("STB_SAMPLEA_CACHE", "0")
    orc.seed_libc(777, 12345)
    assert a2 == L.samplea(0.5, g2.I, orc.i32p(g2.K), orc.u32p(g2.T), n2, t2, None, orc.dp(g2.bpar), None, 1, 0)
    monkeypatch.delenv("STB_SAMPLEA_CACHE")
    # ONE count changed, all shapes the same: the evaluations queued on the guess that the kept set still holds (they
    # are, before the pairs are read) must be thrown away
    L.stb_sampler_cache_clear()
    first = draw(0.5, b10)
    g3 = synth.groups(100, 100, 1000, "wide")
    g3.n[int(np.argmin(g3.n))] += 1                            # (table bounds unchanged: only the hash can tell)
    n3, t3 = ragged(g3)
    changed = draw(0.5, b10, nn=n3, tt=t3)
    monkeypatch.setenv("STB_SAMPLEA_CACHE", "0")
    assert changed == draw(0.5, b10, nn=n3, tt=t3)
    assert first == draw(0.5, b10)
    L.stb_sampler_cache_clear()


def test_aterms_tables_equals_fused_and_single(golden_dir, monkeypatch):
    """stb_groups_aterms_tables (stored tables + gather at any D) against the one-at-a-time path through stored tables
    (STB_ATERMS_FUSE1=0: since round 5 a single discount is summed inside the table walk wherever the lists come from
    the count slab), bit for bit; and against the fused single evaluation at 1e-12"""
    L = capi.lib()
    monkeypatch.setenv("STB_ATERMS_FUSE1", "0")
    g = synth.groups(100, 100, 1000, "wide")
    M = max(int(g.t.max()) + 1, 10)
    N = max(int(g.n.max()) + 1, M)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, 3)
    assert h, capi.last_error()
    try:
        x = np.array([0.3, 0.5, 0.7])
        three, one = np.zeros(3), np.zeros(1)
        capi.check(L.stb_groups_aterms_tables(h, capi.dp(x), 3, capi.dp(three)))
        for d in range(3):
            capi.check(L.stb_groups_aterms(h, capi.dp(x[d:d + 1].copy()), 1, capi.dp(one)))
            assert one[0] == three[d]
        monkeypatch.delenv("STB_ATERMS_FUSE1")
        for d in range(3):
            capi.check(L.stb_groups_aterms(h, capi.dp(x[d:d + 1].copy()), 1, capi.dp(one)))
            assert orc.close(one[0], three[d], 1e-12)
    finally:
        L.stb_groups_free(h)


def test_samplea_getval_callback_equals_arrays():
    L = capi.lib()
    g = synth.groups(20, 30, 300, "wide")
    n, t = ragged(g)
    orc.seed_libc(777, 12345)
    a1 = L.samplea(0.5, g.I, orc.i32p(g.K), orc.u32p(g.T), n, t, None, orc.dp(g.bpar), None, 1, 0)
    offs = np.concatenate([[0], np.cumsum(g.K)])

    @capi.GETVAL
    def getval(pn, pt, i, k):
        pn[0] = int(g.n[offs[i] + k])
        pt[0] = int(g.t[offs[i] + k])

    orc.seed_libc(777, 12345)
    a2 = L.samplea(0.5, g.I, orc.i32p(g.K), orc.u32p(g.T), None, None, getval, orc.dp(g.bpar), None, 1, 0)
    assert a1 == a2


@pytest.mark.parametrize("rec_index", range(20))
def test_sampleb_vs_reference(golden_dir, rec_index):
    L = capi.lib()
    rec = load(golden_dir, "samplers.json")["sampleb"][rec_index]
    g = synth.groups(*SETS[rec["set"]])
    orc.seed_libc(777, 12345)
    got = L.sampleb(fh(rec["b_in"]), g.I, g.shape, g.scale, orc.u32p(g.N), orc.u32p(g.T), fh(rec["apar"]),
                    None, 1, 0)
    want = fh(rec["b_out"])
    if fh(rec["apar"]) == 0.0:
        assert got == want          # Gamma/Gaussian draw only: host RNG path, exact
        return
    xs, ys, code = trace(L)
    want_x = np.array([fh(v) for v in rec["trace"]["x"]])
    want_y = np.array([fh(v) for v in rec["trace"]["y"]])
    assert code == rec["trace"]["code"]
    assert len(xs) == rec["trace"]["count"]
    assert np.array_equal(xs[:3], want_x[:3])
    assert orc.close(ys[:3], want_y[:3], 1e-10)
    relx = np.max(np.abs(xs - want_x) / np.abs(want_x))
    print(f"{rec['set']}: {len(xs)} evals, max rel dx {relx:.2e}, db {abs(got - want):.2e}")
    bar_x, bar_b, bar_y = bars(OBS_B_RELX[rec_index], OBS_B_DB[rec_index])
    assert relx <= bar_x, (relx, bar_x)
    assert orc.close(ys, want_y, bar_y)
    assert abs(got - want) <= bar_b, (abs(got - want), bar_b)


@pytest.mark.parametrize("run", range(8))
def test_samplea_slice_branch_vs_reference(golden_dir, monkeypatch, run):
    """samplea's slice-sampler branch (lib/samplea.c:216-221 -> SliceSimple, lib/sslice.c:33-80) against the reference
    compiled with that branch (oracle/_ref/libstb_ref_slice.so, tests/golden/samplers_slice.json): the bracket, the
    number of posterior evaluations and EVERY abscissa are exact (they depend on the drand48 stream and on comparisons
    of posterior values only), the draw is exact, the posterior values agree at 1e-10"""
    L = capi.lib()
    rec = load(golden_dir, "samplers_slice.json")["runs"][run]
    monkeypatch.setenv("STB_SAMPLER", "slice")
    g = synth.groups(*SETS[rec["set"]])
    n, t = ragged(g)
    orc.seed_libc(777, 12345)
    got = L.samplea(fh(rec["a_in"]), g.I, orc.i32p(g.K), orc.u32p(g.T), n, t, None, orc.dp(g.bpar), None, rec["loops"], 0)
    xs, ys, _ = trace(L)
    want_x = np.array([fh(v) for v in rec["trace"]["x"]])
    want_y = np.array([fh(v) for v in rec["trace"]["y"]])
    assert len(xs) == rec["trace"]["count"], (len(xs), rec["trace"]["count"])
    assert np.array_equal(xs, want_x)
    assert orc.close(ys, want_y, 1e-10), orc.max_err(ys, want_y)
    assert got == fh(rec["a_out"])


def test_sampleb_slice_variant_equals_the_host_slice_sampler_on_the_oracles_posterior(monkeypatch):
    """STB_SAMPLER=slice routes sampleb through SliceSimple too.  The reference's own slice branch of sampleb
    (lib/sampleb.c:141-153) starts from bmax(), which needs digammaInv -- compiled out in the shipped configuration
    (lib/digamma.h:25) -- so nothing can be recorded from it: this variant starts from b_in (DESIGN deviation 7).  What CAN
    be pinned: the run must be SliceSimple (bit-exact against the reference on the CPU, tests/test_host_logic.py) driven by
    bterms.  So the run is repeated on the host -- the same libc streams, Q from the same Beta draws (lib/sampleb.c:90-100),
    the library's SliceSimple with the ORACLE's bterms as its posterior -- and must give the same abscissae, one for one, the
    same draw, and posterior values within 1e-10."""
    L = capi.lib()
    O = orc.oracle()
    monkeypatch.setenv("STB_SAMPLER", "slice")
    L.gsl_rng_beta.restype = C.c_double
    L.gsl_rng_beta.argtypes = [C.c_double, C.c_double]
    POST = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)
    L.SliceSimple.restype = C.c_int
    L.SliceSimple.argtypes = [C.POINTER(C.c_double), POST, C.POINTER(C.c_double), C.c_void_p, C.c_int, C.c_void_p]
    for seed, (I, K, nmax, prof), b_in, apar, loops in ((12345, (20, 30, 300, "realistic"), 10.0, 0.5, 2), (99, (300, 40, 2000, "wide"), 150.0, 0.3, 3),
                                                      (7, (1000, 1, 500, "realistic"), 1.5, 0.8, 1)):
        g = synth.groups(I, K, nmax, prof)
        orc.seed_libc(777, seed)
        b_dev = L.sampleb(b_in, g.I, g.shape, g.scale, orc.u32p(g.N), orc.u32p(g.T), apar, None, loops, 0)
        xs, ys, _ = trace(L)
        assert 0.01 <= b_dev <= 2000 and len(xs) >= loops
        # the same run on the host
        orc.seed_libc(777, seed)
        Q = 1.0 / g.scale
        for i in range(g.I):
            if g.N[i] > 0:
                Q -= np.log(L.gsl_rng_beta(b_in, float(int(g.N[i]))))
        hx, hy = [], []

        def post(x, _):
            y = O.orc_bterms(x, Q, g.shape, g.I, orc.u32p(g.T), apar)
            hx.append(x)
            hy.append(y)
            return y

        b = C.c_double(b_in)
        bounds = (C.c_double * 3)(0.01, 2000.0, 2000.0)
        assert L.SliceSimple(C.byref(b), POST(post), bounds, None, loops, None) == 0
        assert np.array_equal(xs, np.array(hx)), (xs, hx)
        assert orc.close(ys, np.array(hy), 1e-10), orc.max_err(ys, np.array(hy))
        assert b_dev == b.value


def test_rand_stream_is_not_disturbed_by_the_runtime():
    """the HIP runtime draws from rand() while initialising; entry points must shield the caller's
    stream (ARMS consumes it, lib/arms.c:913-918)"""
    libc = C.CDLL(None)
    libc.rand.restype = C.c_int
    L = capi.lib()
    g = synth.groups(3, 20, 30, "wide")
    libc.srand(1)
    want = [libc.rand() for _ in range(3)]
    libc.srand(1)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar),
                            40, 40, 1)
    got = [libc.rand()]
    x, out = np.array([0.4]), np.zeros(1)
    L.stb_groups_aterms(h, capi.dp(x), 1, capi.dp(out))
    got.append(libc.rand())
    L.stb_groups_free(h)
    got.append(libc.rand())
    assert got == want


@pytest.mark.parametrize("run", range(6))
def test_samplea2_vs_reference(golden_dir, run):
    """8f-4: samplea2 (the S-free discount step) against the reference built with -DSAMPLEA_M: the
    sampled table sizes are identical, ARMS makes the same number of evaluations with the same code,
    its three starting abscissae are identical and the posterior there agrees to 1e-10, the draw to 1e-6"""
    import ctypes as C
    import hashlib

    L = capi.lib()
    rec = load(golden_dir, "samplea2.json")["runs"][run]
    g = synth.groups(*SETS[rec["set"]])
    n, t = ragged(g)
    a0 = fh(rec["a_in"])
    tab = capi.Table(rec["maxn"], rec["maxt"], rec["maxn"], rec["maxt"], a0, capi.S_STABLE)
    orc.seed_libc(777, 12345)
    got = L.samplea2(a0, tab.sp, g.I, orc.i32p(g.K), orc.u32p(g.T), n, t, None, orc.dp(g.bpar), None, 1, 0)
    mp = capi.c_u16_p()
    cnt = L.stb_samplea2_partition(C.byref(mp))
    assert cnt == rec["m_count"]
    m = np.ctypeslib.as_array(mp, shape=(cnt,)).copy()
    assert [int(v) for v in m[:64]] == rec["m_head"]
    assert hashlib.sha256(m.tobytes()).hexdigest() == rec["m_sha256"]
    xs, ys, code = trace(L)
    want_x = np.array([fh(v) for v in rec["trace"]["x"]])
    want_y = np.array([fh(v) for v in rec["trace"]["y"]])
    assert code == rec["trace"]["code"] and len(xs) == rec["trace"]["count"]
    assert np.array_equal(xs[:3], want_x[:3])
    assert orc.close(ys[:3], want_y[:3], 1e-10)
    assert np.max(np.abs(xs - want_x) / np.abs(want_x)) <= 1e-4
    assert abs(got - fh(rec["a_out"])) <= 1e-6 * abs(fh(rec["a_out"]))
    # the posterior itself, for that partition, at fixed abscissae (one batched device call)
    hist = np.zeros(rec["maxn"] + 2, dtype=np.uint32)
    mm = 0
    for nn, tt in zip(g.n, g.t):
        nn, tt = int(nn), int(tt)
        if nn == 0 or tt == nn or tt == 0 or tt > nn:
            continue
        if tt == 1:
            hist[nn] += 1
            continue
        sizes = m[mm:mm + tt - 1]
        mm += tt - 1
        for s in sizes:
            if s > 1:
                hist[s] += 1
        rest = nn - int(sizes.astype(np.int64).sum())
        if rest > 1:
            hist[rest] += 1
    h = L.stb_hist_create(orc.u32p(hist), hist.shape[0], g.I, orc.u32p(g.T), orc.dp(g.bpar))
    assert h, capi.last_error()
    x = np.array([fh(p["x"]) for p in rec["aterms2"]])
    out = np.zeros(len(x))
    capi.check(L.stb_hist_aterms2(h, capi.dp(x), len(x), capi.dp(out)))
    L.stb_hist_free(h)
    assert orc.close(out, [fh(p["y"]) for p in rec["aterms2"]], 1e-10)
    tab.free()


def test_bterms_in_one_launch_has_the_bits_of_two(monkeypatch):
    """sampleb's posterior over at most 2048 restaurants is ONE launch (k_bterms_one) since round 5: the same operations in
    the same order as the partial-sum + final-reduce pair, so the same bits (STB_BTERMS_ONE=0 is the pair)"""
    L = capi.lib()
    g = synth.groups(1000, 10, 4000, "realistic")
    x = np.ascontiguousarray(np.linspace(0.5, 900.0, 23))
    outs = []
    for one in ("1", "0"):
        monkeypatch.setenv("STB_BTERMS_ONE", one)
        c = L.stb_bterms_create(orc.u32p(g.T), g.I)
        assert c, capi.last_error()
        try:
            o = np.zeros(len(x))
            capi.check(L.stb_bterms_eval(c, capi.dp(x), len(x), 0.05, 1.1, 0.37, capi.dp(o)))
            outs.append(o)
        finally:
            L.stb_bterms_free(c)
    assert np.all(np.isfinite(outs[0])) and np.array_equal(outs[0], outs[1])


def test_spin_wait_and_stream_wait_give_the_same_samplers(monkeypatch):
    """one-abscissa evaluations return through a word the last kernel writes to pinned memory (the host spins on it,
    STB_SPIN_WAIT=1, the default) or through a wait for the stream (0): the same draws, abscissae and evaluation counts
    of sampleb and samplea"""
    import ctypes as C
    L = capi.lib()
    g = synth.groups(60, 50, 900, "wide")
    nn = (C.POINTER(C.c_uint32) * g.I)()
    tt = (C.POINTER(C.c_uint16) * g.I)()
    off = 0
    for i in range(g.I):
        nn[i] = C.cast(g.n.ctypes.data + 4 * off, C.POINTER(C.c_uint32))
        tt[i] = C.cast(g.t.ctypes.data + 2 * off, C.POINTER(C.c_uint16))
        off += int(g.K[i])
    runs = {}
    for spin in ("1", "0"):
        monkeypatch.setenv("STB_SPIN_WAIT", spin)
        L.stb_sampler_cache_clear()
        orc.seed_libc(777, 12345)
        b = L.sampleb(10.0, g.I, g.shape, g.scale, orc.u32p(g.N), orc.u32p(g.T), 0.3, None, 1, 0)
        nb = L.stb_sampler_trace_count()
        orc.seed_libc(777, 12345)
        a = L.samplea(0.5, g.I, orc.i32p(g.K), orc.u32p(g.T), nn, tt, None, orc.dp(g.bpar), None, 1, 0)
        xs = []
        for i in range(L.stb_sampler_trace_count()):
            x, y = C.c_double(), C.c_double()
            L.stb_sampler_trace_get(i, C.byref(x), C.byref(y))
            xs.append((x.value, y.value))
        runs[spin] = (b, nb, a, xs)
    assert runs["1"] == runs["0"], runs
    assert 0.0 < runs["1"][2] < 1.0 and runs["1"][0] > 0.0 and len(runs["1"][3]) >= 4
    L.stb_sampler_cache_clear()

"""Host-side control code of the product (ARMS, slice sampler, variate generators, closed forms,
growth policy) against the golden fixtures dumped from the reference -- bit for bit, since this is
plain C on the host with the same libm.  No GPU needed: the log-densities are analytic callbacks."""
import ctypes as C
import json
import math
import os
import re

import numpy as np
import pytest

import orc
from libstb_amd import capi

fh = float.fromhex


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def density(kind, p0, p1, p2, calls):
    """the analytic log-densities of oracle/ref_shim.c:ref_density, same expression order"""

    def f(x, _):
        calls.append(x)
        if kind == 0:
            return -0.5 * (x - p0) * (x - p0) / (p1 * p1)
        if kind == 1:
            return (p0 - 1.0) * math.log(x) - p1 * x
        if kind == 2:
            return (p0 - 1.0) * math.log(x) + (p1 - 1.0) * math.log(1.0 - x)
        if kind == 3:
            return -p2 * ((x - p0) * (x - p0) * (x - p0) * (x - p0)) - p1 * x
        return math.log(math.exp(-0.5 * (x - p0) * (x - p0)) + math.exp(-0.5 * (x - p1) * (x - p1)))

    return capi.LOGDENS(f)


def test_arms_matches_reference_bitwise(golden_dir):
    L = capi.lib()
    cases = load(golden_dir, "arms.json")
    assert len(cases) >= 50
    for c in cases:
        p0, p1, p2 = (fh(v) for v in c["p"])
        calls = []
        cb = density(c["kind"], p0, p1, p2, calls)
        orc.seed_libc(c["seed"], 12345)
        xl, xr = C.c_double(fh(c["xl"])), C.c_double(fh(c["xr"]))
        xprev, xsamp = C.c_double(fh(c["xprev"])), C.c_double(0.0)
        code = L.arms_simple(3, C.byref(xl), C.byref(xr), cb, None, c["dometrop"], C.byref(xprev),
                             C.byref(xsamp))
        assert code == c["code"], c
        assert len(calls) == c["ncalls"], c
        assert [x.hex() for x in calls] == c["xs"], c
        if code == 0:
            assert xsamp.value.hex() == c["xsamp"], c


@pytest.mark.ref
def test_arms_many_seeds_against_live_reference():
    L, R = capi.lib(), orc.ref()
    rng = np.random.default_rng(5)
    for trial in range(300):
        kind = int(rng.integers(0, 5))
        p0, p1, p2 = {0: (0.3, 0.1, 0.0), 1: (3.0, 2.0, 0.0), 2: (2.0, 5.0, 0.0), 3: (0.4, 2.0, 500.0),
                      4: (-2.0, 2.5, 0.0)}[kind]
        xl, xr = {0: (-1.0, 2.0), 1: (0.01, 40.0), 2: (0.01, 0.98), 3: (0.01, 0.98), 4: (-6.0, 6.0)}[kind]
        seed = int(rng.integers(1, 2 ** 31))
        metro = int(rng.integers(0, 2))
        xprev = xl + 0.37 * (xr - xl)
        orc.seed_libc(seed, 1)
        xs = np.zeros(256)
        want, nc = C.c_double(), C.c_int()
        code_r = R.ref_arms_probe(kind, p0, p1, p2, xl, xr, metro, xprev, C.byref(want), C.byref(nc),
                                  orc.dp(xs), 256)
        calls = []
        cb = density(kind, p0, p1, p2, calls)
        orc.seed_libc(seed, 1)
        a, b, pv, got = C.c_double(xl), C.c_double(xr), C.c_double(xprev), C.c_double()
        code = L.arms_simple(3, C.byref(a), C.byref(b), cb, None, metro, C.byref(pv), C.byref(got))
        assert code == code_r
        assert len(calls) == nc.value
        assert np.array_equal(np.array(calls[:256]), xs[:min(nc.value, 256)])
        if code == 0:
            assert got.value == want.value


def test_arms_argument_checks():
    """return codes of lib/arms.c:287-316, :155-160"""
    L = capi.lib()
    cb = density(0, 0.0, 1.0, 0.0, [])
    xl, xr, xp, xs = C.c_double(-1), C.c_double(1), C.c_double(0), C.c_double()
    assert L.arms_simple(2, C.byref(xl), C.byref(xr), cb, None, 0, C.byref(xp), C.byref(xs)) == 1001
    conv, q, xc, ne = C.c_double(1.0), C.c_double(50.0), C.c_double(), C.c_int()
    xi = (C.c_double * 3)(-0.5, 0.0, 0.5)
    args = lambda xinit, npoint, cv, qq, ncent: L.arms(xinit, 3, C.byref(xl), C.byref(xr), cb, None, C.byref(cv),
                                                       npoint, 0, C.byref(xp), C.byref(xs), 1, C.byref(qq),
                                                       C.byref(xc), ncent, C.byref(ne))
    assert args(xi, 6, conv, q, 0) == 1002                      # too few envelope points
    assert args((C.c_double * 3)(-1.0, 0.0, 0.5), 100, conv, q, 0) == 1003   # xinit on the bound
    assert args((C.c_double * 3)(-0.5, 0.6, 0.5), 100, conv, q, 0) == 1004   # not ascending
    assert args(xi, 100, conv, C.c_double(101.0), 1) == 1005    # centile out of range
    assert args(xi, 100, C.c_double(-1.0), q, 0) == 1008        # negative convexity
    orc.seed_libc(3, 3)
    assert args(xi, 100, conv, q, 1) == 0 and -1 < xc.value < 1  # centile is returned
    # Metropolis with the previous iterate outside the bounds
    xp2 = C.c_double(5.0)
    assert L.arms_simple(3, C.byref(xl), C.byref(xr), cb, None, 1, C.byref(xp2), C.byref(xs)) == 1007
    assert xs.value == 1.0


def test_expshift():
    L = capi.lib()
    assert L.expshift(3.0, 1.0) == math.exp(3.0 - 1.0 + 50.0)
    assert L.expshift(-200.0, 0.0) == 0.0


def test_slice_sampler_matches_reference_bitwise(golden_dir):
    L = capi.lib()
    for c in load(golden_dir, "slice.json"):
        p0, p1, p2 = (fh(v) for v in c["p"])
        calls = []
        cb = density(c["kind"], p0, p1, p2, calls)
        orc.seed_libc(777, c["seed48"])
        x = C.c_double(fh(c["x0"]))
        bounds = (C.c_double * 2)(fh(c["lo"]), fh(c["hi"]))
        err = L.SliceSimple(C.byref(x), cb, bounds, None, c["loops"], None)
        assert err == c["err"], c
        assert len(calls) == c["ncalls"], c
        assert x.value.hex() == c["x"], c


def test_variate_streams_match_reference_bitwise(golden_dir):
    L = capi.lib()
    g = load(golden_dir, "rng.json")
    orc.seed_libc(777, g["seed48"])
    assert [L.gsl_rng_gaussian_ziggurat(1.0).hex() for _ in range(64)] == g["gaussian"]
    for a in (0.3, 1.0, 2.5, 100.0):
        orc.seed_libc(777, g["seed48"])
        assert [L.gsl_rng_gamma(a).hex() for _ in range(32)] == g[f"gamma_{a}"], a
    orc.seed_libc(777, g["seed48"])
    assert [L.gsl_rng_beta(10.0, 500.0).hex() for _ in range(32)] == g["beta_10_500"]


@pytest.mark.skipif(not os.path.exists("/root/reference/lib/gslrandist.c"), reason="reference tree absent")
def test_ziggurat_tables_rebuilt_exactly():
    """all 384 published table entries are reproduced from the level recursion (read as text)"""
    L = capi.lib()
    src = open("/root/reference/lib/gslrandist.c").read()

    def table(name):
        m = re.search(r"static const (?:double|unsigned long) " + name + r"\[128\] = \{(.*?)\};", src, re.S)
        return [t.strip() for t in m.group(1).replace("\n", " ").split(",") if t.strip()]

    for which, name in ((0, "ytab"), (1, "wtab"), (2, "ktab")):
        want = [float(t.rstrip("UL")) for t in table(name)]
        got = [L.stb_zig_table(which, i) for i in range(128)]
        assert got == want, name


def test_zig_table_sanity():
    L = capi.lib()
    assert L.stb_zig_table(0, 0) == 1.0 and L.stb_zig_table(2, 0) == 0.0
    assert abs(L.stb_zig_table(1, 126) * 2 ** 24 - 3.44428647676) < 1e-10


def test_sapprox_and_digamma_bitwise(golden_dir):
    L = capi.lib()
    g = load(golden_dir, "sapprox.json")
    for r in g["digamma"]:
        assert L.digammaRN(fh(r["x"])).hex() == r["digamma"]
    for r in g["rows"]:
        a = fh(r["a"])
        for fn, key in ((L.S_approx, "S_approx"), (L.S_approx_da, "S_approx_da")):
            got, want = fn(r["n"], r["m"], a), fh(r[key])
            if math.isnan(want):
                assert math.isnan(got), r
            else:
                assert got == want, (key, r, got)


def test_extend_policy_matches_reference_trace(golden_dir):
    L = capi.lib()
    for tr in load(golden_dir, "extend_trace.json"):
        usedN, usedM, maxN, maxM = tr["made"][:4]
        uv = bool(tr["flags"] & 2)
        for st in tr["steps"]:
            n, m = st["n"], st["m"]
            if n != m and m != 1 and not (n < m or m == 0) and (m > usedM or n > usedN) \
                    and not (n > maxN or m > maxM):
                a, b = C.c_uint(), C.c_uint()
                L.stb_extend_policy(usedN, usedM, maxN, maxM, n + 1, m + 1, C.byref(a), C.byref(b))
                usedN, usedM = a.value, b.value
            assert (usedN, usedM) == (st["usedN"], st["usedM"]), (tr["init"], st)
            if uv and m >= 2:
                if (m >= usedM - 1 or n >= usedN - 1) and not (n > maxN or m > maxM):
                    a, b = C.c_uint(), C.c_uint()
                    L.stb_extend_policy(usedN, usedM, maxN, maxM, n + 1, m + 1, C.byref(a), C.byref(b))
                    usedN, usedM = a.value, b.value
                assert (usedN, usedM) == (st["usedN_afterV"], st["usedM_afterV"]), st


def test_extend_policy_equals_oracle_on_random_requests():
    L, O = capi.lib(), orc.oracle()
    rng = np.random.default_rng(11)
    for _ in range(3000):
        maxN = int(rng.integers(10, 5000))
        maxM = int(rng.integers(10, maxN + 1))
        usedN = int(rng.integers(10, maxN + 1))
        usedM = int(rng.integers(10, min(usedN, maxM) + 1))
        n = int(rng.integers(1, maxN + 1))
        m = int(rng.integers(1, min(n, maxM) + 1))
        a, b, c, d = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint()
        L.stb_extend_policy(usedN, usedM, maxN, maxM, n + 1, m + 1, C.byref(a), C.byref(b))
        O.orc_extend_policy(usedN, usedM, maxN, maxM, n + 1, m + 1, C.byref(c), C.byref(d))
        assert (a.value, b.value) == (c.value, d.value)


def test_yaps_sink_is_pluggable():
    L = capi.lib()
    seen = []

    @C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
    def sink(fmt, ap):
        seen.append(fmt)

    L.yaps_yapper(sink)
    try:
        L.yaps_message.argtypes = [C.c_char_p]
        L.yaps_message(b"hello %d\n")
        assert seen == [b"hello %d\n"]
    finally:
        L.yaps_yapper(None)


def test_fill_workspace_covers_every_smaller_batch():
    """a workspace sized for D tables is also used for smaller batches (stb_groups with Dmax): the
    size query is pure host arithmetic and must not shrink when D grows, whatever block shape the
    chain form picks for a batch"""
    L = capi.lib()
    for (N, M) in ((2, 2), (3, 2), (200, 50), (4000, 4000), (10000, 10000), (20000, 700)):
        prev = 0
        for D in (1, 2, 3, 4, 5, 6, 8, 16, 20, 21, 64):
            need = int(L.stb_fill_workspace_bytes(N, M, D))
            assert need >= prev, (N, M, D, need, prev)
            prev = need

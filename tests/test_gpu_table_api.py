"""GPU parity of the reference's own table interface (include/stable.h) -- S_make, accessors,
growth, asymptote -- against the golden fixtures dumped from the reference."""
import json
import math
import os

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from libstb_amd import capi

pytestmark = pytest.mark.gpu
fh = float.fromhex
TOL = 1e-10


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def same(x, y, tol=TOL):
    if math.isinf(y) or math.isinf(x):
        return x == y
    return abs(x - y) <= tol * max(1.0, abs(y))


def test_config1_plumbing(golden_dir):
    """configs[0]: S_make N=200 M=50 a=0.5 then the values test/list.c prints"""
    t = capi.Table(200, 50, 200, 50, 0.5, capi.S_STABLE)
    assert (t.usedN, t.usedM, t.maxN, t.maxM) == (200, 50, 200, 50)
    assert same(t.S(200, 2), 855.40717151108788)
    assert same(t.S(200, 25), 815.82736174046067)
    assert same(t.S(200, 49), 744.08298941125418)
    assert same(t.asympt(200, 25), 814.80290541542593)
    z = np.load(os.path.join(golden_dir, "stable_200x50.npz"))
    want = z["a0.5_table"]
    for n in range(3, 201):
        for m in range(2, min(n - 1, 50) + 1):
            assert same(t.S(n, m), want[orc.row_offset(n, 50) + m - 2])
    # identities (lib/stable.c:944-949)
    assert t.S(7, 7) == 0.0 and t.S(5, 9) == -math.inf and t.S(5, 0) == -math.inf
    assert same(t.S(150, 1), z["a0.5_S1"][149])
    assert t.S(300, 3) == -math.inf  # beyond maxN without S_ASYMPT
    # remake for another discount
    assert t.remake(0.125) == 0
    want = z["a0.125_table"]
    for (n, m) in ((3, 2), (50, 20), (200, 50), (200, 2)):
        assert same(t.S(n, m), want[orc.row_offset(n, 50) + m - 2])
    t.free()


def test_flags_rejected_like_reference():
    L = capi.lib()
    assert not L.S_make(20, 10, 20, 10, 0.5, 0)            # neither table: NULL (lib/stable.c:131)
    assert not L.S_make(20, 10, 20, 10, 1.5, capi.S_STABLE)  # discount outside [0,1)


def test_growth_trace_and_lazy_S1(golden_dir):
    """integer outputs (usedN, usedM) must equal the reference's after every accessor call"""
    for tr in load(golden_dir, "extend_trace.json"):
        t = capi.Table(*tr["init"], fh(tr["a"]), tr["flags"])
        assert [t.usedN, t.usedM, t.maxN, t.maxM] == tr["made"][:4], tr["init"]
        for st in tr["steps"]:
            got = t.S(st["n"], st["m"])
            assert same(got, fh(st["S"])), (tr["init"], st, got)
            assert (t.usedN, t.usedM) == (st["usedN"], st["usedM"]), (tr["init"], st)
            if tr["flags"] & capi.S_UVTABLE and st.get("V") is not None:
                v = t.V(st["n"], st["m"])
                assert same(v, fh(st["V"]), 1e-13), (st, v)
                assert (t.usedN, t.usedM) == (st["usedN_afterV"], st["usedM_afterV"]), st
        t.free()
        # lazy S1 on a fresh table: values are log Gamma(n-a)/Gamma(1-a); the reference returns the
        # value of the GROWN index instead of n (lib/stable.c:845-871) -- see DESIGN.md, deviations
        t = capi.Table(*tr["init"], fh(tr["a"]), tr["flags"])
        a = fh(tr["a"])
        for st in tr["S1_lazy_fresh"]:
            n = st["n"]
            got = t.S1(n)
            if n == 0 or n > t.maxN:
                assert got == -math.inf
            else:
                assert same(got, math.lgamma(n - a) - math.lgamma(1 - a), 1e-12), (n, got)
        t.free()


def test_asympt_and_beyond_max(golden_dir):
    cache = {}
    for r in load(golden_dir, "asympt.json"):
        a = fh(r["a"])
        if a not in cache:
            cache[a] = capi.Table(20, 10, 20, 10, a, capi.S_STABLE | capi.S_ASYMPT)
        t = cache[a]
        assert same(t.asympt(r["n"], r["m"]), fh(r["direct"]), 1e-13), r
        if r["m"] != 1:
            assert same(t.S(r["n"], r["m"]), fh(r["via_S_S"]), 1e-13), r   # S_S -> S_asympt past maxN
        else:
            # S_S(n,1) past maxN: exact lgamma form here (reference answers for n=maxN instead)
            assert same(t.S(r["n"], 1), math.lgamma(r["n"] - a) - math.lgamma(1 - a), 1e-12)
    for t in cache.values():
        t.free()


def test_uv_accessors(golden_dir):
    tabs = {}
    for r in load(golden_dir, "uv_access.json"):
        a = fh(r["a"])
        if a not in tabs:
            tabs[a] = capi.Table(200, 50, 200, 50, a, capi.S_STABLE | capi.S_UVTABLE)
        t = tabs[a]
        n, m = r["n"], r["m"]
        if r["V"] is not None:
            assert same(t.V(n, m), fh(r["V"]), 1e-13), r
        if r["U"] is not None:
            assert same(t.U(n, m), fh(r["U"]), 1e-13), r
        got, want = t.UV(n, m), fh(r["UV"])
        assert same(got, want, 1e-13), r
    z = np.load(os.path.join(golden_dir, "uv_200x50.npz"))
    t = tabs[0.5]
    v = z["a0.5_V"]
    pos = 0
    for n in range(2, 201):
        ln = min(n - 1, 49)
        if n < 198:  # S_V grows the table when n >= usedN-1 (lib/stable.c:903); stay below
            for m in (2, 2 + ln // 2, 1 + ln):
                if m < 48:
                    assert t.V(n, m) == v[pos + m - 2], (n, m)
        pos += ln
    for t in tabs.values():
        t.free()


def test_uv_only_table_has_S1():
    t = capi.Table(100, 30, 100, 30, 0.4, capi.S_UVTABLE)
    assert t.S(50, 3) == -math.inf               # no S table (lib/stable.c:942-943)
    assert same(t.S1(60), math.lgamma(60 - 0.4) - math.lgamma(0.6), 1e-12)
    assert same(t.U(10, 1), 10 - 0.4)
    t.free()


def test_report_format(tmp_path):
    import ctypes as C
    L = capi.lib()
    t = capi.Table(200, 50, 300, 60, 0.5, capi.S_STABLE | capi.S_UVTABLE)
    L.S_tag(t.sp, b"unit")
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    path = str(tmp_path / "rep.txt").encode()
    fp = libc.fopen(path, b"w")
    L.S_report(t.sp, fp)
    libc.fclose(fp)
    text = open(path).read()
    # lib/stable.c:1027-1038
    assert text.startswith("S-table 'unit': a=0.500000, N=200/300, M=50/60, +S+U/V double mem=")
    assert text.endswith("k\n\n")
    t.free()


def test_threads_flag_growth_keeps_old_rows_alive():
    t = capi.Table(20, 10, 400, 300, 0.3, capi.S_STABLE | capi.S_THREADS)
    before = t.S(15, 5)
    for (n, m) in ((30, 5), (120, 40), (399, 250)):
        t.S(n, m)
    assert t.S(15, 5) == before  # same arithmetic at any bounds -> identical bits
    t.free()


def test_float_storage_matches_reference(golden_dir):
    """S_FLOAT (next row 8f-2): values are the double results narrowed to float; the reference's
    float tables must be reproduced up to the rare double difference that straddles a float
    rounding boundary (1 float ulp)."""
    z = np.load(os.path.join(golden_dir, "uv_200x50.npz"))
    for key, a in (("a0.5", 0.5), ("a0.05", 0.05), ("a0.95", 0.95)):
        t = capi.Table(200, 50, 200, 50, a, capi.S_STABLE | capi.S_UVTABLE | capi.S_FLOAT)
        sf, vf = z[key + "_Sf"], z[key + "_Vf"]
        ps = pv = 0
        exact = total = 0
        for n in range(2, 198):
            if n >= 3:
                ln = min(n - 2, 49)
                for m in (2, 2 + ln // 2, 1 + ln):
                    if m <= 47:
                        got, want = np.float32(t.S(n, m)), sf[ps + m - 2]
                        assert abs(float(got) - float(want)) <= float(np.spacing(np.abs(want))), (n, m, got, want)
                        exact += int(got == want)
                        total += 1
                ps += ln
            ln = min(n - 1, 49)
            for m in (2, 2 + ln // 2):
                if m <= 47 and m <= n:
                    got, want = np.float32(t.V(n, m)), vf[pv + m - 2]
                    assert abs(float(got) - float(want)) <= float(np.spacing(np.abs(want))), (n, m, got, want)
            pv += ln
        assert exact >= 0.99 * total
        assert t.S(200, 25) == float(np.float32(t.S(200, 25)))  # really stored as float
        t.free()


def test_float_table_holds_half_the_device_memory():
    """S_FLOAT as the reference means it (lib/stable.h:80-90): halved storage.  A 10^4 x 10^4 float table is written
    once, as floats, and no double slab exists: its device slabs are half a double table's (the fill workspace is the
    same for both); values equal the double table's, narrowed"""
    import ctypes as C
    L = capi.lib()
    N = 10000
    d = capi.Table(N, N, N, N, 0.5, capi.S_STABLE)
    f = capi.Table(N, N, N, N, 0.5, capi.S_STABLE | capi.S_FLOAT)
    dv, fv, hs = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
    L.stb_table_bytes(d.sp, C.byref(dv), C.byref(hs))
    L.stb_table_bytes(f.sp, C.byref(fv), C.byref(hs))
    ws = int(L.stb_fill_workspace_bytes(N, N, 1))
    slab_d, slab_f = dv.value - ws, fv.value - ws
    assert slab_d > 4e8 and 0.49 < slab_f / slab_d < 0.52, (slab_d, slab_f, ws)
    for (n, m) in ((3, 2), (4000, 17), (9999, 9998), (10000, 5000), (7777, 7000)):
        assert f.S(n, m) == float(np.float32(d.S(n, m))), (n, m)
    d.free()
    f.free()


def test_float_table_grows():
    t = capi.Table(20, 10, 300, 200, 0.4, capi.S_STABLE | capi.S_FLOAT)
    d = capi.Table(20, 10, 300, 200, 0.4, capi.S_STABLE)
    for (n, m) in ((30, 5), (120, 40), (299, 150)):
        assert t.S(n, m) == float(np.float32(d.S(n, m)))
        assert (t.usedN, t.usedM) == (d.usedN, d.usedM)
        assert t.usedM >= m and t.usedN >= n
    # (299,150) needs two growth steps: the first caps usedM at the old usedN=122 (the reference
    # then reads past the row end; see stable_host.c, deviations)
    S1, tab = orc.fill_S(0.4, d.usedN, d.usedM)
    assert same(d.S(299, 150), tab[orc.row_offset(299, d.usedM) + 148])
    t.free()
    d.free()


def test_lazy_mirror_copies_only_touched_blocks(monkeypatch):
    """the host mirror is filled 128 rows at a time on first touch: S_make / S_remake copy nothing,
    an accessor copies its block, stb_table_sync (or STB_MIRROR=eager) copies everything; values are
    the same either way"""
    N = 3000
    S1, tab = orc.fill_S(0.45, N, N)
    t = capi.Table(N, N, N, N, 0.45, capi.S_STABLE | capi.S_UVTABLE)
    assert t.mirrored() == (0, 0)
    assert same(t.S(2000, 700), tab[orc.row_offset(2000, N) + 698])
    assert t.mirrored() == (1, 0)
    assert same(t.S(2001, 2), tab[orc.row_offset(2001, N)])          # same block
    assert t.mirrored() == (1, 0)
    t.V(2999, 5)
    assert t.mirrored() == (1, 1)
    assert t.remake(0.2) == 0
    assert t.mirrored() == (0, 0)                                      # new discount: stale blocks dropped
    S1b, tabb = orc.fill_S(0.2, N, N)
    for (n, m) in ((3, 2), (130, 100), (131, 129), (2999, 2998), (3000, 1500)):
        assert same(t.S(n, m), tabb[orc.row_offset(n, N) + m - 2])
    t.sync()
    nb = (N - 3) // 128 + 1
    assert t.mirrored() == (nb, (N - 2) // 128 + 1)
    t.free()
    monkeypatch.setenv("STB_MIRROR", "eager")
    t = capi.Table(N, N, N, N, 0.45, capi.S_STABLE)
    assert t.mirrored() == (nb, 0)
    assert same(t.S(2000, 700), tab[orc.row_offset(2000, N) + 698])
    t.free()


def test_concurrent_readers_while_the_table_grows():
    """S_THREADS (reference README:66-76): S_S / S_V may be called from several threads; growth is
    serialised by the table's mutex and publishes complete storage before the bounds that admit
    readers to it.  Two reader threads hammer cells inside the initial bounds (and whatever has been
    published since) while the main thread makes the table grow step by step; every value read must
    equal the oracle's, and the rand() stream of the process must come out untouched by the HIP
    runtime calls the threads make (the guard swaps the state at depth 0 <-> 1, see csrc/abi.hip)."""
    import ctypes as C
    import threading
    import time

    libc = C.CDLL(None)
    libc.rand.restype = C.c_int
    libc.srand(4242)
    want_rand = [libc.rand() for _ in range(3)]
    libc.srand(4242)

    a, maxN, maxM = 0.37, 1500, 1200
    S1, tab = orc.fill_S(a, maxN, maxM)
    Vt = orc.fill_V(a, maxN, maxM)
    L = orc.oracle()
    t = capi.Table(40, 20, maxN, maxM, a, capi.S_STABLE | capi.S_UVTABLE | capi.S_THREADS)
    stop = threading.Event()
    errors = []

    def reader(seed):
        rng = np.random.default_rng(seed)
        k = 0
        while not stop.is_set():
            N, M = t.usedN, t.usedM            # bounds are published last: everything below them is readable
            n = int(rng.integers(3, N - 1))
            m = int(rng.integers(2, min(n - 1, M - 2) + 1))
            got = t.S(n, m)
            want = L.orc_S_S(orc.dp(tab), orc.dp(S1), maxN, maxM, n, m)
            if not same(got, want):
                errors.append(("S", n, m, got, want))
                return
            gv = t.V(n, m)
            wv = L.orc_S_V(orc.dp(Vt), maxN, maxM, n, m)
            if abs(gv - wv) > 1e-10 * max(1.0, abs(wv)):   # (512 rows and up: V^n_m from the S recurrence's cells, 1e-10)
                errors.append(("V", n, m, gv, wv))
                return
            k += 1
            count[seed] = k

    count = {1: 0, 2: 0}
    th = [threading.Thread(target=reader, args=(s,)) for s in (1, 2)]
    for x in th:
        x.start()
    try:
        for (n, m) in ((60, 10), (200, 50), (201, 199), (700, 300), (1200, 900), (1499, 1100), (1500, 1199)):
            got = t.S(n, m)                    # grows the table under the readers
            assert same(got, L.orc_S_S(orc.dp(tab), orc.dp(S1), maxN, maxM, n, m)), (n, m)
            floor = min(count.values()) + 40   # let both readers work on every generation
            t_end = time.time() + 20
            while min(count.values()) < floor and not errors and time.time() < t_end:
                time.sleep(0.001)
    finally:
        stop.set()
        for x in th:
            x.join()
    assert not errors, errors[:3]
    assert min(count.values()) >= 7 * 40
    assert (t.usedN, t.usedM) == (maxN, maxM)
    t.free()
    assert [libc.rand() for _ in range(3)] == want_rand


def test_quit_on_bound_is_fatal_like_the_reference(tmp_path):
    """S_QUITONBOUND (lib/stable.c:954-959): overrunning maxN/maxM ends the process through
    yaps_quit (exit code 1, message on stderr) -- checked in a child process"""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r); "
        "from libstb_amd import capi; "
        "t = capi.Table(20, 10, 60, 30, 0.5, capi.S_STABLE | capi.S_QUITONBOUND); "
        "L = capi.lib(); L.S_tag(t.sp, b'bounded'); "
        "print('inside', t.S(50, 20) > 0, flush=True); "
        "t.S(61, 5); print('not reached', flush=True)" % ROOT
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stdout, r.stderr)
    assert "inside True" in r.stdout and "not reached" not in r.stdout
    assert "S_S(61,5,0.500000) tagged 'bounded' hit bounds" in r.stderr


def test_uv_lookups_on_the_device(golden_dir):
    """stb_lookup_V / _U / _UV: the ratio table's accessors (lib/stable.c:875-939) over a device-resident V slab, against
    the reference's recorded S_V / S_U / S_UV and against the host accessors on the same cells"""
    import torch

    L = capi.lib()
    recs = load(golden_dir, "uv_access.json")
    for a in sorted({fh(r["a"]) for r in recs}):
        rs = [r for r in recs if fh(r["a"]) == a and r["n"] <= 200 and r["m"] <= 50]
        V = capi.DeviceVTables(200, 50, D=1)
        V.fill([a])
        n = torch.tensor([r["n"] for r in rs] + [7, 9, 300, 40, 3], dtype=torch.int32, device="cuda")
        m = torch.tensor([r["m"] for r in rs] + [1, 10, 5, 51, 4], dtype=torch.int32, device="cuda")   # + identities and out-of-bounds cells
        out = [torch.empty(len(n), dtype=torch.float64, device="cuda") for _ in range(3)]
        capi.check(L.stb_lookup_V(V.tables.data_ptr(), 200, 50, n.data_ptr(), m.data_ptr(), len(n), out[0].data_ptr(), None))
        capi.check(L.stb_lookup_U(V.tables.data_ptr(), 200, 50, a, n.data_ptr(), m.data_ptr(), len(n), out[1].data_ptr(), None))
        capi.check(L.stb_lookup_UV(V.tables.data_ptr(), 200, 50, a, n.data_ptr(), m.data_ptr(), len(n), out[2].data_ptr(), None))
        v, u, uv = (o.cpu().numpy() for o in out)
        for i, r in enumerate(rs):
            if r["V"] is not None:
                assert same(v[i], fh(r["V"]), 1e-13), r
            if r["U"] is not None:
                assert same(u[i], fh(r["U"]), 1e-13), r
            assert same(uv[i], fh(r["UV"]), 1e-13), r
        k = len(rs)
        assert v[k] == 0.0 and u[k] == 7 - a and uv[k] == -np.inf          # m = 1 (lib/stable.c:876, :887)
        assert v[k + 1] == 0.0                                             # n < m
        assert v[k + 2] == 0.0 and v[k + 3] == 0.0                         # outside the slab's bounds: 0 (lib/stable.c:922)
        assert uv[k + 4] == 1.0                                            # m = n + 1 (lib/stable.c:890)


def test_mirror_look_ahead_gives_the_lazy_mirrors_values(monkeypatch):
    import ctypes as C

    """round 6: a miss copies its block and sends the blocks behind it on their way asynchronously.  Every value an accessor
    returns is the one the block-by-block mirror (STB_MIRROR=lazy) returns; a rebuild and growth with copies still under
    way are safe; runs of 1 MB and of 16 MB alike; the V table too"""
    N = M = 2600
    rng = np.random.default_rng(5)
    n = rng.integers(3, N + 1, 4000).astype(np.uint32)
    m = (2 + (rng.random(4000) * (np.minimum(n - 1, M) - 1))).astype(np.uint32)
    u32p = C.POINTER(C.c_uint)
    L = capi.lib()

    def probe(t, which=0):
        out = np.zeros(len(n))
        L.stb_table_probe(t.sp, which, n.ctypes.data_as(u32p), m.ctypes.data_as(u32p), len(n), capi.dp(out))
        return out

    monkeypatch.setenv("STB_MIRROR", "lazy")
    lazy = capi.Table(N, M, N, M, 0.37, capi.S_STABLE | capi.S_UVTABLE)
    want_s, want_v = probe(lazy, 0), probe(lazy, 1)
    lazy.remake(0.61)
    want_s2 = probe(lazy, 0)
    lazy.free()
    for mb in ("1", "16"):
        monkeypatch.delenv("STB_MIRROR")
        monkeypatch.setenv("STB_MIRROR_AHEAD_MB", mb)
        t = capi.Table(N, M, N, M, 0.37, capi.S_STABLE | capi.S_UVTABLE)
        assert t.mirrored() == (0, 0)
        first = t.S(40, 7)                                    # block 0, and a run behind it sets out
        assert t.mirrored()[0] >= 1
        assert np.array_equal(probe(t, 0), want_s) and np.array_equal(probe(t, 1), want_v)
        assert t.S(40, 7) == first
        t.remake(0.61)                                        # everything invalid again
        assert t.mirrored() == (0, 0)
        t.S(1500, 700)                                        # a miss in the middle: runs under way ...
        t.remake(0.37)                                        # ... when the slabs are written again
        t.S(2600, 1300)
        t.remake(0.61)
        assert np.array_equal(probe(t, 0), want_s2)
        t.free()
        monkeypatch.setenv("STB_MIRROR", "lazy")
    monkeypatch.delenv("STB_MIRROR")
    # growth with copies under way (the device slabs are replaced): values of the grown table
    g = capi.Table(600, 200, N, M, 0.37, capi.S_STABLE)
    g.S(100, 50)
    got = probe(g, 0)                                         # grows on the way, several times
    assert np.array_equal(got, want_s)
    g.free()

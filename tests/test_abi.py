"""The C-ABI shared library loads on a machine without a GPU, exports every function that
include/*.h declares, and refuses to compute there (no CPU fallback)."""
import ctypes as C
import glob
import os
import re

import pytest

from libstb_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DECL = re.compile(r"^[A-Za-z_][\w \t\*]*?[\s\*]([A-Za-z_]\w*)\s*\(", re.M)


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)       # comments
        text = re.sub(r"^\s*#.*?$", "", text, flags=re.M)       # preprocessor lines
        text = re.sub(r"\\\n", "", text)
        for m in DECL.finditer(text):
            name = m.group(1)
            if name in ("defined", "sizeof"):
                continue
            names.add(name)
    return names


def test_every_declared_symbol_is_exported():
    L = capi.lib()
    want = declared_functions()
    # the reference's own API must be among them
    for must in ("S_make", "S_remake", "S_free", "S_tag", "S_S", "S_S1", "S_U", "S_UV", "S_V", "S_asympt",
                 "S_report", "samplea", "sampleb", "SliceSimple", "arms", "arms_simple", "expshift",
                 "yaps_message", "yaps_quit", "yaps_sysquit", "yaps_yapper", "S_approx", "S_approx_da",
                 "gsl_rng_gamma", "gsl_rng_beta", "gsl_rng_gaussian_ziggurat",
                 "stb_fill_S", "stb_fill_V", "stb_sweep_S", "stb_restaurant_terms", "stb_bterms",
                 "stb_lookup_S", "stb_groups_create", "stb_groups_aterms"):
        assert must in want, f"{must} not declared in include/"
    missing = [n for n in sorted(want) if not hasattr(L, n)]
    assert not missing, f"declared but not exported: {missing}"


def test_layout_queries_are_pure():
    L = capi.lib()
    assert L.stb_cells(200, 50) == 8526
    assert L.stb_cells(4000, 4000) == 7994001
    assert L.stb_cells(10000, 10000) == 49985001
    assert L.stb_elems(10000, 10000) >= 49985001
    assert L.stb_rowoff(3, 50) == 0 and L.stb_rowoff(4, 50) == 320 and L.stb_rowoff(5, 50) == 640
    # rows start 512-byte aligned and keep >= 256 elements of slack after the last stored value
    for (N, M) in ((200, 50), (77, 76), (1000, 13), (300, 300)):
        for n in range(3, N + 1):
            assert L.stb_rowoff(n, M) % 64 == 0
            assert L.stb_rowoff(n + 1, M) - L.stb_rowoff(n, M) >= min(n - 2, M - 1) + 256
        assert L.stb_elems(N, M) == L.stb_rowoff(N + 1, M)


@pytest.mark.skipif(capi.lib().stb_device_count() > 0, reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    L = capi.lib()
    msgs = []

    @C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
    def sink(fmt, ap):
        msgs.append(fmt)

    L.yaps_yapper(sink)
    try:
        sp = L.S_make(200, 50, 200, 50, 0.5, capi.S_STABLE)
        assert not sp, "S_make must fail without a device"
        assert msgs and b"no HIP device" in msgs[0]
        g = L.stb_groups_create(0, None, None, None, None, None, 10, 10, 1)
        assert not g
        assert b"no HIP device" in L.stb_last_error()
    finally:
        L.yaps_yapper(None)

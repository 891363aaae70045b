"""GPU parity of the SUPERSEDED fill forms (tools/ablation/ablation.hip: k_fill_bfp, k_rec + k_logconv,
k_fill_rows<SCALED>, k_fill_chainx), kept for A/B measurements only -- not part of the product library
and not collected by the repo's test suite.  Build and run on a GPU box from the repo root:

    make -C tools/ablation
    STB_LIB_PATH=$PWD/libstb_amd/lib/libstb_amd_ablation.so python -m pytest tools/ablation -q
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import orc  # noqa: E402
from libstb_amd import capi  # noqa: E402

TOL = 1e-10
VARIANTS = [capi.FILL_SCALED_STEP, capi.FILL_SPLIT, capi.FILL_FUSED, capi.FILL_CHAINX]
pytestmark = pytest.mark.skipif(not capi.lib().stb_has_ablation(), reason="load libstb_amd_ablation.so through STB_LIB_PATH")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("N,M", [(2, 2), (3, 2), (3, 3), (10, 10), (64, 64), (65, 33), (97, 96), (130, 129), (500, 7),
                                 (1000, 1000), (1500, 260)])
def test_ragged_shapes_vs_oracle(N, M, variant):
    a = np.array([0.31, 0.77])
    T = capi.DeviceTables(N, M, D=2)
    T.fill(a, variant)
    for d in range(2):
        S1, tab = orc.fill_S(a[d], N, M)
        assert orc.close(T.packed_host(d), tab, TOL), (N, M, orc.max_err(T.packed_host(d), tab))
        assert orc.close(T.S1[d].cpu().numpy(), S1, TOL)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("a", [0.5, 0.1, 0.9])
def test_4000_full_table_vs_oracle(a, variant):
    N = 4000
    T = capi.DeviceTables(N, N, D=1)
    T.fill([a], variant)
    S1, tab = orc.fill_S(a, N, N)
    assert orc.max_err(T.packed_host(0), tab) <= TOL


@pytest.mark.parametrize("variant", [capi.FILL_SPLIT, capi.FILL_FUSED, capi.FILL_CHAINX])
@pytest.mark.parametrize("a", [0.0, 0.01, 0.07, 0.5, 0.98])
def test_growth_next_to_the_diagonal(a, variant, monkeypatch):
    monkeypatch.setenv("STB_FILL_R", "120")
    monkeypatch.setenv("STB_FILL_C", "2")
    N = 6000
    T = capi.DeviceTables(N, N, D=1)
    T.fill([a], variant)
    S1, tab = orc.fill_S(a, N, N)
    got = T.packed_host(0)
    assert np.all(np.isfinite(got)) and orc.max_err(got, tab) <= TOL


@pytest.mark.parametrize("P", [1, 2, 4])
@pytest.mark.parametrize("N,M", [(900, 700), (2500, 2500), (3000, 130)])
def test_chainx_geometries_agree(monkeypatch, P, N, M):
    monkeypatch.setenv("STB_CHAINX_P", str(P))
    a = np.array([0.07, 0.6])
    T = capi.DeviceTables(N, M, D=2)
    T.tables.fill_(float("nan"))
    T.fill(a, capi.FILL_CHAINX)
    T.status()
    for d in range(2):
        S1, tab = orc.fill_S(a[d], N, M)
        got = T.packed_host(d)
        assert np.all(np.isfinite(got)) and orc.close(got, tab, TOL), (d, orc.max_err(got, tab))

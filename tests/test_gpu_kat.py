"""Known-answer tests of the DEVICE table that do not go through the recurrence: every other table test
compares with the oracle, which restates the same recurrence S^n_m = (n-1-ma) S^{n-1}_m + S^{n-1}_{m-1}.  Here the
default fill form is checked against closed forms:

  * S_approx(n, m, a) for m = 2, 3, 4 (reference lib/sapprox.c:28-71: sums of Gamma-function ratios), valid to ~1e-13
    for dyadic a with m a < 1 (`a` is a float, and lgamma loses its sign where 1 - m a < 0);
  * S^n_1 = Gamma(n - a) / Gamma(1 - a);   S^3_2 = 3 - 3a;   S^n_n = 1;
  * the V ratios' identity S_UV(n, n) = (n + 1)/(n - 1)... is the reference's convention and is covered in
    test_gpu_table_api.py; not repeated here."""
import math

import numpy as np
import pytest

import orc
from libstb_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("a", [1 / 16, 1 / 8, 3 / 16, 7 / 32])
def test_device_table_against_the_closed_forms_of_sapprox(a):
    """N = 2000 rows in the default form (halo blocks: 512 rows and up), columns 2..4 of every row against S_approx at
    1e-12 of max(1, |value|) (the reference's own table agrees with S_approx to 4e-13 there)"""
    L = capi.lib()
    N, M = 2000, 64
    t = capi.Table(N, M, N, M, a, capi.S_STABLE)
    try:
        form = L.stb_fill_tuning(N, M, 1, None, None, None)
        assert form == 6, form                  # the halo-block form: the kernel the headline number is measured on
        worst = 0.0
        for m in (2, 3, 4):
            for n in range(m + 1, N + 1):
                want = L.S_approx(n, m, a)
                got = t.S(n, m)
                worst = max(worst, abs(got - want) / max(1.0, abs(want)))
        print(f"a={a}: max |table - S_approx| / max(1,|S_approx|) = {worst:.2e}")
        assert worst <= 1e-12, worst
    finally:
        t.free()


@pytest.mark.parametrize("a", [0.0, 0.07, 0.37, 0.5, 2 / 3, 0.95])
def test_device_table_identities(a):
    """S^n_1 against lgamma, S^3_2 = log(3 - 3a) (zero at a = 2/3: absolute bar), the diagonal, log 0 outside the support"""
    N, M = 1500, 1500
    t = capi.Table(N, M, N, M, a, capi.S_STABLE)
    try:
        assert abs(t.S(3, 2) - math.log(3 - 3 * a)) <= 1e-14
        lg1 = math.lgamma(1 - a)
        for n in (2, 3, 10, 100, 600, 1499, 1500):
            want = math.lgamma(n - a) - lg1
            assert abs(t.S(n, 1) - want) <= 1e-13 * max(1.0, abs(want)), (n, t.S(n, 1), want)
            assert t.S(n, n) == 0.0
            assert t.S(n, n + 1) == -math.inf
        # S^n_{n-1} = sum_{k=1}^{n-1} k (1 - a) ... in closed form: C(n,2) (1 - a)
        for n in (3, 4, 50, 1000, 1500):
            want = math.log(n * (n - 1) / 2 * (1 - a))
            assert abs(t.S(n, n - 1) - want) <= 1e-13 * max(1.0, abs(want)), (n, t.S(n, n - 1), want)
    finally:
        t.free()

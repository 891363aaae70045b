"""ctypes bindings for the CHECKERS: oracle/liboracle.so (this repo's CPU restatement) and, when
present, oracle/_ref/libstb_ref.so (the real reference compiled by oracle/Makefile).

Test infrastructure only -- the product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from functools import lru_cache

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libstb_ref.so")

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)
c_u16_p = C.POINTER(C.c_uint16)
c_int_p = C.POINTER(C.c_int)
c_u64_p = C.POINTER(C.c_uint64)


def dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(c_double_p)


def u32p(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_u32_p)


def u16p(a: np.ndarray):
    assert a.dtype == np.uint16 and a.flags.c_contiguous
    return a.ctypes.data_as(c_u16_p)


def i32p(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_int_p)


def build_oracle() -> None:
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


@lru_cache(maxsize=None)
def oracle() -> C.CDLL:
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    L = C.CDLL(ORACLE_SO)
    u, d, i = C.c_uint, C.c_double, C.c_int
    L.orc_cells.restype = C.c_uint64
    L.orc_cells.argtypes = [u, u]
    L.orc_row_offset.restype = C.c_uint64
    L.orc_row_offset.argtypes = [u, u]
    L.orc_row_len.restype = u
    L.orc_row_len.argtypes = [u, u]
    L.orc_vcells.restype = C.c_uint64
    L.orc_vcells.argtypes = [u, u]
    L.orc_vrow_offset.restype = C.c_uint64
    L.orc_vrow_offset.argtypes = [u, u]
    L.orc_logadd.restype = d
    L.orc_logadd.argtypes = [d, d]
    L.orc_fill_S.restype = None
    L.orc_fill_S.argtypes = [d, u, u, c_double_p, c_double_p]
    L.orc_fill_V.restype = None
    L.orc_fill_V.argtypes = [d, u, u, c_double_p]
    L.orc_S_S.restype = d
    L.orc_S_S.argtypes = [c_double_p, c_double_p, u, u, u, u]
    L.orc_S_V.restype = d
    L.orc_S_V.argtypes = [c_double_p, u, u, u, u]
    L.orc_S_U.restype = d
    L.orc_S_U.argtypes = [c_double_p, d, u, u, u, u]
    L.orc_S_UV.restype = d
    L.orc_S_UV.argtypes = [c_double_p, d, u, u, u, u]
    L.orc_S_asympt.restype = d
    L.orc_S_asympt.argtypes = [d, u, u]
    L.orc_extend_policy.restype = None
    L.orc_extend_policy.argtypes = [u, u, u, u, i, i, C.POINTER(u), C.POINTER(u)]
    L.orc_make_clamp.restype = None
    L.orc_make_clamp.argtypes = [C.POINTER(u)] * 4
    L.orc_aterms_sum.restype = d
    L.orc_aterms_sum.argtypes = [d, i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p, c_double_p,
                                 c_double_p, u, u]
    L.orc_aterms.restype = d
    L.orc_aterms.argtypes = [d, i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p, u, u,
                             c_double_p]
    L.orc_scan_bounds.restype = None
    L.orc_scan_bounds.argtypes = [i, c_int_p, c_u32_p, c_u16_p, c_int_p, c_int_p]
    L.orc_bterms.restype = d
    L.orc_bterms.argtypes = [d, d, d, i, c_u32_p, d]
    L.orc_aterms2.restype = d
    L.orc_aterms2.argtypes = [d, i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p, c_u16_p]
    L.orc_partition.restype = C.c_size_t
    L.orc_partition.argtypes = [d, c_double_p, c_double_p, u, u, i, c_int_p, c_u32_p, c_u16_p, c_double_p, c_u16_p]
    L.orc_S_approx.restype = d
    L.orc_S_approx.argtypes = [i, i, C.c_float]
    L.orc_rows_stream.restype = i
    L.orc_rows_stream.argtypes = [d, u, u, C.POINTER(C.c_uint), i, c_double_p, i]
    L.orc_time_fill.restype = d
    L.orc_time_fill.argtypes = [d, u, u, i, c_double_p, c_double_p]
    L.orc_time_fill_rows.restype = d
    L.orc_time_fill_rows.argtypes = [d, u, u, u, c_double_p, c_double_p, c_u64_p]
    L.orc_time_fill_batch.restype = d
    L.orc_time_fill_batch.argtypes = [c_double_p, i, u, u, i]
    return L


def have_ref() -> bool:
    return os.path.exists(REF_SO)


REF_SLICE_SO = os.path.join(ORACLE_DIR, "_ref", "libstb_ref_slice.so")


def have_ref_slice() -> bool:
    return os.path.exists(REF_SLICE_SO)


@lru_cache(maxsize=None)
def ref_slice() -> C.CDLL:
    """The real reference with samplea's slice-sampler branch (lib/samplea.c:216-221; oracle/Makefile); build container only."""
    L = C.CDLL(REF_SLICE_SO)
    d, i = C.c_double, C.c_int
    L.ref_samplea_flat.restype = d
    L.ref_samplea_flat.argtypes = [d, i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p, i, i]
    L.ref_trace_count.restype = i
    L.ref_trace_code.restype = i
    for name in ("ref_trace_x", "ref_trace_y"):
        getattr(L, name).restype = d
        getattr(L, name).argtypes = [i]
    L.ref_trace_xl.restype = d
    L.ref_trace_xr.restype = d
    return L


REF_M_SO = os.path.join(ORACLE_DIR, "_ref", "libstb_ref_m.so")


def have_ref_m() -> bool:
    return os.path.exists(REF_M_SO)


@lru_cache(maxsize=None)
def ref_m() -> C.CDLL:
    """The real reference compiled with -DSAMPLEA_M (samplea2 / aterms2); build container only."""
    L = C.CDLL(REF_M_SO)
    u, d, i, vp = C.c_uint, C.c_double, C.c_int, C.c_void_p
    L.S_make.restype = vp
    L.S_make.argtypes = [u, u, u, u, d, C.c_uint32]
    L.S_free.restype = None
    L.S_free.argtypes = [vp]
    L.ref_samplea2_flat.restype = d
    L.ref_samplea2_flat.argtypes = [d, vp, i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p, i, i]
    L.ref_m_size.restype = C.c_size_t
    L.ref_m_get.restype = u
    L.ref_m_get.argtypes = [C.c_size_t]
    L.ref_aterms2_eval.restype = d
    L.ref_aterms2_eval.argtypes = [d, i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p, c_u16_p]
    L.ref_trace_count.restype = i
    L.ref_trace_code.restype = i
    L.ref_trace_x.restype = d
    L.ref_trace_x.argtypes = [i]
    L.ref_trace_y.restype = d
    L.ref_trace_y.argtypes = [i]
    L.ref_trace_xl.restype = d
    L.ref_trace_xr.restype = d
    return L


@lru_cache(maxsize=None)
def ref() -> C.CDLL:
    """The real reference (only where oracle/_ref was built)."""
    L = C.CDLL(REF_SO)
    u, d, i, vp = C.c_uint, C.c_double, C.c_int, C.c_void_p
    L.S_make.restype = vp
    L.S_make.argtypes = [u, u, u, u, d, C.c_uint32]
    L.S_remake.restype = i
    L.S_remake.argtypes = [vp, d]
    L.S_free.restype = None
    L.S_free.argtypes = [vp]
    for name in ("S_S", "S_U", "S_V", "S_UV", "S_asympt"):
        f = getattr(L, name)
        f.restype = d
        f.argtypes = [vp, u, u]
    L.S_S1.restype = d
    L.S_S1.argtypes = [vp, u]
    for name in ("ref_usedN", "ref_usedM", "ref_usedN1", "ref_maxN", "ref_maxM", "ref_startM",
                 "ref_memalloced"):
        f = getattr(L, name)
        f.restype = u
        f.argtypes = [vp]
    L.ref_lga.restype = d
    L.ref_lga.argtypes = [vp]
    L.ref_a.restype = d
    L.ref_a.argtypes = [vp]
    L.ref_sizeof_stable.restype = C.c_size_t
    L.ref_copy_S_row.restype = u
    L.ref_copy_S_row.argtypes = [vp, u, c_double_p]
    L.ref_copy_V_row.restype = u
    L.ref_copy_V_row.argtypes = [vp, u, c_double_p]
    L.ref_copy_Sf_row.restype = u
    L.ref_copy_Sf_row.argtypes = [vp, u, c_float_p]
    L.ref_copy_Vf_row.restype = u
    L.ref_copy_Vf_row.argtypes = [vp, u, c_float_p]
    if hasattr(L, "ref_probe"):
        L.ref_probe.restype = None
        L.ref_probe.argtypes = [vp, i, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_size_t, c_double_p]
    L.ref_copy_S1.restype = u
    L.ref_copy_S1.argtypes = [vp, c_double_p, u]
    L.ref_aterms_open.restype = vp
    L.ref_aterms_open.argtypes = [i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p]
    L.ref_aterms_eval.restype = d
    L.ref_aterms_eval.argtypes = [vp, d]
    L.ref_aterms_maxn.restype = i
    L.ref_aterms_maxn.argtypes = [vp]
    L.ref_aterms_maxt.restype = i
    L.ref_aterms_maxt.argtypes = [vp]
    L.ref_aterms_close.restype = None
    L.ref_aterms_close.argtypes = [vp]
    L.ref_bterms_eval.restype = d
    L.ref_bterms_eval.argtypes = [d, d, d, i, c_u32_p, d]
    L.ref_samplea_flat.restype = d
    L.ref_samplea_flat.argtypes = [d, i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p, i, i]
    L.sampleb.restype = d
    L.sampleb.argtypes = [d, i, d, d, c_u32_p, c_u32_p, d, vp, i, i]
    L.ref_trace_count.restype = i
    L.ref_trace_code.restype = i
    L.ref_trace_x.restype = d
    L.ref_trace_x.argtypes = [i]
    L.ref_trace_y.restype = d
    L.ref_trace_y.argtypes = [i]
    L.ref_trace_xl.restype = d
    L.ref_trace_xr.restype = d
    L.ref_arms_probe.restype = i
    L.ref_arms_probe.argtypes = [i, d, d, d, d, d, i, d, c_double_p, c_int_p, c_double_p, i]
    L.ref_slice_probe.restype = i
    L.ref_slice_probe.argtypes = [i, d, d, d, d, d, c_double_p, i, c_int_p]
    L.S_approx.restype = d
    L.S_approx.argtypes = [i, i, C.c_float]
    L.S_approx_da.restype = d
    L.S_approx_da.argtypes = [i, i, C.c_float]
    L.gsl_rng_gamma.restype = d
    L.gsl_rng_gamma.argtypes = [d]
    L.gsl_rng_beta.restype = d
    L.gsl_rng_beta.argtypes = [d, d]
    L.gsl_rng_gaussian_ziggurat.restype = d
    L.gsl_rng_gaussian_ziggurat.argtypes = [d]
    L.digammaRN.restype = d
    L.digammaRN.argtypes = [d]
    return L


_libc = C.CDLL(None)
_libc.srand.argtypes = [C.c_uint]
_libc.srand48.argtypes = [C.c_long]


def seed_libc(s_rand: int = 777, s_rand48: int = 12345) -> None:
    """The sampler RNG streams of SURVEY 8d: rand() for ARMS, drand48() for slice/beta/gamma."""
    _libc.srand(s_rand)
    _libc.srand48(s_rand48)


# ---------------------------------------------------------------- oracle convenience wrappers

def fill_S(a: float, N: int, M: int):
    L = oracle()
    S1 = np.zeros(N, dtype=np.float64)
    tab = np.zeros(int(L.orc_cells(N, M)), dtype=np.float64)
    L.orc_fill_S(a, N, M, dp(S1), dp(tab))
    return S1, tab


def fill_V(a: float, N: int, M: int):
    L = oracle()
    v = np.zeros(int(L.orc_vcells(N, M)), dtype=np.float64)
    L.orc_fill_V(a, N, M, dp(v))
    return v


def rows_stream(a: float, N: int, M: int, rows, threads: int = 8):
    """rows (ascending, >= 3) of a table too large to keep on the host: {n: array of log S^n_m, m = 2 .. min(n-1, M)}"""
    rows = np.ascontiguousarray(np.asarray(sorted(rows), dtype=np.uint32))
    out = np.zeros((len(rows), M))
    rc = oracle().orc_rows_stream(a, N, M, rows.ctypes.data_as(C.POINTER(C.c_uint)), len(rows), dp(out), threads)
    assert rc == 0
    return {int(n): out[i, : min(int(n) - 1, M) - 1].copy() for i, n in enumerate(rows)}


def row_offset(n: int, M: int) -> int:
    if n <= 3:
        return 0
    if n <= M + 1:
        k = n - 3
        return k * (k + 1) // 2
    return (M - 1) * M // 2 + (n - M - 2) * (M - 1)


def row_len(n: int, M: int) -> int:
    return 0 if n < 3 else min(n - 2, M - 1)


def close(x, y, rel=1e-10):
    """Parity metric of SURVEY 8c: |x-y| <= rel * max(1,|y|) (values cross zero)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    both_inf = np.isinf(x) & np.isinf(y) & (np.sign(x) == np.sign(y))
    with np.errstate(invalid="ignore"):
        ok = np.abs(x - y) <= rel * np.maximum(1.0, np.abs(y))
    return np.all(ok | both_inf)


def max_err(x, y):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    return float(np.max(np.abs(x - y) / np.maximum(1.0, np.abs(y)))) if x.size else 0.0

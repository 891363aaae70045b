"""ctypes binding of libstb_amd.so -- the C-ABI drop-in for libstb's S-table / sampler path.

Two layers, both thin:

* the reference's own interface (include/stable.h, psample.h, arms.h, yaps.h, sapprox.h), bound
  1:1 so that Python tests read like the reference's C callers (`S_make`, `S_S`, `samplea`, ...);
* the additive device interface (include/stb_hip.h) taking raw device pointers; `torch` is used
  here only to own HBM buffers and streams and to hand their addresses across the ABI.

There is no fallback of any kind: if the shared library is missing the import fails, and if no GPU
is present the library's entry points fail with a message (``stb_last_error``).
"""
from __future__ import annotations

import ctypes as C
import os
from functools import lru_cache

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STB_LIB_PATH") or os.path.join(_HERE, "lib", "libstb_amd.so")

# flag bits of include/stable.h
S_STABLE, S_UVTABLE, S_FLOAT, S_VERBOSE, S_QUITONBOUND, S_THREADS, S_ASYMPT = 1, 2, 4, 8, 16, 32, 64
FILL_SCALED, FILL_LOGDOMAIN, FILL_SCALED_STEP, FILL_SPLIT, FILL_FUSED, FILL_PC, FILL_CHAIN, FILL_CHAINX, FILL_CK, FILL_HB = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9

c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)
c_u16_p = C.POINTER(C.c_uint16)
c_int_p = C.POINTER(C.c_int)
c_float_p = C.POINTER(C.c_float)

LOGDENS = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)
GETVAL = C.CFUNCTYPE(None, c_u32_p, c_u16_p, C.c_uint, C.c_uint)


class StbError(RuntimeError):
    pass


@lru_cache(maxsize=None)
def lib() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise StbError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C libstb_amd/csrc` (there is no pure-Python or CPU implementation)")
    # torch ships its own libamdhip64.so.7; libstb_amd.so needs the same SONAME.  Whichever is
    # loaded first serves both, and two HIP runtimes in one process do not see the GPU reliably,
    # so when torch is importable it goes first and this library runs on torch's runtime (plain
    # C callers get the system ROCm runtime instead).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less environments
        pass
    L = C.CDLL(LIB_PATH)
    u, d, i, vp, u64, sz = C.c_uint, C.c_double, C.c_int, C.c_void_p, C.c_uint64, C.c_size_t

    # entry points added in round 5: an older build loaded through STB_LIB_PATH (tools/ab_lib.py, A/B against an
    # earlier round's library) simply lacks them; everything else must be there
    ROUND5 = {"stb_groups_pairs_begin", "stb_groups_pairs_put", "stb_groups_pairs_put_ragged", "stb_groups_pairs_commit",
              "stb_groups_update_pairs", "stb_groups_fallbacks", "stb_grid_shape", "stb_bterms_update",
              # ... and in round 6
              "stb_table_probe", "stb_slow_launches", "stb_shared_gpu_mode", "stb_set_shared_gpu", "stb_note_launch_span",
              "stb_lookup_V", "stb_lookup_U", "stb_lookup_UV", "stb_groups_aterms_multi", "stb_groups_create_node"}

    def sig(name, res, args):
        try:
            f = getattr(L, name)
        except AttributeError:
            if name in ROUND5 and os.environ.get("STB_LIB_PATH"):
                return
            raise
        f.restype = res
        f.argtypes = args

    # ---- include/stable.h
    sig("S_make", vp, [u, u, u, u, d, C.c_uint32])
    sig("S_tag", None, [vp, C.c_char_p])
    sig("S_remake", i, [vp, d])
    sig("S_free", None, [vp])
    for n in ("S_S", "S_U", "S_UV", "S_V", "S_asympt"):
        sig(n, d, [vp, u, u])
    sig("S_S1", d, [vp, u])
    sig("S_report", None, [vp, vp])
    sig("stb_extend_policy", None, [u, u, u, u, i, i, C.POINTER(u), C.POINTER(u)])
    sig("stb_table_sync", i, [vp])
    sig("stb_table_probe", None, [vp, i, C.POINTER(u), C.POINTER(u), sz, c_double_p])
    sig("stb_table_mirrored", None, [vp, C.POINTER(u), C.POINTER(u)])
    sig("stb_table_bytes", None, [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)])
    # ---- include/yaps.h
    sig("yaps_yapper", None, [vp])
    # ---- include/stb_hip.h
    sig("stb_last_error", C.c_char_p, [])
    sig("stb_device_count", i, [])
    sig("stb_device_name", i, [C.c_char_p, i])
    sig("stb_set_device", i, [i])
    sig("stb_get_device", i, [])
    sig("stb_device_enter", i, [i])
    sig("stb_device_leave", None, [i])
    sig("stb_device_malloc", vp, [sz])
    sig("stb_device_free", None, [vp])
    sig("stb_host_malloc", vp, [sz])
    sig("stb_host_free", None, [vp])
    sig("stb_memcpy_h2d", i, [vp, vp, sz, vp])
    sig("stb_memcpy_d2h", i, [vp, vp, sz, vp])
    sig("stb_stream_sync", i, [vp])
    for n in ("stb_cells", "stb_elems", "stb_vcells", "stb_velems"):
        sig(n, u64, [u, u])
    sig("stb_rowoff", u64, [u, u])
    sig("stb_vrowoff", u64, [u, u])
    sig("stb_fill_workspace_bytes", sz, [u, u, i])
    sig("stb_default_variant", i, [])
    sig("stb_fill_S", i, [c_double_p, i, u, u, vp, u64, vp, u64, vp, sz, i, vp])
    sig("stb_fill_tuning", i, [u, u, i, c_int_p, c_int_p, c_int_p])
    sig("stb_fill_status", i, [])
    sig("stb_fill_fallbacks", C.c_uint, [])
    sig("stb_has_ablation", i, [])
    sig("stb_slow_launches", C.c_uint, [])
    sig("stb_shared_gpu_mode", i, [])
    sig("stb_set_shared_gpu", None, [i])
    sig("stb_note_launch_span", None, [C.c_double, C.c_double])
    sig("stb_fill_profile_begin", None, [])
    sig("stb_fill_profile_end", i, [c_double_p, c_int_p])
    sig("stb_fill_profile_span", C.c_double, [])
    sig("stb_fill_V", i, [c_double_p, i, u, u, vp, u64, vp, sz, vp])
    sig("stb_fill_V_exact", i, [c_double_p, i, u, u, vp, u64, vp, sz, vp])
    sig("stb_fill_Vf", i, [c_double_p, i, u, u, vp, u64, vp, sz, vp])
    sig("stb_fill_Sf", i, [c_double_p, i, u, u, vp, u64, vp, u64, vp, sz, vp])
    sig("stb_fill_takes_kind", i, [u, u, i, i])
    sig("stb_bterms_create", vp, [c_u32_p, i])
    sig("stb_bterms_update", i, [vp, c_u32_p, i])
    sig("stb_bterms_eval", i, [vp, c_double_p, i, d, d, d, c_double_p])
    sig("stb_bterms_free", None, [vp])
    sig("stb_table_to_float", i, [vp, vp, u64, vp])
    sig("stb_lookup_S", i, [vp, vp, u, u, vp, vp, u64, vp, vp])
    sig("stb_lookup_V", i, [vp, u, u, vp, vp, u64, vp, vp])
    sig("stb_lookup_U", i, [vp, u, u, C.c_double, vp, vp, u64, vp, vp])
    sig("stb_lookup_UV", i, [vp, u, u, C.c_double, vp, vp, u64, vp, vp])
    sig("stb_sweep_workspace_bytes", sz, [u64, i])
    sig("stb_sweep_S", i, [vp, u64, vp, u64, i, u, u, vp, vp, u64, vp, vp, sz, vp])
    sig("stb_terms_workspace_bytes", sz, [u64, i])
    sig("stb_restaurant_terms", i, [c_double_p, i, vp, vp, u64, vp, vp, sz, vp])
    sig("stb_bterms", i, [c_double_p, i, d, d, d, vp, u64, vp, vp, sz, vp])
    sig("stb_groups_create", vp, [i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p, u, u, i])
    sig("stb_groups_free", None, [vp])
    sig("stb_groups_aterms", i, [vp, c_double_p, i, c_double_p])
    sig("stb_groups_aterms_tables", i, [vp, c_double_p, i, c_double_p])
    sig("stb_groups_aterms_async", i, [vp, c_double_p, i, c_double_p, vp])
    sig("stb_groups_wait", i, [vp])
    sig("stb_groups_aterms_multi", i, [C.POINTER(vp), i, c_double_p, i, c_double_p])
    sig("stb_groups_create_node", i, [i, i, c_int_p, c_u32_p, c_u32_p, c_u16_p, c_double_p, u, u, i, C.POINTER(vp)])
    sig("stb_groups_aterms_device", i, [vp, c_double_p, i, vp, vp])
    sig("stb_groups_update_restaurants", i, [vp, c_u32_p, c_double_p])
    sig("stb_groups_pairs_begin", i, [vp])
    sig("stb_groups_pairs_put", i, [vp, c_u32_p, c_u16_p, u64, C.POINTER(u), C.POINTER(u)])
    sig("stb_groups_pairs_put_ragged", i, [vp, i, c_int_p, vp, vp, C.POINTER(u), C.POINTER(u)])
    sig("stb_groups_pairs_commit", i, [vp, c_u32_p, c_double_p, u, u])
    sig("stb_groups_update_pairs", i, [vp, c_u32_p, c_u16_p])
    sig("stb_groups_fallbacks", C.c_uint, [])
    sig("stb_grid_shape", i, [u, u, i, c_int_p, c_int_p, c_int_p])
    sig("stb_groups_shape", i, [vp, c_int_p, C.POINTER(u64), C.POINTER(u), C.POINTER(u), c_int_p])
    sig("stb_sampler_cache_clear", None, [])
    sig("stb_groups_aterms_timed", i, [vp, c_double_p, i, c_double_p, c_float_p, c_float_p, c_float_p])
    # optional entry points (present once the sampler host code is linked in)
    for name, res, args in (
        ("arms_simple", i, [i, c_double_p, c_double_p, LOGDENS, vp, i, c_double_p, c_double_p]),
        ("arms", i, [c_double_p, i, c_double_p, c_double_p, LOGDENS, vp, c_double_p, i, i, c_double_p,
                     c_double_p, i, c_double_p, c_double_p, i, c_int_p]),
        ("expshift", d, [d, d]),
        ("SliceSimple", i, [c_double_p, LOGDENS, c_double_p, vp, i, vp]),
        ("samplea", d, [d, i, c_int_p, c_u32_p, C.POINTER(c_u32_p), C.POINTER(c_u16_p), vp, c_double_p,
                        vp, i, i]),
        ("sampleb", d, [d, i, d, d, c_u32_p, c_u32_p, d, vp, i, i]),
        ("samplea2", d, [d, vp, i, c_int_p, c_u32_p, C.POINTER(c_u32_p), C.POINTER(c_u16_p), vp, c_double_p, vp, i, i]),
        ("stb_samplea2_partition", sz, [C.POINTER(c_u16_p)]),
        ("stb_hist_create", vp, [c_u32_p, u, i, c_u32_p, c_double_p]),
        ("stb_hist_aterms2", i, [vp, c_double_p, i, c_double_p]),
        ("stb_hist_free", None, [vp]),
        ("S_approx", d, [i, i, C.c_float]),
        ("S_approx_da", d, [i, i, C.c_float]),
        ("digammaRN", d, [d]),
        ("gsl_rng_gamma", d, [d]),
        ("gsl_rng_beta", d, [d, d]),
        ("gsl_rng_gaussian_ziggurat", d, [d]),
        ("stb_sampler_trace_count", i, []),
        ("stb_sampler_trace_get", i, [i, c_double_p, c_double_p]),
        ("stb_sampler_trace_code", i, []),
        ("stb_zig_table", d, [i, i]),
    ):
        if hasattr(L, name):
            sig(name, res, args)
    return L


ABLATION_VARIANTS = (FILL_SCALED_STEP, FILL_SPLIT, FILL_FUSED, FILL_CHAINX)


def has_variant(variant: int) -> bool:
    """the superseded fill forms are only in the library `make -C tools/ablation` builds"""
    return variant not in ABLATION_VARIANTS or bool(lib().stb_has_ablation())


def last_error() -> str:
    return lib().stb_last_error().decode()


def check(rc: int) -> None:
    if rc != 0:
        raise StbError(last_error())


def dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(c_double_p)


# ------------------------------------------------------------------------------------------------
# host interface, mirroring the reference's callers


class Table:
    """`stable_t *` with the reference's accessors as methods (S_make ... S_free)."""

    def __init__(self, initN, initM, maxN, maxM, a, flags=S_STABLE):
        self.L = lib()
        self.sp = self.L.S_make(initN, initM, maxN, maxM, a, flags)
        if not self.sp:
            raise StbError("S_make returned NULL: " + last_error())

    # struct stable_s leading fields (include/stable.h): maxM, maxN, usedM, usedN, startM
    def _u(self, idx):
        return C.cast(self.sp, C.POINTER(C.c_uint))[idx]

    maxM = property(lambda s: s._u(0))
    maxN = property(lambda s: s._u(1))
    usedM = property(lambda s: s._u(2))
    usedN = property(lambda s: s._u(3))
    startM = property(lambda s: s._u(4))

    def remake(self, a):
        return self.L.S_remake(self.sp, a)

    def S(self, n, m):
        return self.L.S_S(self.sp, n, m)

    def S1(self, n):
        return self.L.S_S1(self.sp, n)

    def V(self, n, m):
        return self.L.S_V(self.sp, n, m)

    def U(self, n, m):
        return self.L.S_U(self.sp, n, m)

    def UV(self, n, m):
        return self.L.S_UV(self.sp, n, m)

    def asympt(self, n, m):
        return self.L.S_asympt(self.sp, n, m)

    def sync(self):
        check(self.L.stb_table_sync(self.sp))

    def mirrored(self):
        """(S blocks, V blocks) of 128 rows copied to the host so far"""
        s, v = C.c_uint(), C.c_uint()
        self.L.stb_table_mirrored(self.sp, C.byref(s), C.byref(v))
        return s.value, v.value

    def free(self):
        if self.sp:
            self.L.S_free(self.sp)
            self.sp = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# device interface (torch owns the buffers)


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise StbError("no GPU visible to torch; the device interface has no CPU path")
    return torch


def stream_ptr(stream=None):
    torch = _torch()
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


class DeviceTables:
    """D log-Stirling tables resident in HBM, filled by one batched call (K1/K2)."""

    def __init__(self, N: int, M: int, D: int = 1, device="cuda"):
        torch = _torch()
        self.L = lib()
        self.N, self.M, self.D = N, M, D
        self.cells = int(self.L.stb_cells(N, M))
        self.elems = int(self.L.stb_elems(N, M))
        self.stride = max(32, (self.elems + 31) // 32 * 32)
        self.tables = torch.empty((D, self.stride), dtype=torch.float64, device=device)
        self.S1 = torch.empty((D, N), dtype=torch.float64, device=device)
        self.ws_bytes = int(self.L.stb_fill_workspace_bytes(N, M, D))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.a = None

    def fill(self, a, variant=FILL_SCALED, stream=None):
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(a, dtype=np.float64)))
        assert a.shape[0] == self.D
        self.a = a
        check(self.L.stb_fill_S(dp(a), self.D, self.N, self.M, self.tables.data_ptr(), self.stride,
                                self.S1.data_ptr(), self.N, self.ws.data_ptr(), self.ws_bytes,
                                variant, stream_ptr(stream)))

    def status(self):
        """wait for the last fill of this thread and raise if a chain-form fill gave up"""
        check(self.L.stb_fill_status())

    def rowoff(self, n):
        return int(self.L.stb_rowoff(n, self.M))

    def row(self, d, n):
        """values m=2..min(n-1,M) of row n as a device tensor view"""
        ln = min(n - 2, self.M - 1)
        o = self.rowoff(n)
        return self.tables[d, o:o + ln]

    def packed_host(self, d=0):
        """the table without row padding, in the oracle's packed order (numpy)"""
        t = self.tables[d].cpu().numpy()
        N, M = self.N, self.M
        out = np.empty(self.cells, dtype=np.float64)
        pos = 0
        for n in range(3, N + 1):
            ln = min(n - 2, M - 1)
            o = self.rowoff(n)
            out[pos:pos + ln] = t[o:o + ln]
            pos += ln
        return out

    def lookup(self, n, m, d=0, stream=None):
        torch = _torch()
        n = torch.as_tensor(np.asarray(n, dtype=np.uint32).view(np.int32), device=self.tables.device)
        m = torch.as_tensor(np.asarray(m, dtype=np.uint32).view(np.int32), device=self.tables.device)
        out = torch.empty(n.shape[0], dtype=torch.float64, device=self.tables.device)
        check(self.L.stb_lookup_S(self.tables[d].data_ptr(), self.S1[d].data_ptr(), self.N, self.M,
                                  n.data_ptr(), m.data_ptr(), n.shape[0], out.data_ptr(),
                                  stream_ptr(stream)))
        return out.cpu().numpy()


class DeviceFloatTables(DeviceTables):
    """D log-Stirling tables stored as floats (S_FLOAT), written once by the fill that narrows before the store"""

    def __init__(self, N: int, M: int, D: int = 1, device="cuda"):
        super().__init__(N, M, D, device)
        torch = _torch()
        self.tables = torch.empty((D, self.stride), dtype=torch.float32, device=device)

    def fill(self, a, stream=None):
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(a, dtype=np.float64)))
        self.a = a
        check(self.L.stb_fill_Sf(dp(a), self.D, self.N, self.M, self.tables.data_ptr(), self.stride,
                                 self.S1.data_ptr(), self.N, self.ws.data_ptr(), self.ws_bytes, stream_ptr(stream)))

    def packed_host(self, d=0):
        t = self.tables[d].cpu().numpy()
        out = np.empty(self.cells, dtype=np.float32)
        pos = 0
        for n in range(3, self.N + 1):
            ln = min(n - 2, self.M - 1)
            o = self.rowoff(n)
            out[pos:pos + ln] = t[o:o + ln]
            pos += ln
        return out


class DeviceVTables:
    def __init__(self, N: int, M: int, D: int = 1, device="cuda", dtype="f64"):
        torch = _torch()
        self.L = lib()
        self.N, self.M, self.D = N, M, D
        self.cells = int(self.L.stb_vcells(N, M))
        self.elems = int(self.L.stb_velems(N, M))
        self.stride = max(32, (self.elems + 31) // 32 * 32)
        self.dtype = dtype
        self.tables = torch.empty((D, self.stride), dtype=torch.float32 if dtype == "f32" else torch.float64, device=device)
        self.ws_bytes = int(self.L.stb_fill_workspace_bytes(N, M, D))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)

    def fill(self, a, stream=None, exact=False):
        """exact: the reference's own V recurrence, bit for bit (what tables below 512 rows get anyway)"""
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(a, dtype=np.float64)))
        f = self.L.stb_fill_Vf if self.dtype == "f32" else (self.L.stb_fill_V_exact if exact else self.L.stb_fill_V)
        check(f(dp(a), self.D, self.N, self.M, self.tables.data_ptr(), self.stride,
                self.ws.data_ptr(), self.ws_bytes, stream_ptr(stream)))

    def packed_host(self, d=0):
        t = self.tables[d].cpu().numpy()
        out = np.empty(self.cells, dtype=t.dtype)
        pos = 0
        for n in range(2, self.N + 1):
            ln = min(n - 1, self.M - 1)
            o = int(self.L.stb_vrowoff(n, self.M))
            out[pos:pos + ln] = t[o:o + ln]
            pos += ln
        return out


class DeviceGroups:
    """(n,t) pairs and per-restaurant totals resident in HBM (torch tensors)."""

    def __init__(self, g, device="cuda"):
        torch = _torch()
        self.g = g
        self.G = g.pairs
        self.I = g.I
        self.n = torch.as_tensor(g.n.view(np.int32), device=device)
        self.t = torch.as_tensor(g.t.view(np.int16), device=device)
        self.T = torch.as_tensor(g.T.view(np.int32), device=device)
        self.bpar = torch.as_tensor(g.bpar, device=device)


def sweep(tabs: DeviceTables, dg: DeviceGroups, stream=None):
    """out[d] = sum_{pairs, n>1} S_S_d(n,t)  (K3)"""
    torch = _torch()
    L = lib()
    wsb = int(L.stb_sweep_workspace_bytes(dg.G, tabs.D))
    ws = torch.empty(wsb, dtype=torch.uint8, device=tabs.tables.device)
    out = torch.empty(tabs.D, dtype=torch.float64, device=tabs.tables.device)
    check(L.stb_sweep_S(tabs.tables.data_ptr(), tabs.stride, tabs.S1.data_ptr(), tabs.N, tabs.D,
                        tabs.N, tabs.M, dg.n.data_ptr(), dg.t.data_ptr(), dg.G, out.data_ptr(),
                        ws.data_ptr(), wsb, stream_ptr(stream)))
    return out


def restaurant_terms(x, dg: DeviceGroups, stream=None):
    torch = _torch()
    L = lib()
    x = np.ascontiguousarray(np.atleast_1d(np.asarray(x, dtype=np.float64)))
    D = x.shape[0]
    wsb = int(L.stb_terms_workspace_bytes(dg.I, D))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dg.T.device)
    out = torch.empty(D, dtype=torch.float64, device=dg.T.device)
    check(L.stb_restaurant_terms(dp(x), D, dg.T.data_ptr(), dg.bpar.data_ptr(), dg.I, out.data_ptr(),
                                 ws.data_ptr(), wsb, stream_ptr(stream)))
    return out


def bterms(x, Q, shape, apar, dg: DeviceGroups, stream=None):
    torch = _torch()
    L = lib()
    x = np.ascontiguousarray(np.atleast_1d(np.asarray(x, dtype=np.float64)))
    J = x.shape[0]
    wsb = int(L.stb_terms_workspace_bytes(dg.I, J))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dg.T.device)
    out = torch.empty(J, dtype=torch.float64, device=dg.T.device)
    check(L.stb_bterms(dp(x), J, Q, shape, apar, dg.T.data_ptr(), dg.I, out.data_ptr(), ws.data_ptr(),
                       wsb, stream_ptr(stream)))
    return out

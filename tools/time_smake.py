#!/usr/bin/env python3
"""Wall time of the drop-in path: S_make / S_remake / S_free through the C API (host mirror included)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi
L = capi.lib()
L.S_make.restype = C.c_void_p
L.S_make.argtypes = [C.c_uint] * 4 + [C.c_double, C.c_uint32]
L.S_remake.restype = C.c_int; L.S_remake.argtypes = [C.c_void_p, C.c_double]
L.S_free.restype = None; L.S_free.argtypes = [C.c_void_p]
L.S_S.restype = C.c_double; L.S_S.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for flags, name in ((1, "S_STABLE"), (1 | 2, "S_STABLE|S_UVTABLE"), (1 | 4, "S_STABLE|S_FLOAT")):
    for rep in range(2):
        t0 = time.perf_counter(); sp = L.S_make(N, N, N, N, 0.5, flags); t1 = time.perf_counter()
        assert sp, capi.last_error()
        L.S_remake(sp, 0.6); t2 = time.perf_counter()
        v = L.S_S(sp, N, N // 2)
        L.S_free(sp); t3 = time.perf_counter()
        print(f"{name:22s} N=M={N}: S_make {1e3*(t1-t0):8.1f} ms  S_remake {1e3*(t2-t1):7.1f} ms  S_free {1e3*(t3-t2):6.1f} ms  S({N},{N//2})={v:.6f}", flush=True)

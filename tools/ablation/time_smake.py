#!/usr/bin/env python3
"""Wall time of the drop-in path: S_make / S_remake / S_free through the C API.  The host mirror is
copied on demand (128 rows per touched block); `sync` is what a full mirror costs on top."""
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
        v = L.S_S(sp, N, N // 2); t2b = time.perf_counter()
        for n in range(N // 4, N // 4 + 90): L.S_S(sp, n, 7)      # < 1 % of the rows
        t2c = time.perf_counter()
        L.stb_table_sync(sp); t2d = time.perf_counter()
        L.S_free(sp); t3 = time.perf_counter()
        print(f"{name:22s} N=M={N}: S_make {1e3*(t1-t0):8.1f} ms  S_remake {1e3*(t2-t1):7.2f} ms  first S_S {1e3*(t2b-t2):6.2f} ms  "
              f"90 more rows {1e3*(t2c-t2b):6.2f} ms  sync {1e3*(t2d-t2c):7.1f} ms  S_free {1e3*(t3-t2d):6.1f} ms  S({N},{N//2})={v:.6f}", flush=True)

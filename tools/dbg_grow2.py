import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc
from libstb_amd import capi
import test_gpu_table_api as T
G = os.path.join(ROOT, "tests", "golden")
T.test_config1_plumbing(G); T.test_flags_rejected_like_reference(); T.test_growth_trace_and_lazy_S1(G)
T.test_asympt_and_beyond_max(G); T.test_uv_accessors(G); T.test_uv_only_table_has_S1()
T.test_report_format(pathlib.Path(tempfile.mkdtemp())); T.test_threads_flag_growth_keeps_old_rows_alive()
T.test_float_storage_matches_reference(G)
O = orc.oracle()
for flags in (capi.S_STABLE | capi.S_FLOAT, capi.S_STABLE):
    t = capi.Table(20, 10, 300, 200, 0.4, flags)
    for (n, m) in ((30, 5), (120, 40), (299, 150)):
        got = t.S(n, m)
        N, M = t.usedN, t.usedM
        S1, tab = orc.fill_S(0.4, N, M)
        wrong = []; 
        for nn in range(3, N + 1):
            for mm in range(2, min(nn - 1, M) + 1):
                w = tab[orc.row_offset(nn, M) + mm - 2]; g = t.S(nn, mm)
                if abs(g - w) > 1e-5 * max(1, abs(w)): wrong.append((nn, mm, g, w))
        print("flags", flags, "probe", (n, m), "bounds", (N, M), "wrong", len(wrong), wrong[:3], wrong[-2:])
    t.free()

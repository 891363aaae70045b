"""One small workload per kernel, for rocprofv3 (kernel-trace statistics and PMC passes).

    rocprofv3 --kernel-trace --stats -d OUT -o NAME -- python3 tools/prof_target.py WHAT [reps]
    rocprofv3 --pmc WRITE_SIZE -d OUT -o NAME -- python3 tools/prof_target.py WHAT [reps]

The interpreter itself follows `--` (no env / shell hop: the profiler's library has initialised the
GPU by then).  WHAT:
    fill1 fill8 fill64   stb_fill_S of 1 / 8 / 64 tables, N = M = 10000 (k_fill_hb / k_fill_hb / k_fill_pc)
    fill1ck fill8ck      the same 1 / 8 tables in the checkpointed form (k_fill_ck; needs the ablation library)
    fill1chain fill8chain  ... and in the chain form (what round 2 ran)
    ffill8               8 tables stored as floats (S_FLOAT written once)  (k_fill_hb<4,0,1>)
    vfill vfillx         stb_fill_V, N = M = 10000: from the S recurrence's cells (k_fill_hb<2,0,2>) / the reference's own
                         V recurrence, bit for bit (k_fillv_chain)
    grid64 grid8         the fused 64- / 8-discount aterms over 10^6 pairs, n < 10000 (k_grid_hb: walking waves sum /
                         k_fill_hb<4,1>: tile workers sum)
    grid64chain grid8chain  the same grids in the chain form (what rounds 2-3 ran at 64 discounts)
    eval1f               one-discount aterms on a set that is used again and again (fused, as samplea's kept set)
    sweep64              the same grid through stored tables             (k_fill_pc, k_sweep_partial)
    eval1                one-discount aterms, 10^6 pairs, n < 4000       (k_fill_chain, k_sweep_partial, k_terms_partial)
    bterms               stb_bterms, 10^6 restaurants x 20 abscissae     (k_terms_partial)
    fresh_samplea        samplea on pairs that change between calls (10^6 pairs, n < 4000): upload, cell lists from the count
                         slab (k_count_cells, k_item_count, k_emit_cells), 8 fused evaluations (k_fill_hb<2,1,0>)
    fresh_grid64 fresh_grid8   stb_groups_update_pairs + the first 64- / 8-discount evaluation on the new pairs, n < 10000
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from libstb_amd import capi, synth

what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = capi.lib()
N = 10000


def groups_handle(g, Dmax):
    M = max(int(g.t.max()) + 1, 10)
    Ng = max(int(g.n.max()) + 1, M)
    h = L.stb_groups_create(g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p),
                            g.n.ctypes.data_as(capi.c_u32_p), g.t.ctypes.data_as(capi.c_u16_p), capi.dp(g.bpar), Ng, M, Dmax)
    assert h, capi.last_error()
    return h


if what in ("fill1", "fill8", "fill64", "fill1chain", "fill8chain", "fill1ck", "fill8ck"):
    D = int(what[4:].replace("chain", "").replace("ck", ""))
    if what.endswith("chain"):
        os.environ["STB_CK"] = "0"
        os.environ["STB_HB"] = "0"
    if what.endswith("ck"):
        os.environ["STB_HB"] = "0"
        os.environ["STB_CK"] = "1"
    a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
    T = capi.DeviceTables(N, N, D=D)
    for _ in range(reps):
        T.fill(a)
    torch.cuda.synchronize()
    T.status()
elif what in ("vfill", "vfillx"):
    T = capi.DeviceVTables(N, N, D=1)
    for _ in range(reps):
        T.fill(np.array([0.5]), exact=(what == "vfillx"))
    torch.cuda.synchronize()
    capi.check(L.stb_fill_status())
elif what == "ffill8":
    T = capi.DeviceFloatTables(N, N, D=8)
    a = synth.discount_grid(64)[:8]
    for _ in range(reps):
        T.fill(a)
    torch.cuda.synchronize()
    T.status()
elif what in ("grid64", "grid8", "grid8chain", "grid64chain", "sweep64"):
    if what == "sweep64":
        os.environ["STB_ATERMS_FUSED"] = "0"
    if what.endswith("chain"):
        os.environ["STB_ATERMS_HB"] = "0"
        os.environ["STB_ATERMS_GRID"] = "0"
    Dg = 8 if what.startswith("grid8") else 64
    g = synth.groups(1000, 1000, N, "wide")
    h = groups_handle(g, Dg)
    x = np.ascontiguousarray(synth.discount_grid(64)[:Dg])
    out = np.zeros(Dg)
    for _ in range(reps):
        capi.check(L.stb_groups_aterms(h, capi.dp(x), Dg, capi.dp(out)))
    L.stb_groups_free(h)
elif what in ("eval1", "eval1f"):
    g = synth.groups(1000, 1000, 4000, "wide")
    h = groups_handle(g, 1)
    x, out = np.array([0.45]), np.zeros(1)
    if what == "eval1f":  # (a set whose restaurants have been refreshed is a kept set: its evaluations are fused)
        capi.check(L.stb_groups_update_restaurants(h, g.T.ctypes.data_as(capi.c_u32_p), capi.dp(g.bpar)))
    for _ in range(reps):
        capi.check(L.stb_groups_aterms(h, capi.dp(x), 1, capi.dp(out)))
    L.stb_groups_free(h)
elif what == "bterms":
    g = synth.groups(1000000, 1, 4000, "realistic")
    dg = capi.DeviceGroups(g)
    x = np.linspace(1.0, 100.0, 20)
    for _ in range(reps):
        capi.bterms(x, 0.05, g.shape, 0.5, dg)
    torch.cuda.synchronize()
elif what == "fresh_samplea":
    import ctypes as C
    import orc
    g = synth.groups(1000, 1000, 4000, "wide")
    NP = C.POINTER(C.c_uint32) * g.I
    TP = C.POINTER(C.c_uint16) * g.I
    nn, tt = NP(), TP()
    off = 0
    for i in range(g.I):
        nn[i] = C.cast(g.n.ctypes.data + 4 * off, C.POINTER(C.c_uint32))
        tt[i] = C.cast(g.t.ctypes.data + 2 * off, C.POINTER(C.c_uint16))
        off += int(g.K[i])
    for r in range(reps + 2):
        g.n[12345 + 977 * r] += 1
        orc.seed_libc(777, 12345)
        L.samplea(0.5, g.I, g.K.ctypes.data_as(capi.c_int_p), g.T.ctypes.data_as(capi.c_u32_p), nn, tt, None, capi.dp(g.bpar), None, 1, 0)
    L.stb_sampler_cache_clear()
elif what in ("fresh_grid64", "fresh_grid8"):
    Dg = 8 if what.endswith("8") else 64
    g = synth.groups(1000, 1000, N, "wide")
    h = groups_handle(g, Dg)
    x = np.ascontiguousarray(synth.discount_grid(64)[:Dg])
    out = np.zeros(Dg)
    capi.check(L.stb_groups_aterms(h, capi.dp(x), Dg, capi.dp(out)))
    for r in range(reps + 1):
        g.n[12345 + 977 * r] += 1
        capi.check(L.stb_groups_update_pairs(h, g.n.ctypes.data_as(capi.c_u32_p), g.t.ctypes.data_as(capi.c_u16_p)))
        capi.check(L.stb_groups_aterms(h, capi.dp(x), Dg, capi.dp(out)))
    L.stb_groups_free(h)
else:
    raise SystemExit("unknown workload " + what)
print("done", what, reps)

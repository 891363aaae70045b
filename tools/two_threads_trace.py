"""two host threads, one group set each, evaluating 2-discount grids over and over (tests/test_gpu_threads.py): do their
table walks overlap on the GPU?  Run under rocprofv3 --kernel-trace and look at the overlap of k_fill_hb launches:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o tt -- python3 tools/two_threads_trace.py
    python3 tools/two_threads_trace.py OUT/.../tt_kernel_trace.csv      (analysis)"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fill_hb" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]          # the threaded half
    ov = tot = 0
    for a, b in zip(rows, rows[1:]):
        tot += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
        ov += max(0, int(a["End_Timestamp"]) - int(b["Start_Timestamp"]))
    print(f"{len(rows)} walks in the threaded half: {tot / 1e6:.2f} ms of kernel time, {ov / 1e6:.2f} ms of it beside the next walk ({100.0 * ov / max(tot, 1):.0f} %)")
    sys.exit(0)
import numpy as np
import orc
from libstb_amd import capi, synth
L = capi.lib()
N = M = 3000
sets, xs, outs = [], [], []
for k in range(2):
    g = synth.groups(200, 1000, N, "wide", seed=synth.SEED + k)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, 2)
    sets.append(h); xs.append(np.array([0.31 + 0.2 * k, 0.62 + 0.1 * k])); outs.append(np.zeros(2))
def work(k, reps=40):
    for _ in range(reps):
        capi.check(L.stb_groups_aterms(sets[k], capi.dp(xs[k]), 2, capi.dp(outs[k])))
work(0); work(1)
th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
for t in th: t.start()
for t in th: t.join()

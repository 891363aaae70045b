import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstb_amd import capi
N = int(sys.argv[1]); a = float(sys.argv[2])
A = capi.DeviceTables(N, N, D=1); A.fill([a], capi.FILL_FUSED)
B = capi.DeviceTables(N, N, D=1); B.fill([a], capi.FILL_SPLIT)
ta = A.tables[0].cpu().numpy(); tb = B.tables[0].cpu().numpy()
first = None
for n in range(3, N + 1):
    o = A.rowoff(n); ra = ta[o:o+n-2]; rb = tb[o:o+n-2]
    bad = np.nonzero(np.abs(ra - rb) > 1e-9 * np.maximum(1, np.abs(ra)))[0]
    if len(bad):
        print("row", n, "first bad cols", bad[:5] + 2, "count", len(bad), "fused", ra[bad[:3]], "split", rb[bad[:3]])
        if first is None: first = n
        if n > first + 3: break
print("S1 max diff", np.max(np.abs(A.S1[0].cpu().numpy() - B.S1[0].cpu().numpy())))

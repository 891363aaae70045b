"""Timeline of one fused grid aterms in the checkpointed form: python tools/timeline_grid.py D out.txt   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import orc
from libstb_amd import capi, synth
L = capi.lib()
D = int(sys.argv[1]); out = sys.argv[2]
g = synth.groups(1000, 1000, 10000, "wide")
M = max(int(g.t.max()) + 1, 10); N = max(int(g.n.max()) + 1, M)
x = np.ascontiguousarray(synth.discount_grid(64)[:D]); o = np.zeros(D)
h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), N, M, D)
for _ in range(3):
    capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(o)))
os.environ["STB_CK_TIMELINE"] = out + ".raw"
capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(o)))
del os.environ["STB_CK_TIMELINE"]
L.stb_groups_free(h)
print("raw timeline in", out + ".raw")

import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc
from libstb_amd import capi, synth
libc = C.CDLL(None); libc.rand.restype = C.c_int
L = capi.lib()
g = synth.groups(3, 20, 30, "wide")
def seq(k):
    libc.srand(1); return [libc.rand() for _ in range(k)]
base = seq(6)
for trial in range(3):
    libc.srand(1)
    h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), 40, 40, 1)
    r1 = libc.rand()
    x = np.array([0.4]); out = np.zeros(1)
    L.stb_groups_aterms(h, capi.dp(x), 1, capi.dp(out))
    r2 = libc.rand()
    L.stb_groups_free(h)
    r3 = libc.rand()
    print("trial", trial, "expected", base[:3], "got", r1, r2, r3)
libc.srand48(5); libc.drand48.restype = C.c_double
d0 = libc.drand48(); libc.srand48(5)
h = L.stb_groups_create(g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), 40, 40, 1)
L.stb_groups_free(h)
print("drand48 stream intact:", libc.drand48() == d0)

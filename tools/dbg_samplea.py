import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from libstb_amd import capi, synth
L = capi.lib(); R = orc.ref()
rng = np.random.default_rng(1)
I, K = 3, 20
n = rng.integers(0, 30, size=I * K).astype(np.uint32)
t = np.minimum(n, rng.integers(1, 4, size=I * K)).astype(np.uint16)
g = synth.Groups(I=I, K=np.full(I, K, dtype=np.int32), n=n, t=t,
                 T=t.reshape(I, K).sum(1).astype(np.uint32), N=n.reshape(I, K).sum(1).astype(np.uint32),
                 bpar=np.full(I, 10.0))
print("n", n[:20], "t", t[:20], "T", g.T)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_samplers import ragged, trace
nn, tt = ragged(g)
orc.seed_libc(1, 424242)
a_ref = R.ref_samplea_flat(0.5, g.I, orc.i32p(g.K), orc.u32p(g.T), orc.u32p(g.n), orc.u16p(g.t), orc.dp(g.bpar), 1, 0)
rx = [R.ref_trace_x(i) for i in range(R.ref_trace_count())]; ry = [R.ref_trace_y(i) for i in range(R.ref_trace_count())]
orc.seed_libc(1, 424242)
a_amd = L.samplea(0.5, g.I, orc.i32p(g.K), orc.u32p(g.T), nn, tt, None, orc.dp(g.bpar), None, 1, 0)
xs, ys, code = trace(L)
print("a_ref", a_ref, "a_amd", a_amd)
for i in range(max(len(rx), len(xs))):
    print(i, rx[i] if i < len(rx) else None, ry[i] if i < len(ry) else None, "|", xs[i] if i < len(xs) else None, ys[i] if i < len(ys) else None)

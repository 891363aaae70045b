"""One of several processes that share ONE GPU (tests/test_gpu_shared.py): evaluates 16 discounts x 10^6 pairs at
N = 10^4 a number of times, all processes in step, and reports per-step wall times, the result's bytes, the library's
count of slow launches and whether it switched to the forms without waits.
usage: python tests/shared_worker.py rank world steps out.json   (env: STB_SHARED_GPU, rendezvous through files in dirname(out))"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

rank, world, steps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
D = int(os.environ.get("SHARED_WORKER_D", "16"))
from libstb_amd import capi, synth  # noqa: E402


def barrier(tag):
    """every process of the test passes here together (files: no torch.distributed, no second GPU context)"""
    d = os.path.dirname(out)
    open(os.path.join(d, f"{tag}.{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, f"{tag}.{r}")) for r in range(world)):
        if time.time() - t0 > 300:
            raise SystemExit("barrier timeout")
        time.sleep(0.0005)


L = capi.lib()
g = synth.groups(1000, 1000, 10000, "wide")
N, M = int(g.n.max()), int(min(g.n.max(), g.t.max()))
import ctypes as C  # noqa: E402

u32p, u16p, i32p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), C.POINTER(C.c_int32)
h = L.stb_groups_create(g.I, g.K.ctypes.data_as(i32p), g.T.ctypes.data_as(u32p), g.n.ctypes.data_as(u32p), g.t.ctypes.data_as(u16p),
                        capi.dp(g.bpar), N, M, D)
assert h, capi.last_error()
x = np.ascontiguousarray(synth.discount_grid(64)[rank * D % 64: rank * D % 64 + D])
res = np.zeros(D)
capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(res)))  # set-up
first = res.copy()
ms, same = [], True
barrier("go")
for s in range(steps):
    t0 = time.perf_counter()
    capi.check(L.stb_groups_aterms(h, capi.dp(x), D, capi.dp(res)))
    ms.append((time.perf_counter() - t0) * 1e3)
    if s == 0:
        ref = res.copy()
    same = same and bool((res == ref).all())
barrier("done")
json.dump({"rank": rank, "ms": ms, "same_bits_every_step": same, "result_hex": res.tobytes().hex(), "first_hex": first.tobytes().hex(),
           "slow_launches": int(L.stb_slow_launches()), "shared_mode": int(L.stb_shared_gpu_mode()),
           "fallbacks": int(L.stb_groups_fallbacks()), "x": x.tolist()}, open(out, "w"))
L.stb_groups_free(h)

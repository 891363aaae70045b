import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc
from libstb_amd import capi
O = orc.oracle()
bad = 0
for rep in range(6):
    for flags in (capi.S_STABLE, capi.S_STABLE | capi.S_FLOAT, capi.S_STABLE | capi.S_UVTABLE):
        t = capi.Table(20, 10, 300, 200, 0.4, flags)
        for (n, m) in ((30, 5), (120, 40), (299, 150)):
            got = t.S(n, m)
            N, M = t.usedN, t.usedM
            S1, tab = orc.fill_S(0.4, N, M)
            want = O.orc_S_S(orc.dp(tab), orc.dp(S1), N, M, n, m)
            ok = abs(got - want) <= 1e-6 * max(1, abs(want))
            if not ok:
                bad += 1
                # scan the whole mirror through S_S to see how much is wrong
                wrong = 0; first = None
                for nn in range(3, N + 1):
                    for mm in range(2, min(nn - 1, M) + 1):
                        w = tab[orc.row_offset(nn, M) + mm - 2]
                        g = t.S(nn, mm)
                        if abs(g - w) > 1e-6 * max(1, abs(w)):
                            wrong += 1
                            if first is None: first = (nn, mm, g, w)
                print("rep", rep, "flags", flags, "probe", (n, m), "got", got, "want", want, "bounds", (N, M), "wrong cells", wrong, "first", first)
        t.free()
print("bad", bad)

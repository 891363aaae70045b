"""diagnostic: where the producer wave of every strip of a chain-form fill ran (needs `make -C libstb_amd/csrc stamp`).
Prints, per XCD / compute unit / SIMD, how many producers were placed there, how many producers were alive together
on one SIMD, and the time a strip's producer took against the strip's length.
usage: python tools/census_chain.py [N] [D]        (run from the repo root; STB_CHAIN_* apply)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstb_amd import capi, synth
capi.LIB_PATH = os.environ.get("STB_LIB_PATH") or capi.LIB_PATH.replace("libstb_amd.so", "libstb_amd_stamp.so")
import numpy as np, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = capi.DeviceTables(N, N, D=D)
a = synth.discount_grid(64)[:D] if D > 1 else np.array([0.5])
T.fill(a, capi.FILL_CHAIN); torch.cuda.synchronize()
os.environ["STB_TIMELINE_FILE"] = "gpurun_out/timeline_scratch.txt"
os.environ["STB_CENSUS_FILE"] = "gpurun_out/census_chain.txt"
T.fill(a, capi.FILL_CHAIN); torch.cuda.synchronize()
T.status()
r = np.loadtxt("gpurun_out/census_chain.txt", dtype=np.int64, ndmin=2)
tick, strip, table, hw, xcc, t0, t1, cyc = r.T
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
unit = xcc * 1000 + se * 100 + sh * 16 + cu          # one compute unit
print(f"{len(r)} producers recorded (tickets < 1024); {len(set(unit))} compute units used; XCDs {sorted(set(xcc))}")
print("producers per XCD:", dict(sorted(collections.Counter(xcc.tolist()).items())))
print("producers per SIMD id:", dict(sorted(collections.Counter(simd.tolist()).items())))
per_unit = collections.Counter(unit.tolist())
print("compute units by number of producers hosted:", dict(sorted(collections.Counter(per_unit.values()).items())))
# overlap in time on one SIMD
slot = unit * 4 + simd
worst = collections.Counter()
for s_ in set(slot.tolist()):
    idx = np.where(slot == s_)[0]
    ev = sorted([(t0[i], 1) for i in idx] + [(t1[i], -1) for i in idx])
    live = peak = 0
    for _, dlt in ev:
        live += dlt; peak = max(peak, live)
    worst[peak] += 1
print("SIMDs by peak number of producers alive together:", dict(sorted(worst.items())))
# same-XCD neighbours
key = {(int(tb), int(st)): int(x) for tb, st, x in zip(table, strip, xcc)}
same = sum(1 for (tb, st), x in key.items() if (tb, st + 1) in key and key[(tb, st + 1)] == x)
pairs = sum(1 for (tb, st) in key if (tb, st + 1) in key)
print(f"neighbouring strips on the same XCD: {same} of {pairs}")
dur = (t1 - t0) / 100.0
start = (t0 - t0.min()) / 100.0
for tb in sorted(set(table.tolist()))[:2]:
    m = table == tb
    o = np.argsort(strip[m])
    print(f"table {tb}: strip: start us / producer time us:", " ".join(f"{int(s_)}:{a_:.0f}/{b_:.0f}" for s_, a_, b_ in list(zip(strip[m][o], start[m][o], dur[m][o]))[::max(1, m.sum() // 10)]))
long_ = (t1 - t0) > 0.5 * (t1 - t0).max()
mhz = cyc[long_] / ((t1 - t0)[long_] / 100.0)
print(f"shader clock seen by the long-lived producers (s_memtime cycles per wall-clock us): median {np.median(mhz):.0f} MHz, min {mhz.min():.0f}, max {mhz.max():.0f}")
print(f"first producer start -> last producer end: {(t1.max() - t0.min()) / 100.0:.1f} us")

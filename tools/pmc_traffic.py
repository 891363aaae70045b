#!/usr/bin/env python3
"""HBM traffic of the fill kernels from rocprofv3 PMC passes (MI355X_MICROARCH.md, section HBM):
WRITE_SIZE and FETCH_SIZE are collected in SEPARATE passes (TCC slots), both are in KiB, and on
gfx950 FETCH_SIZE counts wide coalesced reads at half their size -> doubled here.

usage: pmc_traffic.py <label> <write_pass_dir> <fetch_pass_dir> <fills_in_run> [out.json]
Appends/updates profiles/r02_hbm_traffic.json: per kernel, bytes per launch and per fill.
"""
import collections, csv, glob, json, os, sys

label, wdir, fdir, fills = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_hbm_traffic.json")


def load(d, counter):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"]) * 1024.0
        cnt[k] += 1
    return agg, cnt


w, wc = load(wdir, "WRITE_SIZE")
f, fc = load(fdir, "FETCH_SIZE")
res = {}
for k in sorted(set(w) | set(f)):
    if not k.startswith("k_"):
        continue
    launches = max(wc.get(k, 0), fc.get(k, 0))
    wb, fb = w.get(k, 0.0), 2.0 * f.get(k, 0.0)
    res[k] = {"launches_per_fill": launches / fills, "write_bytes_per_fill": wb / fills,
              "fetch_bytes_per_fill_x2": fb / fills, "hbm_bytes_per_fill": (wb + fb) / fills,
              "hbm_bytes_per_launch": (wb + fb) / launches if launches else None}
db = json.load(open(out)) if os.path.exists(out) else {}
db[label] = res
json.dump(db, open(out, "w"), indent=1)
print(json.dumps({label: res}, indent=1))

#!/usr/bin/env python3
"""Print the kernel timeline of a rocprofv3 --kernel-trace csv: start (us from the first), duration, name; gaps marked.
  python tools/trace_timeline.py trace.csv [first_us last_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e30
prev = None
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    if s < lo or s > hi: continue
    gap = f"(+{s - prev:7.1f})" if prev is not None else " " * 10
    name = r["Kernel_Name"]
    if "rocprim" in name:
        import re
        m = re.search(r"wrapped_(\w+?)_config|detail::(\w+)<", name)
        name = "rocprim:" + (m.group(1) or m.group(2) if m else "?")
    print(f"{s:10.1f} {gap} {e - s:8.1f}  {name[:80]}")
    prev = e

#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per-kernel durations and inter-kernel gaps of the last fill."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
names = sorted(set(r['Kernel_Name'][:40] for r in rows))
for nm in names:
    rs = [r for r in rows if r['Kernel_Name'].startswith(nm)]
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rs]
    print(f"{nm:42s} calls {len(d):5d} avg {sum(d)/len(d):8.2f} us min {min(d):7.2f} max {max(d):7.2f} total {sum(d)/1e3:8.3f} ms")
main = [r for r in rows if ('k_rec' in r['Kernel_Name'] or 'k_fill' in r['Kernel_Name'])]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 157
last = main[-per:]
t0 = int(last[0]['Start_Timestamp'])
t1 = max(int(r['End_Timestamp']) for r in rows if int(r['Start_Timestamp']) >= t0)
print('span of last fill (us):', (t1 - t0) / 1e3)
gaps = [(int(last[i + 1]['Start_Timestamp']) - int(last[i]['End_Timestamp'])) / 1e3 for i in range(len(last) - 1)]
print('gaps between main kernels us: mean %.2f max %.2f' % (sum(gaps) / len(gaps), max(gaps)))

set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r04; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fill64 -o fill64 -- python3 $ROOT/tools/prof_target.py fill64 5 > /dev/null 2>&1
f64=$(find $OUT/fill64 -name "*kernel_trace.csv" | tail -1)
python3 - "$f64" > $OUT/r04_fill64_span.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fill_pc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // 5
fills = [rows[i * per:(i + 1) * per] for i in range(5)]
print("# k_fill_pc launches of the 64-table fill (two sub-batches on two streams), from the rocprofv3 kernel trace:")
print("# fill, launches, span first start -> last end (ms), sum of launch durations (ms), streams")
for i, f in enumerate(fills):
    s0 = min(int(r["Start_Timestamp"]) for r in f); e1 = max(int(r["End_Timestamp"]) for r in f)
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in f)
    print(f"{i} {len(f)} {(e1 - s0) / 1e6:.3f} {tot / 1e6:.3f} {len(set(r.get('Stream_Id', r.get('Queue_Id', '?')) for r in f))}")
PY
cat $OUT/r04_fill64_span.txt
find $OUT -name "*kernel_trace.csv" -o -name "*.db" | xargs -r rm -f

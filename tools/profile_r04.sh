#!/bin/bash
# Round-4 profiles: rocprofv3 kernel statistics of the bench command AND of every workload a duration is quoted for
# (one per target, so that no average mixes sizes), per-kernel PMC traffic (WRITE_SIZE / FETCH_SIZE in separate
# passes), SQ counters of the kernels the roofline lines price, and the kernel trace of the 64-table fill (two
# sub-batches on two streams: the span of a fill, not the sum of its launches).
# Run on the GPU box from the repo root:  bash tools/profile_r04.sh
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, then the python3 command line (the interpreter itself follows `--`)
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 "$@" > $OUT/$name.stdout 2> $OUT/$name.stderr
  local f=$(find $OUT/$name -name "*kernel_stats.csv" 2>/dev/null | tail -1)
  [ -n "$f" ] && cp $f $OUT/r04_${name}_kernel_stats.csv
  echo "stats $name: $(head -3 $OUT/r04_${name}_kernel_stats.csv 2>/dev/null | tail -2 | cut -c1-170)"
}
# the timed region of the contract line alone, then the 8-per-GPU share
stats bench_n1_main $ROOT/bench.py --steps 10 --no-cpu-baseline --no-extra --no-batch64
tail -1 $OUT/bench_n1_main.stdout > $OUT/r04_bench_n1_main_under_rocprof.json
stats bench_d8 $ROOT/bench.py --steps 10 --discounts-per-gpu 8 --no-cpu-baseline --no-extra --no-batch64
tail -1 $OUT/bench_d8.stdout > $OUT/r04_bench_d8_under_rocprof.json
# one workload per quoted duration
for w in fill1 fill8 ffill8 fill64 vfill vfillx grid8 grid8chain grid64 grid64chain eval1 eval1f bterms; do stats $w $ROOT/tools/prof_target.py $w 5; done
# the 64-table fill launch by launch: two streams side by side (keep the trace: the span is not in the statistics)
f64=$(find $OUT/fill64 -name "*kernel_trace.csv" | tail -1)
[ -n "$f64" ] && python3 - "$f64" > $OUT/r04_fill64_span.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fill_pc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# five fills of equally many launches, one after the other (a fill's two sub-batches overlap each other, not the next fill)
per = len(rows) // 5
fills = [rows[i * per:(i + 1) * per] for i in range(5)]
print("# k_fill_pc launches of the 64-table fill (two sub-batches on two streams), from the rocprofv3 kernel trace:")
print("# fill, launches, span first start -> last end (ms), sum of launch durations (ms), streams")
for i, f in enumerate(fills):
    s0 = min(int(r["Start_Timestamp"]) for r in f); e1 = max(int(r["End_Timestamp"]) for r in f)
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in f)
    print(f"{i} {len(f)} {(e1 - s0) / 1e6:.3f} {tot / 1e6:.3f} {len(set(r.get('Stream_Id', r.get('Queue_Id', '?')) for r in f))}")
PY
# PMC: WRITE_SIZE and FETCH_SIZE in separate passes (TCC slots), no tracing domains beside them
for w in fill1 fill8 ffill8 vfill grid8 grid64; do
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -o p -- python3 $ROOT/tools/prof_target.py $w 3 > /dev/null 2> $OUT/pmc_${w}_$c.stderr || echo "pmc $w $c failed"
  done
  echo "pmc $w done"
done
find $OUT -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*domain_stats.csv" | grep -v pmc_ | xargs -r rm -f
find $OUT -name "*.db" | xargs -r rm -f
cd $ROOT
lab() { case $1 in fill1) echo N10000_M10000_D1_hb;; fill8) echo N10000_M10000_D8_hb;; ffill8) echo N10000_M10000_D8_hb_float;; grid8) echo grid_N10000_D8;; vfill) echo vfill_N10000_D1;; grid64) echo grid_N10000_D64;; esac; }
rm -f $OUT/r04_hbm_traffic.json
for w in fill1 fill8 ffill8 vfill grid8 grid64; do
  python3 tools/pmc_traffic.py $(lab $w) $OUT/pmc_${w}_WRITE_SIZE $OUT/pmc_${w}_FETCH_SIZE 3 $OUT/r04_hbm_traffic.json > /dev/null || echo "traffic $w failed"
done
# SQ counters of the fills and of the fused grids
for w in fill1 fill8 grid8 grid64; do bash tools/pmc_sq.sh $w > /dev/null 2>&1; cp gpurun_out/pmc_sq/$w.txt $OUT/r04_sq_counters_$w.txt 2>/dev/null; done
ls $OUT/r04_* | head -60; du -sh $OUT

#!/bin/bash
# Round-6 profiles: rocprofv3 kernel statistics of the bench command's timed region and of every workload a duration
# is quoted for (one per target, so that no average mixes sizes), the kernel timelines of the fresh-pairs paths, HBM
# traffic (WRITE_SIZE / FETCH_SIZE in separate --pmc passes) and SQ counters of the kernels the roofline lines price.
# Run on the GPU box from the repo root:  bash tools/profile_r06.sh
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, then the python3 command line (the interpreter itself follows `--`)
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 "$@" > $OUT/$name.stdout 2> $OUT/$name.stderr
  local f=$(find $OUT/$name -name "*kernel_stats.csv" 2>/dev/null | tail -1)
  [ -n "$f" ] && cp $f $OUT/r06_${name}_kernel_stats.csv
  echo "stats $name: $(head -3 $OUT/r06_${name}_kernel_stats.csv 2>/dev/null | tail -2 | cut -c1-170)"
}
stats bench_n1_main $ROOT/bench.py --steps 10 --no-cpu-baseline --no-extra --no-batch64
tail -1 $OUT/bench_n1_main.stdout > $OUT/r06_bench_n1_main_under_rocprof.json
for w in fill1 fill8 fill64 vfill ffill8 grid64 grid8 eval1f fresh_samplea fresh_grid64 fresh_grid8; do stats $w $ROOT/tools/prof_target.py $w 5; done
# the fresh-pairs paths kernel by kernel (one call each, from the traces)
for w in fresh_samplea fresh_grid64; do
  t=$(find $OUT/$w -name "*kernel_trace.csv" | tail -1)
  [ -n "$t" ] && python3 $ROOT/tools/trace_timeline.py $t > $OUT/r06_${w}_timeline_all.txt
done
python3 - $OUT <<'PY'
import sys, re
out = sys.argv[1]
for w, first in (("fresh_samplea", "k_count_cells"), ("fresh_grid64", "k_count_cells")):
    try:
        lines = open(f"{out}/r06_{w}_timeline_all.txt").read().splitlines()
    except OSError:
        continue
    idx = [i for i, l in enumerate(lines) if first in l]
    if len(idx) < 3:
        continue
    a, b = idx[-2], idx[-1]          # the last complete call: from its count kernel to the next one's
    a0 = max(a - 6, 0)
    with open(f"{out}/r06_{w}_timeline.txt", "w") as f:
        f.write(f"# one call of {w} (tools/prof_target.py), kernel by kernel: start us, (gap to the previous kernel's end), duration us, kernel\n")
        f.write("\n".join(lines[a0:b - 3]) + "\n")
PY
# PMC: WRITE_SIZE and FETCH_SIZE in separate passes, no tracing domains beside them
for w in fill1 fill8 grid64 fresh_grid64; do
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -o p -- python3 $ROOT/tools/prof_target.py $w 3 > /dev/null 2> $OUT/pmc_${w}_$c.stderr || echo "pmc $w $c failed"
  done
  echo "pmc $w done"
done
find $OUT -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*domain_stats.csv" | grep -v pmc_ | xargs -r rm -f
find $OUT -name "*.db" | xargs -r rm -f
cd $ROOT
lab() { case $1 in fill1) echo N10000_M10000_D1_hb;; fill8) echo N10000_M10000_D8_hb;; grid64) echo grid_N10000_D64;; fresh_grid64) echo fresh_grid_N10000_D64;; esac; }
reps() { case $1 in fresh_grid64) echo 4;; *) echo 3;; esac; }
rm -f $OUT/r06_hbm_traffic.json
for w in fill1 fill8 grid64 fresh_grid64; do
  python3 tools/pmc_traffic.py $(lab $w) $OUT/pmc_${w}_WRITE_SIZE $OUT/pmc_${w}_FETCH_SIZE $(reps $w) $OUT/r06_hbm_traffic.json > /dev/null || echo "traffic $w failed"
done
for w in fill1 grid64; do bash tools/pmc_sq.sh $w > /dev/null 2>&1; cp gpurun_out/pmc_sq/$w.txt $OUT/r06_sq_counters_$w.txt 2>/dev/null; done
ls $OUT/r06_* | head -60; du -sh $OUT

#!/bin/bash
# Round-2 profiles: rocprofv3 kernel statistics of the bench command and per-kernel PMC traffic.
# Run on the GPU box from the repo root:  bash tools/profile_r02.sh   (writes gpurun_out/prof_r02/)
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, then the python3 command line
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 "$@" > $OUT/$name.stdout 2> $OUT/$name.stderr
  local f=$(find $OUT/$name -name "*kernel_stats.csv" 2>/dev/null | tail -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
  echo "stats $name: $(head -3 $OUT/${name}_kernel_stats.csv 2>/dev/null | tail -2 | cut -c1-150)"
}
stats bench_n1 $ROOT/bench.py --steps 10 --no-cpu-baseline
tail -1 $OUT/bench_n1.stdout > $OUT/bench_n1_under_rocprof.json
# the timed region alone (no extras): the roofline kernel's average here is what bench.py's events give
stats bench_n1_main $ROOT/bench.py --steps 10 --no-cpu-baseline --no-extra --no-batch64
tail -1 $OUT/bench_n1_main.stdout > $OUT/bench_n1_main_under_rocprof.json
stats bench_d8 $ROOT/bench.py --steps 10 --discounts-per-gpu 8 --no-cpu-baseline --no-extra --no-batch64
tail -1 $OUT/bench_d8.stdout > $OUT/bench_d8_under_rocprof.json
for w in vfill eval1 bterms; do stats $w $ROOT/tools/prof_target.py $w 5; done
# PMC: WRITE_SIZE and FETCH_SIZE in separate passes (TCC slots), no tracing domains beside them
for w in fill1 fill8 fill64 vfill grid64 sweep64 eval1; do
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -o p -- python3 $ROOT/tools/prof_target.py $w 3 > /dev/null 2> $OUT/pmc_${w}_$c.stderr || echo "pmc $w $c failed"
  done
  echo "pmc $w done"
done
# keep the summaries, drop the raw traces (the per-launch counter csv of the PMC passes is small)
find $OUT -name "*kernel_trace.csv" -o -name "*.db" -o -name "*agent_info.csv" -o -name "*domain_stats.csv" | grep -v pmc_ | xargs -r rm -f
find $OUT -name "*.db" | xargs -r rm -f
cd $ROOT
lab() { case $1 in fill1) echo N10000_M10000_D1_chain;; fill8) echo N10000_M10000_D8_chain;; fill64) echo N10000_M10000_D64_pc;; vfill) echo vfill_N10000_D1;; grid64) echo grid_N10000_D64;; sweep64) echo sweep_N10000_D64;; eval1) echo eval_N4000_D1;; esac; }
for w in fill1 fill8 fill64 vfill grid64 sweep64 eval1; do
  python3 tools/pmc_traffic.py $(lab $w) $OUT/pmc_${w}_WRITE_SIZE $OUT/pmc_${w}_FETCH_SIZE 3 $OUT/r02_hbm_traffic.json > /dev/null || echo "traffic $w failed"
done
du -sh $OUT; cat $OUT/r02_hbm_traffic.json | head -5

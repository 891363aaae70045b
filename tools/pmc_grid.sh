#!/bin/bash
# diagnostic: SQ counters of the grid kernel at 64 discounts in three modes (everything / staged but no look-ups /
# bare walk), two --pmc passes each.   bash tools/pmc_grid.sh   -> gpurun_out/pmc_grid/<mode>.txt
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_grid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 0 4 12; do
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL"; do
    i=$((i+1))
    STB_GRID_DIAG=$mode timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/m${mode}_g$i -o p -- python3 $ROOT/tools/prof_target.py grid64 3 > /dev/null 2> $OUT/m${mode}_g$i.stderr || echo "mode $mode group $i failed"
  done
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for mode in (0, 4, 12):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/m{mode}_g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_grid_hb" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(f"{out}/mode{mode}.txt", "w") as fo:
        print(f"k_grid_hb<4, 12>, 64 discounts x 10^6 pairs, N = M = 10^4, STB_GRID_DIAG={mode}", file=fo)
        for c, v in sorted(acc.items()):
            print(f"  {c:28s} {sum(v)/len(v):16.0f}  (avg of {len(v)} launches)", file=fo)
    print(open(f"{out}/mode{mode}.txt").read())
PY
find $OUT -name "*.db" | xargs -r rm -f

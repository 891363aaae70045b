#!/bin/bash
# diagnostic: SQ / LDS counters of one workload of tools/prof_target.py (default fill1), one --pmc pass per group.
# Run on the GPU box from the repo root:  bash tools/pmc_sq.sh [fill1|fill8|...]   -> gpurun_out/pmc_sq/<workload>.txt
set -o pipefail
ROOT=$(pwd)
W=${1:-fill1}
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM" \
           "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/${W}_g$i -o p -- python3 $ROOT/tools/prof_target.py $W 3 > /dev/null 2> $OUT/${W}_g$i.stderr || echo "group $i failed"
done
cd $ROOT
python3 - "$OUT" "$W" <<'PY'
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/{w}_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/{w}.txt", "w") as fo:
    for k, cs in acc.items():
        if "k_fill" not in k and "k_sweep" not in k and "k_grid" not in k:
            continue
        print(k, file=fo)
        for c, v in sorted(cs.items()):
            print(f"  {c:32s} {sum(v)/len(v):16.0f}  (avg of {len(v)} launches)", file=fo)
print(open(f"{out}/{w}.txt").read())
PY
find $OUT -name "*.db" | xargs -r rm -f

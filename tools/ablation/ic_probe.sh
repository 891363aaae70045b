cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for w in fill1 fill8 grid8; do
  timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $ROOT/gpurun_out/ic_$w -o p -- python3 $ROOT/tools/prof_target.py $w 3 > /dev/null 2> $ROOT/gpurun_out/ic_$w.err || echo "failed $w"
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for w in ("fill1","fill8","grid8"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/ic_{w}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "k_fill" in k:
            print(w, k, {c: sum(v)/len(v) for c, v in cs.items()})
PY

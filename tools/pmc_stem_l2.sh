R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_stem2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in default s012skip8 s012skip32; do
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/$v -- python3 $R/tools/bench_stem.py $v > $O/$v.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/$v/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "stem012" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$v", {k: "%.3g" % (sum(v[3:]) / len(v[3:])) for k, v in agg.items()})
PY
done

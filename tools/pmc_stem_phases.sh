#!/bin/bash
# LDS bank conflicts of the fused stem BY PHASE: one SQ pass per timing-only build (tools/build_variant.py s012skipN stem012.hip
# -DPPN_S012_SKIP=N; 1 convert, 2 layer 0, 4 layer 1, 8 layer 2 left out) -- the difference to the full kernel is that phase's share
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/${1:-stem_phases}"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
for v in default s012skip1 s012skip2 s012skip4 s012skip8; do
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/$v -- python3 $R/tools/bench_stem.py $v > $O/$v.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_summary.py $O stem012 2>/dev/null | head -0
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/$v/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "stem012" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$v", {k: "%.3g" % (sum(v[3:]) / len(v[3:])) for k, v in agg.items()})
PY
  grep round $O/$v.log | tail -1
done

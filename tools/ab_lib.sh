#!/bin/bash
# same-box A/B of a library variant (tools/bin/libppn_NAME.so) against the in-tree build: bench three lanes / one lane, training step
N=$1; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${2:-ablib}; mkdir -p $O; cd $R
for i in 1 2; do
  for v in default $N; do
    if [ $v = default ]; then unset PPN_LIB; else export PPN_LIB=$R/tools/bin/libppn_$v.so; fi
    timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes 2 > $O/l3_${v}_$i.json 2>/dev/null
    timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes 1 > $O/l1_${v}_$i.json 2>/dev/null
    timeout -k 10 300 python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/tr_${v}_$i.json 2>/dev/null
    python3 - <<PY
import json
a=json.load(open("$O/l3_${v}_$i.json")); b=json.load(open("$O/l1_${v}_$i.json")); c=json.load(open("$O/tr_${v}_$i.json"))
print("$v run $i: two lanes", a["value"], a["value_windows"]["median"], "| one lane", b["value"], b["value_windows"]["median"], "conv stack ms", b["conv_stack"]["ms"], "| train ms/step", c["ms_per_step"])
PY
  done
done

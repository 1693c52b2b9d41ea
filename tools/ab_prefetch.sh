#!/bin/bash
# same-box A/B of the weight-prefetch hint (ppn_conv_desc.prefetch): bench.py one lane and three lanes, PPN_PREFETCH=0/1, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05b}; mkdir -p $O; cd $R
for i in 1 2; do
  for pf in 0 1; do
    PPN_PREFETCH=$pf timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes 3 > $O/l3_pf${pf}_$i.json 2>/dev/null
    PPN_PREFETCH=$pf timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes 1 --layers > $O/l1_pf${pf}_$i.json 2> $O/l1_pf${pf}_$i.layers.txt
    python3 - <<PY
import json
for n in ("l3","l1"):
    d=json.load(open("$O/%s_pf${pf}_$i.json"%n))
    print("prefetch=$pf run $i", n, d["value"], d["value_windows"]["median"], d["roofline"]["avg_launch_us"], d.get("conv_stack",{}).get("ms"))
PY
  done
done

#!/bin/bash
# same-box A/B of an environment switch on the benchmark: tools/ab_env.sh VAR TAG [rounds]   (VAR=0 vs VAR=1, interleaved; three lanes
# = the headline, one lane with the per-launch table)
V=$1; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${2:-ab}; N=${3:-2}; mkdir -p $O; cd $R
for i in $(seq 1 $N); do
  for on in 0 1; do
    env $V=$on timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes 3 > $O/l3_${on}_$i.json 2>/dev/null
    env $V=$on timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes 1 --layers > $O/l1_${on}_$i.json 2> $O/l1_${on}_$i.layers.txt
    python3 - <<PY
import json
for n in ("l3","l1"):
    d=json.load(open("$O/%s_${on}_$i.json"%n))
    print("$V=$on run $i", n, d["value"], d["value_windows"]["median"], d["roofline"]["avg_launch_us"], d.get("conv_stack",{}).get("ms"))
PY
  done
done
grep -v amdgpu $O/l1_0_1.layers.txt | awk '{printf "%-42s %8s\n", $1, $2}' | head -12 > $O/a.txt; grep -v amdgpu $O/l1_1_1.layers.txt | awk '{printf "%-42s %8s\n", $1, $2}' | head -12 > $O/b.txt; paste $O/a.txt $O/b.txt

#!/bin/bash
# tools/ab_env_lanes.sh VAR LANES [rounds]: VAR=0 vs VAR=1 on bench.py --lanes LANES, interleaved
V=$1; Ln=$2; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abl
for i in $(seq 1 ${3:-3}); do for on in 0 1; do
env $V=$on timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes $Ln > gpurun_out/abl/l_${on}_$i.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/abl/l_${on}_$i.json')); print('$V=$on lanes $Ln run $i', d['value'], d['value_windows']['min'], d['value_windows']['median'], d['value_windows']['max'])"
done; done

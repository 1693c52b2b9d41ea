#!/bin/bash
# same-box A/B of an environment switch on the training step: tools/ab_train_env.sh VAR "v1 v2 ..." [rounds]
V=$1; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abtrain
for i in $(seq 1 ${3:-2}); do for val in $2; do
env $V=$val timeout -k 10 300 python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/abtrain/tr_${val}_$i.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/abtrain/tr_${val}_$i.json')); print('train $V=$val run $i', d['ms_per_step'], d['value'])"
done; done

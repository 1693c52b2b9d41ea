#!/bin/bash
# Same-box A/B of the training step between the in-tree libppn.so and a variant built by tools/build_variant.py:
#   python tools/build_variant.py NAME train.hip -DFLAG=1 && gpurun -- 'bash tools/ab_train_lib.sh NAME'
# (round 2: an 18 KB instead of 35 KB reduction buffer in bn_reduce_kernel -- 24.69-25.55 vs 24.71-25.02 ms, no change)
V=${1:-bnlds}
for rep in 1 2 3; do
for v in default $V; do
if [ $v = default ]; then unset PPN_LIB; else export PPN_LIB=${GRAFT_REPO_ROOT:-/root/repo}/tools/bin/libppn_$v.so; fi
timeout -k 10 200 python tools/bench_train.py --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
done; done

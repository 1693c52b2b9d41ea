#!/bin/bash
# like ab_train_env.sh, but also prints the duration of the named kernel family from a short rocprofv3 run per value
# tools/ab_train_env2.sh VAR "v1 v2" KERNEL_SUBSTRING
V=$1; K=$3; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abtrain2; export TMPDIR=/tmp
for val in $2; do
  (cd /tmp && env $V=$val rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abtrain2/p_$val -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --steps 3 --warmup 1 > /dev/null 2>&1)
  f=$(find gpurun_out/abtrain2/p_$val -name "*kernel_stats.csv" | head -1)
  echo "$V=$val: $(grep "$K" $f | head -3 | cut -d, -f1-4 | cut -c1-60,60-200)"
  find gpurun_out/abtrain2/p_$val -name "*.csv" -size +1M -delete
done

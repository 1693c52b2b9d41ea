#!/bin/bash
# Kernel trace of the training step + the timeline / small-launch listing of its last iteration (run through gpurun).
#   tools/train_trace.sh TAG [families]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-tt}; O=$R/gpurun_out/$T; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/bench_train.py --steps 10 --warmup 3 > $O/bench.txt 2>&1; tail -2 $O/bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o train -- python3 $R/tools/bench_train.py --steps 3 --warmup 1 > $O/profiled.txt 2>&1
TR=$(find $O/stats -name "*kernel_trace.csv" | head -1)
(cd $R && python3 tools/train_timeline.py $TR --small $2) > $O/timeline.txt 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O -name "*.csv" -size +3M -delete
head -30 $O/timeline.txt

#!/bin/bash
# rocprofv3 kernel stats of the full (second-order) and first-order training step -> gpurun_out/<tag>/train_*.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r03t}; O="$R/gpurun_out/$T"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/bench_train.py --steps 5 --warmup 2 > $O/train_bench.txt 2>&1
python3 $R/tools/bench_train.py --steps 5 --warmup 2 --first-order >> $O/train_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -o train -- python3 $R/tools/bench_train.py --steps 3 --warmup 1 > $O/train_profiled.txt 2>&1
cp $(find $O/train_stats -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
find $O -name "*.csv" -size +3M -delete
tail -3 $O/train_bench.txt

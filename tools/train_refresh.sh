#!/bin/bash
# Training-step evidence only (a subset of refresh_profiles.sh): bench, rocprofv3 kernel stats of the first- and
# second-order step, the main-stream timeline and the host timeline.  tools/train_refresh.sh TAG -> gpurun_out/TAG/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r04t}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/bench_train.py --phases --first-order > $O/train_bench.txt 2>&1
python3 $R/bench.py --workload train --steps 10 --warmup 2 > $O/bench_train.json 2> $O/bench_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -o train -- python3 $R/tools/bench_train.py --steps 3 --warmup 1 --first-order > $O/train_profiled.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats_so -o train -- python3 $R/tools/bench_train.py --steps 3 --warmup 1 > $O/train_profiled_so.txt 2>&1
(cd $R && python3 tools/train_timeline.py $(find $O/train_stats_so -name "*kernel_trace.csv" | head -1)) > $O/train_timeline.txt 2>&1
(cd $R && python3 tools/host_timeline.py) 2>&1 | grep -v -i "warn\|amdgpu\|local_pass" > $O/train_host_timeline.txt
(cd $R && python3 tools/bench_wgrad.py) 2>&1 | grep -v amdgpu.ids > $O/wgrad_bench.txt
find $O -name "*.csv" -size +3M -delete
echo done

#!/bin/bash
# round 5, first GPU call: same-box baseline (bench --layers), in-sequence vs back-to-back stamps, MFMA-busy PMC pass
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
timeout -k 10 400 python3 bench.py --layers --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench_layers.txt; echo "bench rc=$?"; head -c 700 $O/bench.json; echo
timeout -k 10 300 python3 tools/clock_conv_seq.py > $O/clock_seq.txt 2>&1; echo "clock_seq rc=$?"; grep -v amdgpu.ids $O/clock_seq.txt | tail -60
timeout -k 10 400 bash tools/pmc_mfma.sh r05a > $O/pmc_mfma.log 2>&1; echo "pmc rc=$?"; tail -45 $O/pmc_mfma.log

#!/bin/bash
# Re-measure everything profiles/ holds, on the GPU box (run through gpurun).  Output: gpurun_out/<tag>/...
#   tools/refresh_profiles.sh r01
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r01}; O="$R/gpurun_out/$T"; rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
echo "[1/6] bench (unprofiled, with per-launch table and CPU baseline)"
python3 $R/bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
echo "[2/6] rocprofv3 kernel stats of bench.py: one lane on the multi-lane plan (--shared-plan: per-dispatch durations comparable with the HIP-event table of the headline) and the default lanes"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify --lanes 1 --shared-plan > $O/bench_profiled.json 2> $O/bench_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -o bench2 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify > $O/bench_profiled_lanes.json 2> $O/bench_profiled_lanes.err
echo "[3/6] PMC pass FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-verify --lanes 1 --shared-plan > $O/fetch.json 2> $O/fetch.err
echo "[4/6] PMC pass WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-verify --lanes 1 --shared-plan > $O/write.json 2> $O/write.err
echo "[5/6] training step (phases + JSON), bench.py --workload train"
python3 $R/tools/bench_train.py --phases --first-order > $O/train_bench.txt 2>&1
python3 $R/bench.py --workload train --steps 10 --warmup 2 > $O/bench_train.json 2> $O/bench_train.err
echo "[6/6] rocprofv3 kernel stats of the training step"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -o train -- python3 $R/tools/bench_train.py --steps 3 --warmup 1 --first-order > $O/train_profiled.txt 2>&1
echo "[6b] rocprofv3 kernel stats + traffic of the stand-alone decode on planted-crowd heads (BASELINE configs[4])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/decode_stats -o decode -- python3 $R/tools/bench_decode.py 32 > $O/decode_profiled.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/decode_fetch -- python3 $R/tools/bench_decode.py 32 > /dev/null 2>&1 || true
echo "[7] in-kernel clock and phase stamps of the dominant conv kernel (diagnostic build: python tools/build_variant.py clock conv_big.hip -DPPN_CLOCK)"
if [ -f $R/tools/bin/libppn_clock.so ]; then
  (cd $R && python3 tools/clock_conv.py && python3 tools/clock_conv.py --head) 2>&1 | grep -v amdgpu.ids > $O/conv_clock.txt
fi
if [ -f $R/tools/bin/libppn_clockhead.so ]; then (cd $R && python3 tools/clock_head.py) 2>&1 | grep -v amdgpu.ids > $O/head_clock.txt; fi
if [ -f $R/tools/bin/libppn_clock64.so ]; then (cd $R && python3 tools/clock_conv64.py) 2>&1 | grep -v amdgpu.ids > $O/conv64_clock.txt; fi
if [ -f $R/tools/bin/libppn_clockwg.so ]; then (cd $R && python3 tools/clock_wgrad.py) 2>&1 | grep -v amdgpu.ids > $O/wgrad_clock.txt; fi
echo "[7b] weight-gradient micro-benchmark + its SQ / TCC counters (separate --pmc passes)"
(cd $R && python3 tools/bench_wgrad.py) 2>&1 | grep -v amdgpu.ids > $O/wgrad_bench.txt
(cd $R && tools/pmc_wgrad.sh $T/pmc_wgrad > /dev/null 2>&1 && python3 tools/pmc_summary.py gpurun_out/$T/pmc_wgrad wgrad_kernel > $O/wgrad_pmc.txt) || true
echo "[8] full (second-order) training step: kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats_so -o train -- python3 $R/tools/bench_train.py --steps 3 --warmup 1 > $O/train_profiled_so.txt 2>&1
(cd $R && python3 tools/train_timeline.py $(find $O/train_stats_so -name "*kernel_trace.csv" | head -1)) > $O/train_timeline.txt 2>&1 || true
(cd $R && python3 tools/host_timeline.py) 2>&1 | grep -v -i "warn\|amdgpu\|local_pass" > $O/train_host_timeline.txt || true
python3 $R/tools/pmc_traffic_summary.py $O $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1 || true
echo "[9] MFMA utilisation by counter (one SQ + GRBM pass over bench.py --lanes 1 --shared-plan): tools/pmc_mfma.sh"
(cd $R && bash tools/pmc_mfma.sh $T/pmc_mfma > $O/pmc_mfma.log 2>&1 && cp $O/pmc_mfma/mfma_busy.txt $O/mfma_busy.txt && cp $O/pmc_mfma/mfma_busy.json $O/mfma_busy.json) || true
echo "[10] in sequence vs back to back (tools/clock_conv_seq.py, -DPPN_CLOCK=2 build) and the one-launch BasicBlock's phase stamps"
if [ -f $R/tools/bin/libppn_clock2.so ]; then (cd $R && python3 tools/clock_conv_seq.py) 2>&1 | grep -v amdgpu.ids > $O/conv_in_sequence.txt || true; fi
if [ -f $R/tools/bin/libppn_clockb64.so ]; then (cd $R && python3 tools/clock_block64.py) 2>&1 | grep -v amdgpu.ids > $O/block64_clock.txt || true; fi
find $O -name "*.csv" -size +3M -delete
echo done

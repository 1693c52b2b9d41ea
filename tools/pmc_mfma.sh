#!/bin/bash
# MFMA utilisation BY COUNTER over the benchmarked step (north_star: "rocprof reports ... MFMA utilisation for
# the conv stack"): one PMC pass (SQ + GRBM only, never combined with a trace domain other than --kernel-trace) over bench.py with the
# multi-lane plan's kernels, one launch in flight.  Run on the GPU box:  bash tools/pmc_mfma.sh <outdir-under-gpurun_out>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/${1:-pmc_mfma}"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/bench.py --steps 6 --warmup 2 --lanes 1 --shared-plan --no-extras --no-verify --no-cpu-baseline > $O/mfma.json 2> $O/mfma.err
python3 $R/tools/pmc_mfma_summary.py $O/mfma $O/mfma_busy.txt $O/mfma_busy.json
echo done

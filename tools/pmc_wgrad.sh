#!/bin/bash
# usage: tools/pmc_wgrad.sh <outdir>   (GPU box; separate PMC passes over tools/bench_wgrad.py; summary: tools/pmc_summary.py <dir> wgrad_kernel)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/$1"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p1 -- python3 $R/tools/bench_wgrad.py layer6 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/p2 -- python3 $R/tools/bench_wgrad.py layer6 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/p3 -- python3 $R/tools/bench_wgrad.py layer6 > $O/p3.log 2>&1
echo done

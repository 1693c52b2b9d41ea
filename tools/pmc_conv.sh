#!/bin/bash
# usage: tools/pmc_conv.sh <shape-filter> <outdir>   (runs on the GPU box; three separate PMC passes)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
F="$1"; O="$R/gpurun_out/$2"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p1 -- python3 $R/tools/bench_conv.py "$F" > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_INSTS_SALU --output-format csv -d $O/p2 -- python3 $R/tools/bench_conv.py "$F" > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/p3 -- python3 $R/tools/bench_conv.py "$F" > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $O/p4 -- python3 $R/tools/bench_conv.py "$F" > $O/p4.log 2>&1 || true
echo done

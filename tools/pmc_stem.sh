#!/bin/bash
# usage: tools/pmc_stem.sh <outdir> [variant]   (runs on the GPU box; separate PMC passes over tools/bench_stem.py)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/$1"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p1 -- python3 $R/tools/bench_stem.py ${2:-default} > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/p2 -- python3 $R/tools/bench_stem.py ${2:-default} > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_LEVEL_WAVES --output-format csv -d $O/p3 -- python3 $R/tools/bench_stem.py ${2:-default} > $O/p3.log 2>&1 || true
echo done

#!/bin/bash
# HBM traffic of the bench kernels: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over `bench.py`,
# as MI355X_MICROARCH.md prescribes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2).  Run on the GPU box.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/${1:-pmc_bench}"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/write.json 2> $O/write.err
echo done

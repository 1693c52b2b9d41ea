#!/bin/bash
# One GPU-box call of the round: full GPU test suite, then (unless a step was killed at its limit) the named extras.
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh TAG [tests|clock|bench|benchlite ...]'
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04x}; shift
O=$R/gpurun_out; mkdir -p $O
cd $R
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
rc=0
for step in "$@"; do
  ok $rc || { echo "previous step was killed at its limit: stopping"; exit 1; }
  case $step in
    tests)  timeout -k 10 1000 python3 -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 -x > $O/${T}_pytest.log 2>&1; rc=$?; tail -5 $O/${T}_pytest.log;;
    tests_all) timeout -k 10 1100 python3 -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/${T}_pytest.log 2>&1; rc=$?; tail -15 $O/${T}_pytest.log;;
    tests:*) timeout -k 10 1000 python3 -m pytest ${step#tests:} -q -m gpu -p no:cacheprovider --timeout 900 -s > $O/${T}_pytest_sel.log 2>&1; rc=$?; tail -25 $O/${T}_pytest_sel.log;;
    clock)  (timeout -k 10 120 python3 tools/clock_conv.py; timeout -k 10 120 python3 tools/clock_conv.py --f16) 2>&1 | grep -v amdgpu.ids > $O/${T}_clock.txt; rc=$?; cat $O/${T}_clock.txt;;
    bench)  timeout -k 10 600 python3 bench.py --layers > $O/${T}_bench.json 2> $O/${T}_bench_layers.txt; rc=$?; head -c 1500 $O/${T}_bench.json; echo;;
    benchlite) timeout -k 10 300 python3 bench.py --layers --no-extras --no-cpu-baseline > $O/${T}_benchlite.json 2> $O/${T}_benchlite_layers.txt; rc=$?; head -c 600 $O/${T}_benchlite.json; echo;;
    *) echo "unknown step $step";;
  esac
  echo "step $step rc=$rc"
done

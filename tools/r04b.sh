cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_conv_tiles_gpu.py tests/test_fullsize_gpu.py tests/test_conv_gpu.py tests/test_x3_gpu.py -q -m gpu -p no:cacheprovider -x > $O/r04b_pytest.log 2>&1; rc=$?; tail -4 $O/r04b_pytest.log
[ $rc -eq 124 ] && exit 1
for v in def off def2 off2; do
  if [ ${v:0:3} = off ]; then export PPN_EFF144=0; else unset PPN_EFF144; fi
  timeout -k 10 300 python3 bench.py --layers --no-extras --no-cpu-baseline > $O/r04b_bench_$v.json 2> $O/r04b_layers_$v.txt || exit 1
  python3 -c "
import json;r=json.load(open('$O/r04b_bench_$v.json'));print('$v',r['value'],r['value_windows']['median'],r['roofline']['kernel'],r['roofline']['frac'],r['conv_stack']['ms'])"
done

cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_decode_gpu.py tests/test_rt_gpu.py tests/test_fullsize_gpu.py -q -m gpu -p no:cacheprovider -x > $O/r04e_pytest.log 2>&1; rc=$?; tail -3 $O/r04e_pytest.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 200 python3 tools/bench_decode_fused.py 2>&1 | grep -v amdgpu.ids | tee $O/r04e_decode.txt
for v in frac nofrac frac2 nofrac2; do
  if [ ${v:0:6} = nofrac ]; then export PPN_SHARED_FRAC=0; else unset PPN_SHARED_FRAC; fi
  timeout -k 10 300 python3 bench.py --layers --no-extras --no-cpu-baseline > $O/r04e_bench_$v.json 2> $O/r04e_layers_$v.txt || exit 1
  python3 -c "
import json;r=json.load(open('$O/r04e_bench_$v.json'));print('$v',r['value'],r['value_windows'],r['roofline']['kernel'],r['roofline']['launches_per_step'],r['roofline']['frac'],r['conv_stack']['ms'],r['decode']['ms'])"
done

cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_decode_gpu.py tests/test_rt_gpu.py tests/test_e2e_gpu.py tests/test_conv_tiles_gpu.py -q -m gpu -p no:cacheprovider -x -k "not forced_tile" > $O/r04c_pytest.log 2>&1; rc=$?; tail -4 $O/r04c_pytest.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python3 bench.py --layers --no-extras --no-cpu-baseline > $O/r04c_bench.json 2> $O/r04c_layers.txt || exit 1
python3 -c "
import json;r=json.load(open('$O/r04c_bench.json'));print(r['value'],r['value_windows']['median'],r['roofline']['kernel'],r['roofline']['frac'],r['conv_stack']['ms'],r['decode'])"
PPN_DECODE_SPREAD=0 timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline > $O/r04c_bench_nospread.json 2> /dev/null || exit 1
python3 -c "
import json;r=json.load(open('$O/r04c_bench_nospread.json'));print('nospread',r['value'],r['value_windows']['median'],r['decode'])"
timeout -k 10 300 python3 bench.py --layers --no-extras --no-cpu-baseline --lanes 1 > $O/r04c_bench_1lane.json 2> $O/r04c_layers_1lane.txt || exit 1
python3 -c "
import json;r=json.load(open('$O/r04c_bench_1lane.json'));print('1lane',r['value'],r['value_windows']['median'],r['roofline']['kernel'],r['roofline']['frac'],r['conv_stack']['ms'],r['decode'])"

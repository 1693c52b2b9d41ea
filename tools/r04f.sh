cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_x3_gpu.py -q -m gpu -p no:cacheprovider -x -s > $O/r04f_pytest.log 2>&1; rc=$?; tail -3 $O/r04f_pytest.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python3 bench.py --dtype f16x3 --layers --no-extras --no-cpu-baseline --lanes 1 > $O/r04f_bench_x3_1lane.json 2> $O/r04f_layers_x3.txt || exit 1
cat $O/r04f_layers_x3.txt | grep -v Warn
python3 -c "
import json;r=json.load(open('$O/r04f_bench_x3_1lane.json'));print(r['value'],r['roofline']['kernel'],r['roofline']['frac'],r['conv_stack'])"

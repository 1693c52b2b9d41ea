cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_x3_gpu.py tests/test_forward_gpu.py tests/test_e2e_gpu.py tests/test_rt_gpu.py -q -m gpu -p no:cacheprovider -s -x > $O/r04i_pytest.log 2>&1; rc=$?; tail -4 $O/r04i_pytest.log; grep -E "exact prefix|HIP vs reference people" $O/r04i_pytest.log | cut -c1-250
[ $rc -eq 124 ] && exit 1
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python3 bench.py --no-cpu-baseline > $O/r04i_bench.json 2> $O/r04i_bench.err || { tail -5 $O/r04i_bench.err; exit 1; }
python3 -c "
import json;r=json.load(open('$O/r04i_bench.json'));f=r['f16_mode'];print(r['value'], f.get('images_per_sec'), f.get('f16_agreement',{}).get('reproduced_exactly'), f.get('exact_prefix_3'), f.get('error'))"

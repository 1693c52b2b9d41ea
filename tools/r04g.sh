cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_forward_gpu.py tests/test_e2e_gpu.py tests/test_fullsize_gpu.py tests/test_conv_gpu.py tests/test_rt_gpu.py -q -m gpu -p no:cacheprovider -s > $O/r04g_pytest.log 2>&1; rc=$?; tail -4 $O/r04g_pytest.log; grep -E "HIP vs reference people|bfloat16: vs emulated|float16: vs emulated" $O/r04g_pytest.log | cut -c1-330
[ $rc -eq 124 ] && exit 1
for v in f16stem bf16stem; do
  if [ $v = bf16stem ]; then export PPN_STEM_DTYPE=bfloat16; else unset PPN_STEM_DTYPE; fi
  timeout -k 10 400 python3 bench.py --no-cpu-baseline > $O/r04g_bench_$v.json 2> $O/r04g_bench_$v.err || { tail -5 $O/r04g_bench_$v.err; exit 1; }
  python3 -c "
import json;r=json.load(open('$O/r04g_bench_$v.json'));print('$v',r['value'],r['value_windows']['median'],r['bf16_agreement']['reproduced_exactly'],r['bf16_agreement']['same_root'],r['bf16_agreement']['tuned_checkpoint']['reproduced_exactly'],r['ap_vs_reference']['bf16'][-1],r['batch_consistency']['ok'])"
done

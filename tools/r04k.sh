cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_forward_gpu.py tests/test_e2e_gpu.py tests/test_conv_gpu.py tests/test_rt_gpu.py tests/test_fullsize_gpu.py tests/test_x3_gpu.py -q -m gpu -p no:cacheprovider -s > $O/r04k_pytest.log 2>&1; rc=$?; tail -4 $O/r04k_pytest.log; grep -E "exact prefix 3:|HIP vs reference people|f16:" $O/r04k_pytest.log | cut -c1-230
[ $rc -eq 124 ] && exit 1
timeout -k 10 500 python3 bench.py --no-cpu-baseline > $O/r04k_bench.json 2> $O/r04k_bench.err || { tail -5 $O/r04k_bench.err; exit 1; }
python3 -c "
import json;r=json.load(open('$O/r04k_bench.json'));f=r['f16_mode'];b=r['bf16_agreement'];print(r['value'], 'bf16', b['reproduced_exactly'], b['same_root'], b['tuned_checkpoint']['reproduced_exactly'], r['ap_vs_reference']['bf16'][-1], '| f16', f['images_per_sec'], f['f16_agreement']['reproduced_exactly'], f['f16_agreement_tuned_checkpoint']['reproduced_exactly'], r['ap_vs_reference']['f16'][-1], '| xp', f['exact_prefix_3']['images_per_sec'], f['exact_prefix_3']['agreement']['reproduced_exactly'])"

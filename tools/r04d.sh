cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python3 tools/bench_decode_fused.py 2>&1 | grep -v amdgpu.ids | tee $O/r04d_decode.txt
PPN_DECODE_SPREAD=0 timeout -k 10 200 python3 tools/bench_decode_fused.py 2>&1 | grep -v amdgpu.ids | tee -a $O/r04d_decode.txt
PPN_DECODE_SPREAD=16 timeout -k 10 200 python3 tools/bench_decode_fused.py 2>&1 | grep -v amdgpu.ids | tee -a $O/r04d_decode.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04d_prof -o dec -- python3 $GRAFT_REPO_ROOT/tools/bench_decode_fused.py > /dev/null 2>&1)
f=$(find $O/r04d_prof -name "*kernel_stats.csv" | head -1); grep -E "root_mask|parse_kernel" $f | cut -c1-60,200- ; grep -E "root_mask|parse_kernel" $f | awk -F, '{print $1" calls "$2" avg_ns "$4}' | cut -c1-200 | tee -a $O/r04d_decode.txt
find $O/r04d_prof -name "*.csv" -size +2M -delete
timeout -k 10 300 python3 tools/bench_ksplit_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r04d_ksplit_probe.txt

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05p
for i in 1 2; do for m in 0 1 2 4 3 7; do
if [ $m = 0 ]; then P=0; else P=1; fi
PPN_POLICY1_MASK=$m timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes 2 --tile-policy $P > gpurun_out/r05p/m${m}_$i.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r05p/m${m}_$i.json')); print('policy-1 mask $m run $i', d['value'], d['value_windows']['min'], d['value_windows']['median'], d['value_windows']['max'], 'conv stack in sequence', d['conv_stack']['ms'])"
done; done

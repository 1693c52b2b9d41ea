cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05q
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r05q/full.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r05q/full.json')); print('headline', d['value'], 'tile_policy_1 section', d['tile_policy_1'])"
timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes 2 --tile-policy 1 > gpurun_out/r05q/p1.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r05q/p1.json')); print('standalone policy 1', d['value'], d['value_windows'])"

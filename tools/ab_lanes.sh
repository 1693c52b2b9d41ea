cd $GRAFT_REPO_ROOT
for l in 1 2 3 4; do for tp in 0 1; do
timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes $l --tile-policy $tp > gpurun_out/r05d/lanes_${l}_${tp}.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r05d/lanes_${l}_${tp}.json')); print('lanes $l policy $tp', d['value'], d['value_windows']['min'], d['value_windows']['median'], d['value_windows']['max'])"
done; done

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05n
for i in 1 2 3; do for l in 2 3; do
timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --no-verify --lanes $l > gpurun_out/r05n/l${l}_$i.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r05n/l${l}_$i.json')); print('lanes $l run $i', d['value'], d['value_windows']['min'], d['value_windows']['median'], d['value_windows']['max'])"
done; done

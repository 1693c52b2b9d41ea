cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05h
for i in 1 2; do for pf in 0 1; do
PPN_PREFETCH=$pf timeout -k 10 300 python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05h/tr_${pf}_$i.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r05h/tr_${pf}_$i.json')); print('train prefetch=$pf run $i', d['ms_per_step'], d['value'])"
done; done

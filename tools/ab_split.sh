#!/bin/bash
# Same-box A/B of the two-segment conv launches (tile policy 2) against the default single-tile choice.
# Round 2 result: policy 2 is 1.8 % (three lanes) to 3.6 % (one lane) slower; profiles/r02/ab_split.txt.
for rep in 1 2; do
for v in 2 0; do
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --layers --tile-policy $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('policy=$v lanes3', d['value'], d['ms_per_step'], 'stack', d['conv_stack']['ms'], d['roofline']['kernel'][-22:], d['roofline']['frac'])"
done; done

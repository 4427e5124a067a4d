#!/bin/bash
# A/B of the chained hidden layers on the bench's own step: tools/ab_chain.sh [bench args]
for rep in 1 2; do
for c in 0 1; do
  FDNN_CHAIN=$c python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-lazy --no-small --no-serving "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('chain=$c', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'single', d.get('single_stream',{}).get('ms_per_step'))"
done; done

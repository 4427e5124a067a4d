#!/bin/bash
# Layer-0 time per frame at batch sizes that fill whole / partial rounds of workgroups.
for n in ${L0_SIZES:-4096 8192 10000 10240 12288 16384}; do
  FDNN_BENCH_NOCHECK=1 python bench.py --frames $n --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_step']
        print($n, 'l0 ms', k['l0'], 'ns/frame', round(k['l0'] * 1e6 / $n, 2), 'total', d['ms_per_step'])
"
done

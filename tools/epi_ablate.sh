#!/bin/bash
# ON THE GPU BOX: hidden-layer epilogue ablations (variants built beforehand by tools/build_variant.sh)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
for v in "" base dbg8 dbg16 dbg32 dbg24; do
  if [ -z "$v" ]; then L=""; else L="FDNN_LIB=$R/fast-dnn_amd/lib/libfast-dnn-$v.so"; [ -f "$R/fast-dnn_amd/lib/libfast-dnn-$v.so" ] || continue; fi
  echo "== variant '$v'"
  env $L FDNN_BENCH_NOCHECK=1 python bench.py --no-cpu-baseline --steps 100 --warmup 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
"
done

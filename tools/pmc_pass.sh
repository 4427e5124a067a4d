#!/bin/bash
# One ad-hoc rocprofv3 counter pass over bench.py, ON THE GPU BOX, printing per-kernel averages:
#   gpurun --timeout 600 -- 'bash tools/pmc_pass.sh l0a SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS ...'
# (at most ~8 SQ counters per pass; the pass runs under `timeout` because rocprofv3 hangs after
# rejecting a counter set the hardware cannot collect together)
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $ROOT/gpurun_out
timeout -k 5 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o p -- \
  python $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 ${BENCH_ARGS:-} > /dev/null 2> $ROOT/gpurun_out/pmc_$TAG.log
echo "rocprofv3 rc=$?"
python - "$OUT/p_counter_collection.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "fdnn" not in k or "fastdiv" in k or "xor80" in k:
        continue
    name = k.split("::")[-1].split("(")[0][:60]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    print(n, {k: round(sum(v[2:]) / max(1, len(v[2:]))) for k, v in c.items()})
PY

#!/bin/bash
# rocprofv3 kernel stats of one bench.py run, condensed: tools/kstats.sh [bench args]
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats_prof
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_prof -o k -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline "$@" > /dev/null 2>&1
python - <<PY
import csv
for r in list(csv.DictReader(open("/tmp/kstats_prof/k_kernel_stats.csv")))[:9]:
    print(r["Name"].replace("fdnn::(anonymous namespace)::", "")[:60].ljust(60), r["Calls"].rjust(5), "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "min %.1f" % (float(r["MinNs"]) / 1e3))
PY

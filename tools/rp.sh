#!/bin/bash
# rocprofv3 kernel stats of any command, condensed: tools/rp.sh <command ...>
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_prof
( cd $R && timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_prof -o k -- "$@" > /tmp/rp_cmd.log 2>&1 )
tail -4 /tmp/rp_cmd.log
python - <<PY
import csv, glob
f = glob.glob("/tmp/rp_prof/**/k_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"].replace("fdnn::(anonymous namespace)::", "")[:70].ljust(70), r["Calls"].rjust(6), "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "min %.1f" % (float(r["MinNs"]) / 1e3), "%s %%" % r["Percentage"])
PY

#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats + separate PMC passes over a stream of small / mid-size
# calls (tools/small_call.py), condensed into gpurun_out/<tag>_small_<frames>_{kernel_stats.csv,pmc.json}.
#   gpurun --timeout 1200 -- 'bash tools/profile_small.sh r06 100; bash tools/profile_small.sh r06 1000'
# PMC passes never combine with sys/hip/hsa tracing (node stability); every pass runs under its own timeout.
set -u
TAG=${1:-rXX}; N=${2:-100}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
CMD="python $ROOT/tools/small_call.py $N 300"
T="timeout -k 5 240"
P=$OUT/prof_${TAG}_small_$N
$T rocprofv3 --kernel-trace --stats --output-format csv -d ${P} -o k -- $CMD > $OUT/${TAG}_small_${N}_call.log 2> $OUT/${TAG}_small_${N}_stats.log
cp ${P}/k_kernel_stats.csv $OUT/${TAG}_small_${N}_kernel_stats.csv 2>/dev/null
pass() { local name=$1; shift; $T rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d ${P}_pmc_$name -o p -- $CMD > /dev/null 2> $OUT/${TAG}_small_${N}_pmc_$name.log; echo "pass $name rc=$?"; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python $ROOT/tools/small_pmc_summary.py $OUT/${TAG}_small_${N}_pmc.json ${P}_pmc_sq/p_counter_collection.csv ${P}_pmc_fetch/p_counter_collection.csv ${P}_pmc_write/p_counter_collection.csv ${P}_pmc_tcc/p_counter_collection.csv <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[2:]:
    try:
        rows = list(csv.DictReader(open(f)))
    except Exception as e:
        print("missing", f, e); continue
    for r in rows:
        k = r["Kernel_Name"]
        if "fdnn" not in k or "fastdiv" in k or "xor80" in k or "image" in k:
            continue
        name = k.split("fdnn::")[-1].split("(")[0].replace("(anonymous namespace)::", "")[:80]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for n, c in acc.items():
    d = {k: round(sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])), 1) for k, v in c.items()}  # second half of the dispatches: warm
    d["dispatches"] = max(len(v) for v in c.values())
    if "FETCH_SIZE" in d: d["hbm_side_read_bytes"] = int(2 * d["FETCH_SIZE"] * 1000)  # gfx950: FETCH_SIZE counts 128-B requests as 64 B
    if "WRITE_SIZE" in d: d["hbm_side_write_bytes"] = int(d["WRITE_SIZE"] * 1000)
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d: d["l2_hit"] = round(d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d: d["mfma_busy_of_sq_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, d["SQ_BUSY_CYCLES"]), 4)
    out[n] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
cat $OUT/${TAG}_small_${N}_call.log; head -14 $OUT/${TAG}_small_${N}_kernel_stats.csv | cut -c1-200

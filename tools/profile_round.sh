#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats + separate PMC passes over
# `python bench.py`, then condense them into profiles/-style summaries under gpurun_out/.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01'
# Copy gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_summary.json and <tag>_bench_under_rocprof.json
# into profiles/ afterwards.  PMC passes never combine with sys/hip/hsa tracing (node stability).
# Every pass runs under its own `timeout`: a counter set the hardware cannot collect in one
# pass makes rocprofv3 abort and then hang instead of exiting.
set -u
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
BENCH="python $ROOT/bench.py --single-stream-only --steps 40 --warmup 5"
T="timeout -k 5 240"

$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o k -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.log
cp $OUT/prof_$TAG/k_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null

pass() {  # name counters...
  local name=$1; shift
  $T rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/prof_${TAG}_pmc_$name -o p -- $BENCH > /dev/null 2> $OUT/${TAG}_pmc_$name.log
  echo "pass $name rc=$?"
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_LDS_BANK_CONFLICT
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum

python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_summary.json $OUT/prof_${TAG}_pmc_sq/p_counter_collection.csv \
  $OUT/prof_${TAG}_pmc_fetch/p_counter_collection.csv $OUT/prof_${TAG}_pmc_write/p_counter_collection.csv \
  $OUT/prof_${TAG}_pmc_tcc/p_counter_collection.csv
python $ROOT/tools/roofline_from_rocprof.py $OUT/${TAG}_kernel_stats.csv $OUT/${TAG}_pmc_summary.json $OUT/${TAG}_roofline.json
head -8 $OUT/${TAG}_kernel_stats.csv

cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py > gpurun_out/r02_bench_try2.json 2> gpurun_out/r02_bench_try2.err; tail -2 gpurun_out/r02_bench_try2.err
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_try2.json'))
print(d['value'], d['ms_per_step'], d['single_stream'], d['lazy_40pct']['frames_per_s'], d['cpu_baseline']['value'], d['kernel_ms_per_step'])
"

cd $GRAFT_REPO_ROOT
python bench.py --steps 100 --warmup 10 > gpurun_out/r02_bench_try1.json 2> gpurun_out/r02_bench_try1.err
tail -3 gpurun_out/r02_bench_try1.err
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_try1.json'))
print(d['value'], d['ms_per_step'], d['single_stream'], d['lazy_40pct']['frames_per_s'], d['cpu_baseline'])
print(json.dumps(d['roofline'])[:600])
"

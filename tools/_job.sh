cd $GRAFT_REPO_ROOT
for w in 96 128 192 256; do
  echo "== nt, FDNN_NORM_BG_WGS=$w"
  FDNN_NORM_BG_WGS=$w STEPS=150 timeout 300 python tools/server_bench.py 2>&1 | grep single_stream | cut -c1-160
done
python -m pytest tests/test_gpu_server.py tests/test_gpu_production_shapes.py -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r02_bench_try3.json 2> gpurun_out/r02_bench_try3.err; tail -2 gpurun_out/r02_bench_try3.err
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_try3.json'))
print(d['value'], d['ms_per_step'], d['single_stream']['frames_per_s'], d['lazy_40pct']['frames_per_s'], d['cpu_baseline']['value'], d['serving'])
"

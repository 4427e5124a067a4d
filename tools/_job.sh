cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for w in 384 512 768; do
  echo "== FDNN_NORM_BG_WGS=$w"
  FDNN_NORM_BG_WGS=$w STEPS=100 timeout 300 python tools/server_bench.py 2>&1 | grep single_stream
done
FDNN_NORM_BG_WGS=512 bash tools/overlap_trace.sh 2 2>&1 | tail -12

cd $GRAFT_REPO_ROOT
for ft in 0 128 160 256 320; do
  echo "== FDNN_FRAME_TILE=$ft (0 = library's choice)"
  FDNN_FRAME_TILE=$ft FDNN_GEMM_160=s1 FRAMES="1500 2000 2560 3000 4000 5000 6000 7000 8000" python tools/l0_scan.py 2>&1 | grep -v amdgpu | awk '{print $1, $8, $9, $10, $11}'
done

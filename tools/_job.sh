cd $GRAFT_REPO_ROOT
FDNN_LIB=$GRAFT_REPO_ROOT/fast-dnn_amd/lib/libfast-dnn-dbgts.so python tools/wg_timeline.py 2>&1 | grep -v amdgpu | tail -16

cd $GRAFT_REPO_ROOT
python tools/l0_scan.py 2>&1 | grep -v amdgpu

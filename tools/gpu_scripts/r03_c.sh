cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/dup_probe.py 2>&1 | grep -v amdgpu

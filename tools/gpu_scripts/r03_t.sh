cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout -k 10 300 python tools/update_times.py 2>&1 | grep -v amdgpu | tail -3
timeout -k 10 300 python tools/update_times.py 24 24 2048 2>&1 | grep -v amdgpu | tail -2

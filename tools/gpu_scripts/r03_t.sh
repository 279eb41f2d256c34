cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep -v amdgpu | tail -2
timeout -k 10 300 python tools/curriculum_iter.py 512 60 2>&1 | grep -v amdgpu | head -1

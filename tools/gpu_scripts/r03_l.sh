cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout -k 10 300 python tools/curriculum_iter.py 512 60 2>&1 | grep -v amdgpu
timeout -k 10 300 python tools/curriculum_iter.py 1024 60 2>&1 | grep -v amdgpu

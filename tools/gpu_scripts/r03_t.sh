cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
timeout -k 10 300 python tools/curriculum_iter.py 512 60 2>&1 | grep -v amdgpu
timeout -k 10 300 python tools/actor_times.py 2>&1 | grep "reuse="

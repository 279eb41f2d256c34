cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_update_gpu.py tests/test_learner_gpu.py tests/test_big_goldens_gpu.py tests/test_entrypoints_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep -v amdgpu | tail -2

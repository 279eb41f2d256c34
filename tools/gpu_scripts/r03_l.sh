cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_env_gpu.py tests/test_reset_gpu.py tests/test_actor_gpu.py -x -q -m gpu 2>&1 | tail -3
MAPF_STEP_PLANE=1 timeout -k 10 900 python -m pytest tests/test_env_gpu.py -x -q -m gpu 2>&1 | tail -3

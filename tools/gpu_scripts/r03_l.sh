cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_env_gpu.py tests/test_reset_gpu.py tests/test_actor_gpu.py tests/test_flush_gpu.py -q -m gpu -x 2>&1 | tail -3
MAPF_STEP_PLANE=0 timeout -k 10 900 python -m pytest tests/test_env_gpu.py -q -m gpu -x 2>&1 | tail -2
bash tools/gpu_scripts/r03_k2.sh 2>&1 | grep "==\|G =\|per-block"

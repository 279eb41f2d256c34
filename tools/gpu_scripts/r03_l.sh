cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_env_gpu.py tests/test_reset_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_scripts/r03_k2.sh 2>&1 | grep "==\|G =\|per-block"

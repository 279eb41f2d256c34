cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_actor_gpu.py tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -12

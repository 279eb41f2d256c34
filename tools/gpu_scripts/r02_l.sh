cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/learner_modes.py 2>&1 | tail -4
timeout -k 10 600 python -m pytest tests/test_learner_gpu.py tests/test_model_gpu.py tests/test_entrypoints_gpu.py -q -m gpu 2>&1 | tail -4

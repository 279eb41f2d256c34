cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_learner_gpu.py -q -m gpu -x -k "own_stream" 2>&1 | tail -5

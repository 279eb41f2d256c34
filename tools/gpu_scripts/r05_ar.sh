#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_update_gpu.py tests/test_relevance_gpu.py tests/test_learner_gpu.py tests/test_big_goldens_gpu.py -x -q -m gpu > $O/t_tiles.log 2>&1; echo rc=$?
tail -25 $O/t_tiles.log

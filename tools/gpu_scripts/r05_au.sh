#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05au; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_learner_gpu.py tests/test_update_gpu.py tests/test_curriculum_gpu.py tests/test_entrypoints_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo tests=$?
tail -5 $O/tests.log
for v in 1 0 1 0; do
MAPF_PREFETCH_TARGET=$v ITERS=300 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "graph=True  prune=True" | sed "s/^/prefetch_target=$v /"
done | tee $O/update6_times_prefetch_target.txt

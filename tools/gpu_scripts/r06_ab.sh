#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ab; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_update_gpu.py tests/test_learner_gpu.py tests/test_curriculum_gpu.py -q -x > $O/tests.log 2>&1; rc=$?; echo tests=$rc; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
WARM=300 ITERS=200 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "graph=True  prune=True" | tee -a $O/update6.txt
MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 200 2>&1 | grep "MODE=" | tee -a $O/train_loop.txt
done

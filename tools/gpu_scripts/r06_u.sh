#!/bin/bash
# the encoder kernels of a bucket-sized launch compute the true row count only (`_bounded` entry points): tests, then A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_encoder_gpu.py tests/test_update_gpu.py tests/test_learner_gpu.py tests/test_curriculum_gpu.py -q -x > $O/tests.log 2>&1; rc=$?; echo tests=$rc; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
for b in 1 0 1 0; do
WARM=300 MAPF_BOUNDED_ROWS=$b ITERS=200 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "graph=True  prune=True" | sed "s/^/bounded_rows=$b /" | tee -a $O/update6_bounded.txt
done
for b in 1 0 1 0; do
MAPF_BOUNDED_ROWS=$b MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 200 2>&1 | grep "MODE=" | sed "s/^/bounded_rows=$b /" | tee -a $O/train_loop_bounded.txt
done

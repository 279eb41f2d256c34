#!/bin/bash
# round 6: the reworked graph-replayed update (row tables in front of the fork, packs beside them, priorities in the prefetch stage,
# one launch for the six 3x3 weight gradients): tests, then timings A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out
t0=$(date +%s)
timeout -k 10 900 python -m pytest tests/test_encoder_gpu.py tests/test_update_gpu.py tests/test_learner_gpu.py tests/test_curriculum_gpu.py tests/test_big_goldens_gpu.py -q -x > $O/r06d_tests.log 2>&1; rc=$?; echo tests=$rc $(( $(date +%s) - t0 ))s
tail -15 $O/r06d_tests.log
[ $rc -eq 0 ] || exit 1
for m in 1 0; do
MAPF_WGRAD_MERGED=$m ITERS=200 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "prune=True" | sed "s/^/wgrad_merged=$m /" | tee -a $O/r06d_update6_times.txt
done
MAPF_WGRAD_MERGED=1 ITERS=60 timeout -k 10 300 python tools/update_times.py 40 32 4096 2>&1 | grep "prune=True" | sed "s/^/C2 wgrad_merged=1 /" | tee -a $O/r06d_update40_times.txt
MAPF_WGRAD_MERGED=0 ITERS=60 timeout -k 10 300 python tools/update_times.py 40 32 4096 2>&1 | grep "prune=True" | sed "s/^/C2 wgrad_merged=0 /" | tee -a $O/r06d_update40_times.txt

#!/bin/bash
# the train loop at the reference's shape under different stream arrangements; weight-gradient partition sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
for m in base lprio alow mask:128 mask:192 serial base; do
MODE=$m timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 300 2>&1 | grep "MODE=" | tee -a $O/train_loop_overlap.txt
done
for p in 16 18 20 21; do
WARM=300 MAPF_WGRAD_PARTS=$p ITERS=200 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "graph=True  prune=True" | sed "s/^/parts=$p /" | tee -a $O/update6_parts.txt
done

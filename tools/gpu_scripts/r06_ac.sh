#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ac; mkdir -p $O
for f in 1 0 1 0 1 0; do
MAPF_FOLD_FILLS=$f WARM=300 ITERS=300 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "graph=True  prune=True" | sed "s/^/fold_fills=$f /" | tee -a $O/update6.txt
done

#!/bin/bash
# steady-state (no captures in the timed stretch) update times, bucket step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
for cfg in "1024 20" "256 20" "1024 20" "256 20"; do
set -- $cfg
WARM=400 MAPF_GRAPH_UROW_STEP=$1 MAPF_WGRAD_PARTS=$2 ITERS=200 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "graph=True  prune=True" | sed "s/^/urow_step=$1 parts=$2 /" | tee -a $O/update6_times.txt
done

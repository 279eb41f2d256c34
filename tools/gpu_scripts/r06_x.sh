#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; mkdir -p $O
for v in 8192 100000 8192 100000; do
MAPF_SIDE_MAX_ROWS=$v ITERS=60 timeout -k 10 300 python tools/update_times.py 40 32 4096 2>&1 | grep "graph=False prune=True" | sed "s/^/C2 side_max_rows=$v /" | tee -a $O/update40_side.txt
done

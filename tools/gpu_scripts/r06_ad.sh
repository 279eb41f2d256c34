#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ad; mkdir -p $O
for p in 21 42 21 42 63; do
MAPF_WGRAD_PARTS=$p ITERS=60 timeout -k 10 300 python tools/update_times.py 40 32 4096 2>&1 | grep "graph=False prune=True" | sed "s/^/C2 wgrad_parts=$p /" | tee -a $O/update40_parts.txt
done

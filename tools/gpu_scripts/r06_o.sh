#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
for cfg in "1 0" "0 0" "1 1" "0 0" "1 0"; do
set -- $cfg
MAPF_STREAM_REGISTRY=$1 MAPF_STREAM_SPLIT_CAPTURE=$2 MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 100 2>&1 | grep "MODE=" | sed "s/^/registry=$1 split_capture=$2 /" | cut -c1-220 | tee -a $O/registry.txt
done

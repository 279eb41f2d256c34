#!/bin/bash
# two environments per workgroup (<= 32 agents, more environments than CUs): the working tree against HEAD, pair launches on and off
set -e
mkdir -p gpurun_out
SH="1x4096x6 1x4096x16 1x1400x16 1x1024x6 1x512x6 1x300x6 1x4096x24 1x4096x32 1x512x24 3x1025x17 18x192x6 18x192x24 1x4096x40"
echo "## MAPF_RECUR_PAIR=1 (default)" > gpurun_out/recur_pair.txt
MAPF_RECUR_PAIR=1 timeout -k 10 300 python tools/micro/recur_multi.py run $SH >> gpurun_out/recur_pair.txt 2>&1
echo "## MAPF_RECUR_PAIR=0" >> gpurun_out/recur_pair.txt
MAPF_RECUR_PAIR=0 timeout -k 10 300 python tools/micro/recur_multi.py run $SH >> gpurun_out/recur_pair.txt 2>&1
cat gpurun_out/recur_pair.txt

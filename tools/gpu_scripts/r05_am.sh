#!/bin/bash
# fixed-level soak at a shape whose policy recurrence steps two environments per workgroup (4096 x 6 agents, 20 x 20)
set -e
mkdir -p gpurun_out
timeout -k 10 400 python train.py --envs 4096 --agents 6 --map 20 --minutes 3 --interval 30 > gpurun_out/train_fixed6_pair.log 2>&1
tail -25 gpurun_out/train_fixed6_pair.log

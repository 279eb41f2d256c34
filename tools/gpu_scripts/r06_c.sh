#!/bin/bash
# round 6: 6-rank shared-GPU rehearsals of train.py (fixed level + curriculum), launched without torch.distributed.run (its agent
# process is a 7th process on the card: the box's guard killed r06_b there)
cd $GRAFT_REPO_ROOT
O=gpurun_out
t0=$(date +%s)
MAPF_TRAIN_SHARE_GPU=1 timeout -k 10 300 python tools/launch_ranks.py 6 train.py --agents 40 --map 32 --envs 256 --minutes 1 --interval 15 --learning-starts 20000 --dist-backend gloo > $O/r06c_train_fixed_6rank.log 2>&1; echo train_fixed6=$? $(( $(date +%s) - t0 ))s
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $O/r06c_train_fixed_6rank.log | tail -14
t0=$(date +%s)
MAPF_TRAIN_SHARE_GPU=1 timeout -k 10 300 python tools/launch_ranks.py 6 train.py --envs 128 --minutes 1 --interval 15 --learning-starts 20000 --dist-backend gloo > $O/r06c_train_curriculum_6rank.log 2>&1; echo train_cur6=$? $(( $(date +%s) - t0 ))s
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $O/r06c_train_curriculum_6rank.log | tail -16

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; 
for cfg in "0 0" "8 2" "8 3" "4 5" "16 1" "8 4" "0 0"; do
set -- $cfg
MAPF_RBWD_STAGGER_SLOTS=$1 MAPF_RBWD_STAGGER_UNITS=$2 timeout -k 10 120 python tools/micro/recur_bwd_time.py 18 192 40
done 2>&1 | grep -v amdgpu.ids | tee $O/rbwd_stagger.txt
for cfg in "0 0" "8 2"; do
set -- $cfg
MAPF_RBWD_STAGGER_SLOTS=$1 MAPF_RBWD_STAGGER_UNITS=$2 timeout -k 10 120 python tools/micro/recur_bwd_time.py 18 192 16
done 2>&1 | grep -v amdgpu.ids | tee -a $O/rbwd_stagger.txt

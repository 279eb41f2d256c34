#!/bin/bash
# time to the reference's stop criterion under different launch shapes of train.py (statistics every 10 s: finer stop resolution)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O
run() { tag=$1; shift; rm -rf models; t0=$(date +%s); timeout -k 10 420 python train.py --minutes 6.5 --interval 10 "$@" > $O/train_$tag.log 2> $O/train_$tag.err; echo "$tag rc=$? $(( $(date +%s) - t0 ))s  $(grep 'number of updates' $O/train_$tag.log | tail -1)  $(grep 'update speed' $O/train_$tag.log | tail -1)"; }
run envs512 --envs 512
run envs256 --envs 256
run envs1024_upi2 --envs 1024 --updates-per-iter 2

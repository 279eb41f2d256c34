#!/bin/bash
# time to the stop criterion against the statistics interval (promotions happen at statistics time)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v; mkdir -p $O
run() { tag=$1; shift; rm -rf models; t0=$(date +%s); timeout -k 10 400 python train.py --minutes 6 --envs 512 "$@" > $O/train_$tag.log 2> $O/train_$tag.err; echo "$tag rc=$? $(( $(date +%s) - t0 ))s  $(grep 'number of updates' $O/train_$tag.log | tail -1)  $(grep 'update speed' $O/train_$tag.log | tail -1)"; CK=models/$(ls -t models | head -1); timeout -k 10 100 python tools/eval_checkpoint.py $CK > $O/eval_$tag.txt 2>> $O/train_$tag.err; cat $O/eval_$tag.txt | cut -c1-120; }
run int5_seed0 --interval 5 --seed 0
run int10_seed1 --interval 10 --seed 1
run int5_seed2 --interval 5 --seed 2
run int2_seed0 --interval 2 --seed 0

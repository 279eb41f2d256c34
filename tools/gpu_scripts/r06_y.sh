#!/bin/bash
# `python train.py` with its defaults (1024 environments per level, --promote-interval 5) to the stop criterion
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
rm -rf models
t0=$(date +%s)
timeout -k 10 420 python train.py --minutes 6.5 --interval 20 > $O/train_default.log 2> $O/train_default.err; echo train_default=$? $(( $(date +%s) - t0 ))s
grep "stop criterion reached" $O/train_default.log; grep "update speed" $O/train_default.log | tail -1
CK=models/$(ls -t models | head -1)
timeout -k 10 100 python tools/eval_checkpoint.py $CK > $O/eval_default.txt 2>> $O/train_default.err; cut -c1-110 $O/eval_default.txt

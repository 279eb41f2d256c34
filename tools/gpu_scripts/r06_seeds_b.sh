#!/bin/bash
# round 6, final tree: the curriculum to the stop criterion with the promotion rule checked every 5 s (train.py --promote-interval, its
# default) and statistics every 20 s as in rounds 4-5, three seeds, each followed by its checkpoint on the three fixtures
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06seedsb; mkdir -p $O
for seed in ${SEEDS:-0 1 2}; do
rm -rf models
t0=$(date +%s)
timeout -k 10 400 python train.py --envs 512 --minutes 6 --interval 20 --seed $seed > $O/train_to_stop_seed$seed.log 2> $O/train_seed$seed.err; echo train_seed$seed=$? $(( $(date +%s) - t0 ))s
grep "stop criterion reached" $O/train_to_stop_seed$seed.log; grep "update speed" $O/train_to_stop_seed$seed.log | tail -1
CK=models/$(ls -t models | head -1)
timeout -k 10 100 python tools/eval_checkpoint.py $CK > $O/eval_seed$seed.txt 2>> $O/train_seed$seed.err; cut -c1-110 $O/eval_seed$seed.txt
done

#!/bin/bash
# round 6: the reference's whole curriculum to its own stop criterion (train.py), seeds 0 and 1, each followed by its final checkpoint on the
# reference's three evaluation fixtures (test.py:82-145)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06seeds; mkdir -p $O
for seed in ${SEEDS:-0 1}; do
rm -rf models
t0=$(date +%s)
timeout -k 10 520 python train.py --envs 512 --minutes 8 --interval 20 --seed $seed > $O/train_to_stop_seed$seed.log 2> $O/train_seed$seed.err; echo train_seed$seed=$? $(( $(date +%s) - t0 ))s
tail -3 $O/train_to_stop_seed$seed.log | head -1; grep -c "number of updates" $O/train_to_stop_seed$seed.log
CK=models/$(ls -t models | head -1)
timeout -k 10 200 python tools/eval_checkpoint.py $CK > $O/eval_seed$seed.txt 2>> $O/train_seed$seed.err; echo eval=$?
cat $O/eval_seed$seed.txt
done

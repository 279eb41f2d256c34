#!/bin/bash
# launch shape of train.py with the final tree: the curriculum to the stop criterion at 256 and 1024 (the default) environments per level
cd $GRAFT_REPO_ROOT
for e in 256 1024; do
rm -rf models
t0=$(date +%s)
timeout -k 10 560 python train.py --envs $e --minutes 9 --interval 20 > gpurun_out/r05v_train_to_stop_envs$e.log 2> gpurun_out/r05v_train_envs$e.err; echo "envs=$e train=$? $(( $(date +%s) - t0 ))s"
grep -c "number of updates" gpurun_out/r05v_train_to_stop_envs$e.log
tail -9 gpurun_out/r05v_train_to_stop_envs$e.log | head -3
done

# soak: train.py 5 minutes with the final code (graph-replayed actors + update), then evaluation of the checkpoint on the 16-agent fixture
cd $GRAFT_REPO_ROOT
rm -rf models
timeout -k 10 420 python train.py --envs 512 --minutes 5 > gpurun_out/r04_train_curriculum_5min_final.log 2> gpurun_out/r04_train_5min_final.err; echo train=$?
grep "update speed\|number of updates\|buffer update" gpurun_out/r04_train_curriculum_5min_final.log | tail -4
tail -3 gpurun_out/r04_train_5min_final.err
ls models | tail -3

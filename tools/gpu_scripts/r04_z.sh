# soak after the actor-path changes (input projection kernel, q_head, library warm-up): curriculum 5 minutes + fixed level (config 2) 1.5 minutes
cd $GRAFT_REPO_ROOT
rm -rf models
timeout -k 10 420 python train.py --envs 512 --minutes 5 > gpurun_out/r04_train_curriculum_5min_final2.log 2> gpurun_out/r04_train_5min_final2.err; echo train=$?
grep "update speed\|number of updates\|buffer update" gpurun_out/r04_train_curriculum_5min_final2.log | tail -3
tail -2 gpurun_out/r04_train_5min_final2.err
rm -rf models
timeout -k 10 200 python train.py --envs 4096 --agents 40 --map 32 --minutes 1.5 --learning-starts 20000 > gpurun_out/r04_train_c2_90s.log 2> gpurun_out/r04_train_c2_90s.err; echo train_c2=$?
grep "update speed\|number of updates\|buffer update" gpurun_out/r04_train_c2_90s.log | tail -3
tail -2 gpurun_out/r04_train_c2_90s.err

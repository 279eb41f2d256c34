# training-shape launches (round-4 review item 8): python train.py at --envs 512 (round 4's runs) -- wall time and updates to the reference's stop criterion
cd $GRAFT_REPO_ROOT
rm -rf models
timeout -k 10 1000 python train.py --envs 512 --minutes 12 --interval 20 > gpurun_out/r05_train_envs512_final.log 2> gpurun_out/r05_train_envs512_final.err; echo train=$?
grep -c "number of updates" gpurun_out/r05_train_envs512_final.log
grep "number of updates\|update speed\|buffer update speed" gpurun_out/r05_train_envs512_final.log | tail -6
tail -12 gpurun_out/r05_train_envs512_final.log
tail -2 gpurun_out/r05_train_envs512_final.err

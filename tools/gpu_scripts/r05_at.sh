# round 5: the reference's whole curriculum (train.py until its own stop criterion, final code of round 5), then the final checkpoint and a
# random-init network on the reference's three evaluation fixtures (test.py:82-145)
cd $GRAFT_REPO_ROOT
rm -rf models
timeout -k 10 800 python train.py --envs 512 --minutes 12 --interval 20 > gpurun_out/r05t_train_to_stop.log 2> gpurun_out/r05t_train_to_stop.err; echo train=$?
tail -12 gpurun_out/r05t_train_to_stop.log
ls -t models | head -3
CK=models/$(ls -t models | head -1)
timeout -k 10 300 python tools/eval_checkpoint.py --random-init $CK > gpurun_out/r05t_eval_after_curriculum.txt 2> gpurun_out/r05t_eval.err; echo eval=$?
cat gpurun_out/r05t_eval_after_curriculum.txt

cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/curriculum_iter.py 512 60 2>&1 | grep -v amdgpu | tee gpurun_out/r03_curriculum_iter.txt
timeout -k 10 200 python tools/curriculum_iter.py 1024 60 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03_curriculum_iter.txt
timeout -k 10 420 python train.py --envs 512 --minutes 5 --interval 20 --learning-starts 20000 > gpurun_out/r03_train_cur.log 2>&1; echo train=$?
grep "buffer update speed\|update speed" gpurun_out/r03_train_cur.log | tail -8; tail -12 gpurun_out/r03_train_cur.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b; timeout -k 10 420 python train.py --envs 512 --minutes 5 --interval 20 --learning-starts 20000 2>&1 | grep -v amdgpu > gpurun_out/r03b/train_curriculum_5min_tail.log; echo train=$?
tail -18 gpurun_out/r03b/train_curriculum_5min_tail.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout -k 10 1150 python -u train.py --envs 512 --minutes 18 --interval 60 --learning-starts 20000 2>&1 | grep --line-buffered -v amdgpu > gpurun_out/r03b/train_curriculum_to_end.log; echo train=$?
tail -16 gpurun_out/r03b/train_curriculum_to_end.log

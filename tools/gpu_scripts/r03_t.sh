cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -4
mkdir -p gpurun_out/r03b; timeout -k 10 420 python train.py --envs 512 --minutes 5 --interval 20 --learning-starts 20000 2>&1 | grep -v amdgpu > gpurun_out/r03b/train_curriculum_5min_nt1.log; echo train=$?
tail -22 gpurun_out/r03b/train_curriculum_5min_nt1.log

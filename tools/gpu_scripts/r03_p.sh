cd $GRAFT_REPO_ROOT
timeout -k 10 200 python train.py --agents 40 --map 32 --envs 4096 --minutes 1 --interval 15 --learning-starts 20000 2>&1 | grep -v amdgpu | tail -14
timeout -k 10 200 python train.py --agents 128 --map 64 --envs 2048 --minutes 0.7 --interval 15 --learning-starts 20000 --double-q 2>&1 | grep -v amdgpu | tail -8

cd $GRAFT_REPO_ROOT
for ov in 0 1; do echo "overlap $ov"; timeout -k 10 200 python -u train.py --agents 128 --map 64 --envs 2048 --minutes 0.8 --interval 15 --learning-starts 20000 --double-q --overlap-actors $ov 2>&1 | grep --line-buffered -v amdgpu | grep "update speed" | tail -2; done
for ov in 0 1; do echo "64 agents overlap $ov"; timeout -k 10 200 python -u train.py --agents 64 --map 40 --envs 2048 --minutes 0.6 --interval 12 --learning-starts 20000 --overlap-actors $ov 2>&1 | grep --line-buffered -v amdgpu | grep "update speed" | tail -2; done

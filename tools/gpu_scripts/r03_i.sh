cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_actor_gpu.py tests/test_curriculum_gpu.py tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python tools/actor_times.py --tape 2>&1 | grep "reuse="
timeout -k 10 300 python tools/actor_times.py 2>&1 | grep "reuse="
timeout -k 10 300 python tools/curriculum_iter.py 512 60 2>&1 | grep -v amdgpu

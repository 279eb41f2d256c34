cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_env_gpu.py tests/test_reset_gpu.py tests/test_actor_gpu.py -x -q -m gpu 2>&1 | tail -3
for x in 0 1; do echo "MAPF_STEP_XCHG=$x"; MAPF_STEP_XCHG=$x timeout -k 10 300 python tools/shape_sweep.py 4096,32,40 16384,32,40 4096,40,16 4096,64,40 2048,64,128 65536,20,6 2>&1 | grep -v amdgpu; done
bash tools/gpu_scripts/r03_k2.sh 2>&1 | grep "==\|G =\|per-block"

timeout 900 python -m pytest tests/test_env_gpu.py tests/test_actor_gpu.py -m gpu -x -q 2>&1 | tail -4 | grep -v amdgpu.ids | cut -c1-300
MAPF_STEP_THREADS=64 timeout 900 python -m pytest tests/test_env_gpu.py -m gpu -x -q 2>&1 | tail -2 | grep -v amdgpu.ids | cut -c1-300
timeout 300 python tools/stamps.py 2>&1 | grep -v amdgpu.ids | grep stamp
timeout 600 python tools/tune_step.py 2>&1 | grep -v amdgpu.ids | grep -E "impl=0 |impl=10 E= 4096"
timeout 600 python bench.py --no-cpu-baseline --no-dqn 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'])"

cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_env_gpu.py tests/test_actor_gpu.py -q -m gpu -x 2>&1 | tail -3
MAPF_STEP_NT=1 timeout -k 10 900 python -m pytest tests/test_env_gpu.py -q -m gpu -x 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline --no-dqn 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value',d['value'],'frac',r['frac'],'ooc',r['frac_out_of_cache'],r['kernel_avg_us'],r['kernel_avg_us_out_of_cache'])"

cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_entrypoints_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','learner_updates_per_sec','actor_loop_env_steps_per_sec','train_loop_updates_per_sec','train_loop_env_steps_per_sec','train_loop_ms_per_iter')})"

cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r02_j_tests.log 2>&1; echo tests=$?
tail -15 gpurun_out/r02_j_tests.log
TACT=20 timeout -k 10 200 python tools/profile_actor.py 2>&1 | tail -1
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r02_j_bench.json 2> gpurun_out/r02_j_bench.err; echo bench=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_j_bench.json'))
for k in ('value','ms_per_step','learner_updates_per_sec','learner_ms_per_update','actor_loop_env_steps_per_sec','actor_loop_ms_per_iter','train_loop_updates_per_sec','train_loop_ms_per_iter'): print(k, d.get(k))
PY

# round 4, last verification of the final tree: full GPU suite, smoke, bench line
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout -k 10 800 python -m pytest tests -q -m gpu -x > gpurun_out/r04e_gputests.log 2>&1; rc=$?; echo gputests=$rc $(( $(date +%s) - t0 ))s
tail -3 gpurun_out/r04e_gputests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py > gpurun_out/r04e_bench.json 2> gpurun_out/r04e_bench.err; echo bench=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04e_bench.json').read().strip().splitlines()[-1])
for k in ['value','ms_per_step','learner_ms_per_update','learner_updates_per_sec','learner_distinct_fraction','learner_reachable_fraction','actor_loop_ms_per_iter','actor_loop_tape_policy_ms_per_iter','actor_loop_tape_policy_env_steps_per_sec','train_loop_ms_per_iter','train_loop_updates_per_sec','train_loop_env_steps_per_sec']:
    print(k, d.get(k))
print('roofline', d['roofline']['frac'], d['roofline'].get('frac_out_of_cache'), d['roofline'].get('frac_hbm_proper'), 'cpu', d['cpu_baseline']['value'])
PY

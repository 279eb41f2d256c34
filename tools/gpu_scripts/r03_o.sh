cd $GRAFT_REPO_ROOT
t0=$(date +%s.%N); timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2; t1=$(date +%s.%N); echo "smoke wall $(echo "$t1 - $t0" | bc) s"
t0=$(date +%s.%N); timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo rc=$?; t1=$(date +%s.%N); echo "bench wall $(echo "$t1 - $t0" | bc) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","timed_repeats","timed_region_ms","learner_updates_per_sec","actor_loop_env_steps_per_sec","actor_loop_tape_policy_env_steps_per_sec","train_loop_updates_per_sec")})
print(d["roofline"]["frac"], d["roofline"]["frac_out_of_cache"], d["cpu_baseline"]["value"])
PY

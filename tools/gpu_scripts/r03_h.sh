cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b
mkdir -p $O
timeout -k 10 400 python tools/c5_bench.py > $O/c5_rates.txt 2>&1 && timeout -k 10 400 python tools/c5_bench.py --double-q >> $O/c5_rates.txt 2>&1 && timeout -k 10 400 python tools/c5_bench.py 64 40 2048 >> $O/c5_rates.txt 2>&1; echo c5=$?
grep -v amdgpu.ids $O/c5_rates.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err; echo bench=$?; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03b/bench_quick.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","learner_ms_per_update","learner_updates_per_sec","learner_ms_per_update_all_observations","actor_loop_ms_per_iter","actor_loop_tape_policy_ms_per_iter","train_loop_updates_per_sec","train_loop_env_steps_per_sec"):
    print(k, d.get(k))
print(d["roofline"]["frac"], d["roofline"].get("frac_out_of_cache"))
PY

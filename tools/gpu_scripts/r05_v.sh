cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r05_bench_v.json 2> $O/r05_bench_v.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/r05_bench_v.json") if l.startswith("{")][-1])
for k in ['value','ms_per_step','learner_ms_per_update','actor_loop_ms_per_iter','actor_loop_tape_policy_ms_per_iter','train_loop_ms_per_iter','pipeline_env_steps_per_sec','dqn_error']:
    print(k, d.get(k))
print('roofline', d['roofline']['frac'], d['roofline'].get('frac_out_of_cache'), d['roofline'].get('frac_hbm_proper'), 'cpu', d.get('cpu_baseline',{}).get('value'))
print('encoder', {k:v for k,v in d['encoder_roofline'].items() if k!='clock_note'})
PY
MAPF_BENCH_SHARE_GPU=1 MAPF_BENCH_WATCHDOG=280 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --no-out-of-cache > $O/r05_bench_2rank_shared_gpu.json 2> $O/r05_bench_2rank.err; echo bench2=$?
python - <<PY
import json
d=json.loads([l for l in open("$O/r05_bench_2rank_shared_gpu.json") if l.startswith("{")][-1])
for k in ['n_gpus','value','learner_ms_per_update','actor_loop_ms_per_iter','actor_loop_tape_policy_ms_per_iter','train_loop_ms_per_iter','dqn_error']:
    print(k, d.get(k))
PY
timeout -k 10 300 python tools/c5_bench.py --double-q 2>&1 | tail -1 > $O/r05_c5_rates.txt; cat $O/r05_c5_rates.txt

cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python tools/micro/tall_gemm_bench.py > $O/r05_tall_gemm_bench_3.txt 2>&1; echo "bench rc=$?"
tail -12 $O/r05_tall_gemm_bench_3.txt
for own in 1 0; do
MAPF_OWN_TALL_GEMM=$own timeout -k 10 400 python bench.py --no-out-of-cache --no-cpu-baseline --steps 20 --warmup 5 > $O/r05_bench_own$own.json 2> $O/r05_bench_own$own.err; echo "bench own=$own rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/r05_bench_own$own.json") if l.startswith("{")][-1])
print({k:round(v,3) for k,v in d.items() if k in ("learner_ms_per_update","train_loop_ms_per_iter","actor_loop_ms_per_iter","actor_loop_tape_policy_ms_per_iter","learner_ms_per_update_all_observations")}, d.get("dqn_error"))
PY
done

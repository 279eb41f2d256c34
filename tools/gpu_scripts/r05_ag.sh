cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_update_gpu.py tests/test_relevance_gpu.py tests/test_learner_gpu.py tests/test_big_goldens_gpu.py -x -q > gpurun_out/r05_ag_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r05_ag_tests.log
bash tools/gpu_scripts/r05_q.sh > gpurun_out/r05_q.out 2>&1
grep "plan_mark\|obs_dup\|wall" gpurun_out/r05_learner_iteration_breakdown.md | head -5
grep -n "plan_mark\|obs_dup\|span" gpurun_out/r05_update40_timeline.md | head -6
timeout -k 10 400 python bench.py --no-out-of-cache --no-cpu-baseline --steps 20 --warmup 5 --train-iters 30 --dqn-updates 30 > gpurun_out/r05_bench_ag.json 2> gpurun_out/r05_bench_ag.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r05_bench_ag.json") if l.startswith("{")][-1])
print({k:round(v,3) for k,v in d.items() if k in ("learner_ms_per_update","train_loop_ms_per_iter","actor_loop_ms_per_iter","actor_loop_tape_policy_ms_per_iter")}, d.get("dqn_error"))
PY

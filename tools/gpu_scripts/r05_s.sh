cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1100 python -m pytest tests/test_reset_gpu.py tests/test_actor_gpu.py tests/test_learner_gpu.py tests/test_replay_gpu.py tests/test_curriculum_gpu.py tests/test_entrypoints_gpu.py tests/test_eval_gpu.py -x -q > $O/r05_s_tests.log 2>&1; echo "tests rc=$?"
tail -12 $O/r05_s_tests.log
timeout -k 10 400 python bench.py --no-out-of-cache --no-cpu-baseline --steps 20 --warmup 5 --train-iters 30 --dqn-updates 30 > $O/r05_bench_s.json 2> $O/r05_bench_s.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/r05_bench_s.json") if l.startswith("{")][-1])
print({k:round(v,3) for k,v in d.items() if k in ("learner_ms_per_update","train_loop_ms_per_iter","actor_loop_ms_per_iter","actor_loop_tape_policy_ms_per_iter","actor_loop_every_row_ms_per_iter")}, d.get("dqn_error"))
PY
timeout -k 10 300 python -m pytest tests/test_big_goldens_gpu.py -x -q -s -k update_bf16 2>&1 | grep "^b40\|^b6\|^b128\|passed\|failed" > $O/r05_grad_errors_vs_reference.txt; cat $O/r05_grad_errors_vs_reference.txt | cut -c1-300

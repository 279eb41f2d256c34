cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_update_gpu.py tests/test_big_goldens_gpu.py -x -q > $O/r05_r_tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/r05_r_tests.log
bash tools/gpu_scripts/r05_q.sh > $O/r05_q.out 2>&1
head -12 $O/r05_learner_iteration_breakdown.md
grep -n "sum_bias\|sum_conv0\|adam_kernel\|span" $O/r05_update40_timeline.md
timeout -k 10 400 python bench.py --no-out-of-cache --no-cpu-baseline --steps 20 --warmup 5 --train-iters 30 --dqn-updates 30 > $O/r05_bench_r.json 2> $O/r05_bench_r.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/r05_bench_r.json") if l.startswith("{")][-1])
print({k:round(v,3) for k,v in d.items() if k in ("learner_ms_per_update","train_loop_ms_per_iter","actor_loop_ms_per_iter","actor_loop_tape_policy_ms_per_iter")}, d.get("dqn_error"))
PY

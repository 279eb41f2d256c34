cd $GRAFT_REPO_ROOT
O=gpurun_out
for t in 128 256 768 own0; do
if [ $t = own0 ]; then export MAPF_OWN_TALL_GEMM=0; else export MAPF_TALL_WGS=$t; fi
timeout -k 10 400 python bench.py --no-out-of-cache --no-cpu-baseline --steps 20 --warmup 5 --train-iters 30 --dqn-updates 30 > $O/r05_bench_t$t.json 2> $O/r05_bench_t$t.err; echo "bench t=$t rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/r05_bench_t$t.json") if l.startswith("{")][-1])
print({k:round(v,3) for k,v in d.items() if k in ("learner_ms_per_update","train_loop_ms_per_iter","actor_loop_ms_per_iter")}, d.get("dqn_error"))
PY
done

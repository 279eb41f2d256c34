cd $GRAFT_REPO_ROOT
O=gpurun_out
MAPF_BENCH_SHARE_GPU=1 MAPF_BENCH_WATCHDOG=280 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --no-out-of-cache > $O/r05_bench_2rank_torchrun.json 2> $O/r05_bench_2rank_torchrun.err; echo bench2=$?
python - <<PY
import json
d=json.loads([l for l in open("$O/r05_bench_2rank_torchrun.json") if l.startswith("{")][-1])
for k in ['n_gpus','value','learner_ms_per_update','actor_loop_ms_per_iter','actor_loop_tape_policy_ms_per_iter','train_loop_ms_per_iter','dqn_error']:
    print(k, d.get(k))
print(d['config']['parallelism'])
PY
tail -3 $O/r05_bench_2rank_torchrun.err

# end-of-round evidence (round 6), part B: full GPU suite, smoke, the bench line, the update / train-loop A/B times, 6-rank rehearsal
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
t0=$(date +%s)
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/gputests.log 2>&1; rc=$?; echo gputests=$rc $(( $(date +%s) - t0 ))s
tail -3 $O/gputests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
t0=$(date +%s)
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo bench=$? $(( $(date +%s) - t0 ))s
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06f/bench.json').read().strip().splitlines()[-1])
for k,v in d.items():
    if (isinstance(v,(int,float)) or v is None or k in ('dqn_error','learner_path','learner_ref_shape_path','cpu_baseline_error')): print(k, v)
print('roofline', {k: v for k, v in d['roofline'].items() if k.startswith('frac') or k.startswith('kernel_avg')}, 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['workers'])
print('encoder', {k: v for k, v in d['encoder_roofline'].items() if k != 'clock_note'})
PY
WARM=300 ITERS=200 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "prune=True" | tee $O/update6_times.txt
MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 200 2>&1 | grep "MODE=" | tee $O/train_loop.txt
t0=$(date +%s)
MAPF_BENCH_SHARE_GPU=1 timeout -k 10 600 python bench.py --gpus 6 --dist-backend gloo --envs 512 --steps 20 --warmup 5 > $O/bench_6rank.json 2> $O/bench_6rank.err; echo bench6=$? $(( $(date +%s) - t0 ))s; grep -c '^{' $O/bench_6rank.json

#!/bin/bash
# final tree: full GPU suite, smoke, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
t0=$(date +%s)
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/r05_ao_gputests.log 2>&1; rc=$?; echo gputests=$rc $(( $(date +%s) - t0 ))s
tail -3 $O/r05_ao_gputests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $O/bench_ao.json 2> $O/bench_ao.err; echo bench=$?
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_ao.json').read().strip().splitlines()[-1])
for k in ['value','ms_per_step','learner_ms_per_update','actor_loop_ms_per_iter','actor_weights_refresh_ms','actor_loop_tape_policy_ms_per_iter','train_loop_ms_per_iter','pipeline_env_steps_per_sec']:
    print(k,d.get(k))
print('frac',d['roofline']['frac'],'cpu',d['cpu_baseline']['value'])
P

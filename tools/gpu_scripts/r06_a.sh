#!/bin/bash
# round 6, first call: the new bench line on one rank, the bench entry-point tests (2 ranks, fault injection), a 6-rank shared-GPU rehearsal
cd $GRAFT_REPO_ROOT
O=gpurun_out
t0=$(date +%s)
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r06a_bench.json 2> $O/r06a_bench.err; echo bench=$? $(( $(date +%s) - t0 ))s
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a_bench.json').read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,(int,float)) or v is None or k in ('dqn_error','learner_path','learner_ref_shape_path'): print(k, v)
print('roofline', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['workers'])
PY
tail -5 $O/r06a_bench.err
t0=$(date +%s)
timeout -k 10 900 python -m pytest tests/test_entrypoints_gpu.py -q -x -k "bench" > $O/r06a_benchtests.log 2>&1; echo benchtests=$? $(( $(date +%s) - t0 ))s
tail -15 $O/r06a_benchtests.log

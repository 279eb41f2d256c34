#!/bin/bash
# bench with the snapshot refresh timed by itself
set -e
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_ak.json 2> gpurun_out/bench_ak.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_ak.json').read().strip().splitlines()[-1])
for k in ['value','actor_loop_ms_per_iter','actor_weights_refresh_ms','actor_loop_tape_policy_ms_per_iter','actor_loop_every_row_ms_per_iter','train_loop_ms_per_iter','learner_ms_per_update','pipeline_env_steps_per_sec']:
    print(k,d.get(k))
print(d['roofline']['frac'])
P

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: round(v,3) for k,v in d.items() if k in ('curriculum_actor_iter_ms','learner_ref_shape_ms_per_update','train_loop_ref_shape_ms_per_iter','learner_ms_per_update','train_loop_tape_policy_ms_per_iter','train_loop_ms_per_iter','actor_loop_tape_policy_ms_per_iter','actor_loop_ms_per_iter')}, d.get('dqn_error'))
PY
}
for ord in "learner_side,capture_actors,capture_learner,actors,actor_stage" "learner_side,actors,actor_stage,capture_actors,capture_learner" "capture_actors,capture_learner,learner_side,actors,actor_stage"; do
MAPF_STREAM_ORDER=$ord MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 100 2>&1 | grep "MODE=" | sed "s/^/ORDER=$ord /" | cut -c1-200 | tee -a $O/order.txt
MAPF_STREAM_ORDER=$ord timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-out-of-cache > $O/b.json 2>$O/b.err; show $O/b.json | tee -a $O/order.txt
done

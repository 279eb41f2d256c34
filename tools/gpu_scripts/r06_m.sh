#!/bin/bash
# hardware queues: the runtime multiplexes streams (and the branches of a replayed graph) onto GPU_MAX_HW_QUEUES queues (default 4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
for q in 4 8 16 2 8; do
GPU_MAX_HW_QUEUES=$q MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 100 2>&1 | grep "MODE=" | sed "s/^/GPU_MAX_HW_QUEUES=$q /" | tee -a $O/hwq.txt
done
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: round(v,3) for k,v in d.items() if k in ('value','curriculum_actor_iter_ms','learner_ref_shape_ms_per_update','train_loop_ref_shape_ms_per_iter','learner_ms_per_update','train_loop_tape_policy_ms_per_iter','train_loop_ms_per_iter','actor_loop_tape_policy_ms_per_iter','actor_loop_ms_per_iter')}, d['roofline']['frac'], d.get('dqn_error'))
PY
}
for q in 8 4 8; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-out-of-cache > $O/b_$q.json 2>$O/b_$q.err; echo "bench GPU_MAX_HW_QUEUES=$q"; show $O/b_$q.json
done | tee -a $O/hwq.txt

#!/bin/bash
# one stream per role (mapf_rl_amd/streams.py): tests, then bench.py's reference-shape train loop behind the config-2 legs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_learner_gpu.py tests/test_curriculum_gpu.py tests/test_actor_gpu.py tests/test_update_gpu.py tests/test_reset_gpu.py -q -x > $O/tests.log 2>&1; rc=$?; echo tests=$rc; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: round(v,3) for k,v in d.items() if k in ('curriculum_actor_iter_ms','learner_ref_shape_ms_per_update','train_loop_ref_shape_ms_per_iter','learner_ref_shape_graph_captures','train_loop_ref_shape_graph_captures','learner_ms_per_update','train_loop_tape_policy_ms_per_iter','train_loop_ms_per_iter','actor_loop_tape_policy_ms_per_iter')}, d.get('dqn_error'))
PY
}
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-out-of-cache > $O/b3.json 2>$O/b3.err; show $O/b3.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-out-of-cache --only-ref-shape > $O/b1.json 2>$O/b1.err; show $O/b1.json
MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 100 2>&1 | grep "MODE="

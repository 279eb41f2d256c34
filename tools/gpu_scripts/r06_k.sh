#!/bin/bash
# why does bench.py's train_loop_ref_shape (3.96 ms) not show the overlap tools/micro/train_loop_overlap.py measures (3.42 ms)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: round(v,3) for k,v in d.items() if k in ('curriculum_actor_iter_ms','learner_ref_shape_ms_per_update','train_loop_ref_shape_ms_per_iter','learner_ref_shape_graph_captures','train_loop_ref_shape_graph_captures','learner_ms_per_update')}, d.get('dqn_error'))
PY
}
MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 100 2>&1 | grep "MODE="
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-out-of-cache --only-ref-shape > $O/b1.json 2>$O/b1.err; show $O/b1.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-out-of-cache --only-ref-shape --ref-shape-updates 300 --ref-shape-warmup 200 > $O/b2.json 2>$O/b2.err; show $O/b2.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-out-of-cache > $O/b3.json 2>$O/b3.err; show $O/b3.json
MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 100 2>&1 | grep "MODE="

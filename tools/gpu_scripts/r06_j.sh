#!/bin/bash
# full GPU suite + smoke + bench + the 6-agent update's timeline (evidence for the reworked update)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
t0=$(date +%s)
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/gputests.log 2>&1; rc=$?; echo gputests=$rc $(( $(date +%s) - t0 ))s
tail -4 $O/gputests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo bench=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06j/bench.json').read().strip().splitlines()[-1])
for k,v in d.items():
    if (isinstance(v,(int,float)) or v is None or k in ('dqn_error','learner_path','learner_ref_shape_path')): print(k, v)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAGENTS=6 MAPLEN=20 NENVS=2048 TUPD=60 MAPF_UPDATE_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -- python3 $R/tools/profile_update.py > $R/$O/graph.log 2>&1; echo graph=$?
cd $R
python tools/update_timeline.py $O/prof_graph adam_kernel 400 > $O/update6_graph_timeline.md
rm -rf $O/prof_graph
head -30 $O/update6_graph_timeline.md

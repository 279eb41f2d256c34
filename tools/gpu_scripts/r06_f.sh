#!/bin/bash
# round 6: reordered stage replays (online forward behind the packs only), 240-workgroup weight-gradient launch, bucket-step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_update_gpu.py tests/test_learner_gpu.py -q -x > $O/tests.log 2>&1; rc=$?; echo tests=$rc; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for cfg in "1024 20" "1024 21" "256 20" "1024 20" "256 20"; do
set -- $cfg
MAPF_GRAPH_UROW_STEP=$1 MAPF_WGRAD_PARTS=$2 ITERS=200 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "graph=True  prune=True" | sed "s/^/urow_step=$1 parts=$2 /" | tee -a $O/update6_times.txt
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAGENTS=6 MAPLEN=20 NENVS=2048 TUPD=60 MAPF_UPDATE_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -- python3 $R/tools/profile_update.py > $R/$O/graph.log 2>&1; echo graph=$?
cd $R
python tools/update_timeline.py $O/prof_graph adam_kernel 400 > $O/update6_graph_timeline.md
rm -rf $O/prof_graph
cat $O/update6_graph_timeline.md

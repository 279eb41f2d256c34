#!/bin/bash
# timeline of one update at the reference's training shape (6 agents, 20 x 20), eager and graph-replayed
set -e
O=gpurun_out/r05an; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAGENTS=6 MAPLEN=20 NENVS=2048 TUPD=12 MAPF_UPDATE_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_eager -- python3 $R/tools/profile_update.py > $R/$O/eager.log 2>&1; echo eager=$?
NAGENTS=6 MAPLEN=20 NENVS=2048 TUPD=60 MAPF_UPDATE_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -- python3 $R/tools/profile_update.py > $R/$O/graph.log 2>&1; echo graph=$?
cd $R
python tools/update_timeline.py $O/prof_eager adam_kernel 400 > $O/update6_eager_timeline.md
python tools/update_timeline.py $O/prof_graph adam_kernel 400 > $O/update6_graph_timeline.md
rm -rf $O/prof_eager $O/prof_graph
head -3 $O/update6_graph_timeline.md; tail -2 $O/update6_graph_timeline.md

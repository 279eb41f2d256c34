#!/bin/bash
# round 6: per-update GPU timeline of the reworked graph-replayed 6-agent update
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAGENTS=6 MAPLEN=20 NENVS=2048 TUPD=60 MAPF_UPDATE_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -- python3 $R/tools/profile_update.py > $R/$O/graph.log 2>&1; echo graph=$?
cd $R
python tools/update_timeline.py $O/prof_graph adam_kernel 400 > $O/update6_graph_timeline.md
rm -rf $O/prof_graph
cat $O/update6_graph_timeline.md

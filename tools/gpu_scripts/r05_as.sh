#!/bin/bash
# several windows per recurrence tile at the reference's training shape: update times with and without, then the timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05as; mkdir -p $O
for v in 1 0 1 0; do
MAPF_TILE_WINDOWS=$v ITERS=200 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "prune=True" | sed "s/^/tiles=$v /"
done | tee $O/update6_times_tiles.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAGENTS=6 MAPLEN=20 NENVS=2048 TUPD=60 MAPF_UPDATE_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -- python3 $R/tools/profile_update.py > $R/$O/graph.log 2>&1; echo graph=$?
cd $R
python tools/update_timeline.py $O/prof_graph adam_kernel 400 > $O/update6_graph_timeline_tiles.md
rm -rf $O/prof_graph
head -1 $O/update6_graph_timeline_tiles.md; grep "recurrent" $O/update6_graph_timeline_tiles.md

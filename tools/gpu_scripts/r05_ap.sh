#!/bin/bash
# kernel breakdown of the greedy actor iteration at config 5's shape (2048 envs x 128 agents, 64 x 64)
set -e
O=gpurun_out/r05ap; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAGENTS=128 NENVS=2048 MAPLEN=64 TACT=20 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/tools/profile_actor.py > $R/$O/actor128.log 2>&1; echo rc=$?
cd $R
python tools/trace_breakdown.py $O/prof comm_mask_kernel 30 12 > $O/actor128_iteration_breakdown.md
rm -rf $O/prof
grep "actor loop" $O/actor128.log; head -24 $O/actor128_iteration_breakdown.md | cut -c1-150

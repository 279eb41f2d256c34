# round 4: timeline + launch count of the merged, graph-replayed curriculum iteration; train.py 5 minutes (graph update on / off)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_cur
MODES=graph timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/tools/curriculum_iter.py 512 60 > $R/gpurun_out/prof_cur.log 2>&1; rc=$?; echo prof=$rc
cd $R
python tools/update_timeline.py gpurun_out/prof_cur comm_mask_kernel 400 > gpurun_out/r04_curriculum_iteration_timeline.md
find gpurun_out/prof_cur -name "*.csv" -size +1M -delete
head -2 gpurun_out/r04_curriculum_iteration_timeline.md
rm -rf models
timeout -k 10 400 python train.py --envs 512 --minutes 5 > gpurun_out/r04_train_curriculum_5min_graph.log 2> gpurun_out/r04_train_5min_graph.err; echo train_graph=$?
grep "update speed\|number of updates\|buffer update" gpurun_out/r04_train_curriculum_5min_graph.log | tail -6
rm -rf models
MAPF_UPDATE_GRAPH=0 timeout -k 10 400 python train.py --envs 512 --minutes 5 > gpurun_out/r04_train_curriculum_5min_nograph.log 2> gpurun_out/r04_train_5min_nograph.err; echo train_nograph=$?
grep "update speed\|number of updates\|buffer update" gpurun_out/r04_train_curriculum_5min_nograph.log | tail -6

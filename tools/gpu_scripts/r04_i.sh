# round 4: GPU timeline of one graph-replayed curriculum actor iteration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_cur
MODES=graph timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/tools/curriculum_iter.py 512 60 > $R/gpurun_out/prof_cur.log 2>&1; rc=$?; echo prof=$rc
cd $R
python tools/update_timeline.py gpurun_out/prof_cur obs_changed_kernel 400 > gpurun_out/r04_curriculum_iteration_timeline.md
find gpurun_out/prof_cur -name "*.csv" -size +1M -delete
head -3 gpurun_out/r04_curriculum_iteration_timeline.md; tail -2 gpurun_out/r04_curriculum_iteration_timeline.md; tail -3 gpurun_out/prof_cur.log

# learner kernel stats / per-iteration breakdown / one-update timeline at config 2 (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_learner
TUPD=8 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner.log 2>&1; echo learner=$?
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_learner learner > gpurun_out/r05_learner_kernel_stats.md
python tools/trace_breakdown.py gpurun_out/prof_learner encoder_bwd_kernel 30 > gpurun_out/r05_learner_iteration_breakdown.md
python tools/update_timeline.py gpurun_out/prof_learner adam_kernel 400 > gpurun_out/r05_update40_timeline.md
find gpurun_out/prof_learner -name "*kernel_trace.csv" -delete
cat gpurun_out/r05_learner_iteration_breakdown.md | head -50
tail -3 gpurun_out/prof_learner.log

# kernel breakdown of the learner update (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_learner
TUPD=6 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner.log 2>&1; echo learner=$?
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_learner learner > gpurun_out/prof_learner.md
python tools/trace_breakdown.py gpurun_out/prof_learner encoder_bwd_kernel 30 > gpurun_out/prof_learner_iter.md
python tools/trace_breakdown.py gpurun_out/prof_learner encoder_bwd_kernel 30 60 > gpurun_out/prof_learner_big.md
find gpurun_out/prof_learner -name "*kernel_trace.csv" -delete
cat gpurun_out/prof_learner_iter.md
tail -5 gpurun_out/prof_learner.log

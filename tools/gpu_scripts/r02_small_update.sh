# learner update at the reference's own agent count (6): kernel trace of one steady-state iteration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_learner6
NAGENTS=6 TUPD=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner6 -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner6.log 2>&1; echo learner6=$?
cd $R
python tools/trace_breakdown.py gpurun_out/prof_learner6 encoder_bwd_kernel 40 12 > gpurun_out/prof_learner6_iter.md
find gpurun_out/prof_learner6 -name "*.csv" -size +1M -delete
cat gpurun_out/prof_learner6_iter.md

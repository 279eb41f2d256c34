# learner update with the unreachable observations pruned: kernel breakdown of one steady-state iteration (config 2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_learner
TUPD=8 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner.log 2>&1; echo learner=$?
cd $R
python tools/trace_breakdown.py gpurun_out/prof_learner encoder_bwd_kernel 30 24 > gpurun_out/prof_learner_iter.md
find gpurun_out/prof_learner -name "*.csv" -size +1M -delete
cat gpurun_out/prof_learner_iter.md

# the largest non-hand-written launches of one learner update (config 2), in launch order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_learner
TUPD=6 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner.log 2>&1; echo learner=$?
cd $R
python tools/trace_breakdown.py gpurun_out/prof_learner encoder_bwd_kernel 6 60 | tail -64
find gpurun_out/prof_learner -name "*.csv" -size +1M -delete

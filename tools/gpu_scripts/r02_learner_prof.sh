# learner update breakdowns at config 2 (40 agents) and at config 5's agent count (128)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_learner $R/gpurun_out/prof_learner128
TUPD=6 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner.log 2>&1; echo learner=$?
NAGENTS=128 TUPD=4 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner128 -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner128.log 2>&1; echo learner128=$?
cd $R
python tools/trace_breakdown.py gpurun_out/prof_learner encoder_bwd_kernel 30 > gpurun_out/prof_learner_iter.md
python tools/trace_breakdown.py gpurun_out/prof_learner128 encoder_bwd_kernel 30 > gpurun_out/prof_learner128_iter.md
find gpurun_out/prof_learner gpurun_out/prof_learner128 -name "*.csv" -size +1M -delete
head -14 gpurun_out/prof_learner_iter.md; head -14 gpurun_out/prof_learner128_iter.md

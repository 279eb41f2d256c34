cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O; rm -rf $O/prof_learner6
NAGENTS=6 MAPLEN=20 NENVS=2048 TUPD=8 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner6 -- python3 $R/tools/profile_update.py > $O/prof_learner6.log 2>&1; echo learner6=$?
cd $R
python tools/trace_breakdown.py $O/prof_learner6 encoder_bwd_kernel 40 10 > $O/learner6_iteration_breakdown.md
find $O/prof_learner6 -name "*.csv" -size +1M -delete
cat $O/learner6_iteration_breakdown.md

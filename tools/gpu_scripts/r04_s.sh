# config 5's shape (64x64, 128 agents): learner update and actor iteration by kernel category
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04s
rm -rf $O && mkdir -p $O
NAGENTS=128 TUPD=6 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner -- python3 $R/tools/profile_update.py > $O/prof_learner.log 2>&1; echo learner=$?
NAGENTS=128 TACT=6 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_actor -- python3 $R/tools/profile_actor.py > $O/prof_actor.log 2>&1; echo actor=$?
cd $R
python tools/trace_breakdown.py $O/prof_learner encoder_bwd_kernel 30 > $O/c5_learner_iteration_breakdown.md
python tools/trace_breakdown.py $O/prof_actor comm_mask_kernel 30 12 > $O/c5_actor_iteration_breakdown.md
find $O -name "*.csv" -size +1M -delete
head -22 $O/c5_learner_iteration_breakdown.md; head -14 $O/c5_actor_iteration_breakdown.md

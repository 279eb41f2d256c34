# end-of-round evidence (round 4), final code: bench.py kernel stats (the command the bench line comes from), learner / actor
# breakdowns, the 6-agent update timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04d
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py > $O/prof_bench.json 2> $O/prof_bench.err; echo bench=$?
TUPD=8 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner -- python3 $R/tools/profile_update.py > $O/prof_learner.log 2>&1; echo learner=$?
TACT=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_actor -- python3 $R/tools/profile_actor.py > $O/prof_actor.log 2>&1; echo actor=$?
cd $R
python tools/summarize_rocprof.py $O/prof_bench bench env_step_kernel > $O/bench_kernel_stats.md
python tools/summarize_rocprof.py $O/prof_learner learner > $O/learner_kernel_stats.md
python tools/summarize_rocprof.py $O/prof_actor actor > $O/actor_kernel_stats.md
python tools/trace_breakdown.py $O/prof_learner encoder_bwd_kernel 30 > $O/learner_iteration_breakdown.md
python tools/trace_breakdown.py $O/prof_actor comm_mask_kernel 30 12 > $O/actor_iteration_breakdown.md
find $O -name "*.csv" -size +1M -delete
head -14 $O/bench_kernel_stats.md; head -24 $O/learner_iteration_breakdown.md; head -16 $O/actor_iteration_breakdown.md
tail -c 300 $O/prof_bench.json

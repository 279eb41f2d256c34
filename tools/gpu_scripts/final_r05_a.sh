# end-of-round evidence (round 5), part A: bench.py kernel stats (the command the bench line comes from), HBM traffic counters of the
# env-step kernel (separate --pmc passes) at 4096 environments, learner / actor kernel stats, per-iteration breakdowns and timelines
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py > $O/prof_bench.json 2> $O/prof_bench.err; echo bench=$?
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_4096 -- python3 $R/bench.py --no-cpu-baseline --no-dqn --no-out-of-cache --steps 20 --warmup 5 --envs 4096 > $O/pmc_${c}_4096.log 2>&1; echo pmc_${c}=$?
done
TUPD=8 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner -- python3 $R/tools/profile_update.py > $O/prof_learner.log 2>&1; echo learner=$?
TACT=40 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_actor -- python3 $R/tools/profile_actor.py > $O/prof_actor.log 2>&1; echo actor=$?
cd $R
python tools/summarize_rocprof.py $O/prof_bench bench env_step_kernel > $O/bench_kernel_stats.md
python tools/summarize_rocprof.py $O/prof_learner learner > $O/learner_kernel_stats.md
python tools/trace_breakdown.py $O/prof_learner encoder_bwd_kernel 30 > $O/learner_iteration_breakdown.md
python tools/update_timeline.py $O/prof_learner adam_kernel 400 > $O/update40_timeline.md
python tools/summarize_rocprof.py $O/prof_actor actor > $O/actor_kernel_stats.md
python tools/trace_breakdown.py $O/prof_actor comm_mask_kernel 30 12 > $O/actor_iteration_breakdown.md
python tools/update_timeline.py $O/prof_actor comm_mask_kernel 100 > $O/actor_iteration_timeline.md
for c in FETCH_SIZE WRITE_SIZE; do
python tools/pmc_summary.py $O/pmc_${c}_4096 "env_step_kernel<unsigned int, 4, true" > $O/pmc_${c}_4096.txt 2>&1
done
find $O -name "*.csv" -size +1M -delete
cat $O/pmc_FETCH_SIZE_4096.txt $O/pmc_WRITE_SIZE_4096.txt
head -14 $O/bench_kernel_stats.md | cut -c1-160; head -24 $O/learner_iteration_breakdown.md; head -16 $O/actor_iteration_breakdown.md; cat $O/actor_iteration_timeline.md | cut -c1-100
tail -c 600 $O/prof_bench.json

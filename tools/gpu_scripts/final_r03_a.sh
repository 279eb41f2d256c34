# end-of-round evidence, part A: bench.py kernel stats (the same command the bench line comes from), HBM traffic counters of the
# env-step kernel (separate --pmc passes) at the BASELINE size and at E = 16384 (working set beyond the Infinity Cache), actor /
# learner per-iteration breakdowns, the 128-agent (config 5) learner
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py > $O/prof_bench.json 2> $O/prof_bench.err; echo bench=$?
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --no-dqn --no-out-of-cache --steps 20 --warmup 5 > $O/pmc_$c.log 2>&1; echo pmc_$c=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_16k -- python3 $R/bench.py --no-cpu-baseline --no-dqn --no-out-of-cache --steps 20 --warmup 5 --envs 16384 > $O/pmc_${c}_16k.log 2>&1; echo pmc_${c}_16k=$?
done
TACT=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_actor -- python3 $R/tools/profile_actor.py > $O/prof_actor.log 2>&1; echo actor=$?
TUPD=8 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner -- python3 $R/tools/profile_update.py > $O/prof_learner.log 2>&1; echo learner=$?
PRUNE=0 TUPD=6 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner_all -- python3 $R/tools/profile_update.py > $O/prof_learner_all.log 2>&1; echo learner_all=$?
NAGENTS=128 TUPD=4 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner128 -- python3 $R/tools/profile_update.py > $O/prof_learner128.log 2>&1; echo learner128=$?
cd $R
python tools/summarize_rocprof.py $O/prof_bench bench env_step_kernel > $O/bench_kernel_stats.md
python tools/summarize_rocprof.py $O/prof_actor actor > $O/actor_kernel_stats.md
python tools/summarize_rocprof.py $O/prof_learner learner > $O/learner_kernel_stats.md
python tools/summarize_rocprof.py $O/prof_learner_all learner_all > $O/learner_all_kernel_stats.md
python tools/trace_breakdown.py $O/prof_actor comm_mask_kernel 30 12 > $O/actor_iteration_breakdown.md
python tools/trace_breakdown.py $O/prof_learner encoder_bwd_kernel 30 > $O/learner_iteration_breakdown.md
python tools/trace_breakdown.py $O/prof_learner_all encoder_bwd_kernel 30 > $O/learner_all_iteration_breakdown.md
python tools/trace_breakdown.py $O/prof_learner128 encoder_bwd_kernel 30 > $O/c5_learner_iteration_breakdown.md
for c in FETCH_SIZE WRITE_SIZE; do
python tools/pmc_summary.py $O/pmc_$c "env_step_kernel<unsigned int, 4, true" > $O/pmc_$c.txt 2>&1
python tools/pmc_summary.py $O/pmc_${c}_16k "env_step_kernel<unsigned int, 4, true" > $O/pmc_${c}_16k.txt 2>&1
done
find $O -name "*.csv" -size +1M -delete
cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_FETCH_SIZE_16k.txt $O/pmc_WRITE_SIZE_16k.txt
head -14 $O/bench_kernel_stats.md; head -24 $O/learner_iteration_breakdown.md; head -8 $O/learner_all_iteration_breakdown.md; head -20 $O/actor_iteration_breakdown.md; head -8 $O/c5_learner_iteration_breakdown.md
grep "actor loop\|update" $O/prof_actor.log $O/prof_learner.log | head
tail -c 2500 $O/prof_bench.json

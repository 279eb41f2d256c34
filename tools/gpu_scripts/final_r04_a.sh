# end-of-round evidence (round 4), part A: bench.py kernel stats (the command the bench line comes from), HBM traffic counters of the
# env-step kernel (separate --pmc passes) at 4096 / 16,384 / 32,768 environments, learner / actor breakdowns, config-5 rates, merged sweep
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py > $O/prof_bench.json 2> $O/prof_bench.err; echo bench=$?
for c in FETCH_SIZE WRITE_SIZE; do
for e in 4096 16384 32768; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$e -- python3 $R/bench.py --no-cpu-baseline --no-dqn --no-out-of-cache --steps 20 --warmup 5 --envs $e > $O/pmc_${c}_$e.log 2>&1; echo pmc_${c}_$e=$?
done
done
TUPD=8 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner -- python3 $R/tools/profile_update.py > $O/prof_learner.log 2>&1; echo learner=$?
TACT=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_actor -- python3 $R/tools/profile_actor.py > $O/prof_actor.log 2>&1; echo actor=$?
cd $R
python tools/summarize_rocprof.py $O/prof_bench bench env_step_kernel > $O/bench_kernel_stats.md
python tools/summarize_rocprof.py $O/prof_learner learner > $O/learner_kernel_stats.md
python tools/trace_breakdown.py $O/prof_learner encoder_bwd_kernel 30 > $O/learner_iteration_breakdown.md
python tools/trace_breakdown.py $O/prof_actor comm_mask_kernel 30 12 > $O/actor_iteration_breakdown.md
for c in FETCH_SIZE WRITE_SIZE; do
for e in 4096 16384 32768; do
python tools/pmc_summary.py $O/pmc_${c}_$e "env_step_kernel<unsigned int, 4, true" > $O/pmc_${c}_$e.txt 2>&1
done
done
find $O -name "*.csv" -size +1M -delete
for e in 4096 16384 32768; do echo "== $e environments"; cat $O/pmc_FETCH_SIZE_$e.txt $O/pmc_WRITE_SIZE_$e.txt; done
head -14 $O/bench_kernel_stats.md; head -24 $O/learner_iteration_breakdown.md; head -16 $O/actor_iteration_breakdown.md
timeout -k 10 300 python tools/multi_sweep.py 512 2048 8192 32768 > gpurun_out/r04_multi_sweep.md 2> gpurun_out/r04_multi_sweep.err; echo msweep=$?; cat gpurun_out/r04_multi_sweep.md
timeout -k 10 300 python tools/c5_bench.py > gpurun_out/r04_c5_rates.txt 2>&1; echo c5=$?; tail -6 gpurun_out/r04_c5_rates.txt
timeout -k 10 300 python tools/update_times.py 6 20 2048 > gpurun_out/r04_update_times_6.log 2>&1; tail -4 gpurun_out/r04_update_times_6.log

# end-of-round evidence: bench.py kernel stats (the same command the bench line comes from), HBM traffic counters of the
# env-step kernel (separate --pmc passes), actor-loop and learner-update breakdowns
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/pmc_*
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err; echo bench=$?
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --no-dqn --steps 20 --warmup 5 > $R/gpurun_out/pmc_$c.log 2>&1; echo pmc_$c=$?
done
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_bench bench > gpurun_out/prof_bench.md
find gpurun_out/prof_bench -name "*kernel_trace.csv" -delete
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE "env_step_kernel<unsigned int, 4, true" > gpurun_out/pmc_fetch.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_WRITE_SIZE "env_step_kernel<unsigned int, 4, true" > gpurun_out/pmc_write.txt 2>&1
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +2M -delete
cat gpurun_out/pmc_fetch.txt gpurun_out/pmc_write.txt | tail -6
bash tools/gpu_scripts/prof_dqn.sh > /dev/null 2>&1; echo dqn=$?
head -12 gpurun_out/prof_actor_iter.md; head -12 gpurun_out/prof_learner_iter.md
tail -c 1500 gpurun_out/prof_bench.json

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof5 $R/gpurun_out/pmc5_*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5 -- python3 $R/bench.py --no-cpu-baseline --no-dqn > $R/gpurun_out/prof5.log 2>&1; echo prof=$?
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc5_$c -- python3 $R/bench.py --no-cpu-baseline --no-dqn --steps 20 --warmup 5 > $R/gpurun_out/pmc5_$c.log 2>&1; echo pmc_$c=$?
done
cd $R; python tools/pmc_summary.py gpurun_out/pmc5_FETCH_SIZE "env_step_kernel<unsigned int, 4, true"; python tools/pmc_summary.py gpurun_out/pmc5_WRITE_SIZE "env_step_kernel<unsigned int, 4, true"
python tools/summarize_rocprof.py gpurun_out/prof5 x | head -8 | cut -c1-150
tail -2 gpurun_out/prof5.log | cut -c1-400

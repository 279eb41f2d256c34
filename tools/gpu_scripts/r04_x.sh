# HBM traffic counters of the merged multi-level env step (env_step_multi_kernel) at 7 levels x 32,768 and x 8,192 environments: separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04x
rm -rf $O && mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
for e in 8192 32768; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$e -- python3 $R/tools/multi_sweep.py $e > $O/pmc_${c}_$e.log 2>&1; echo pmc_${c}_$e=$?
done
done
cd $R
for e in 8192 32768; do echo "== 7 levels x $e environments"; for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py $O/pmc_${c}_$e "env_step_multi_kernel"; done; done > gpurun_out/r04_multi_pmc.txt
find $O -name "*.csv" -size +1M -delete
cat gpurun_out/r04_multi_pmc.txt; tail -3 $O/pmc_FETCH_SIZE_32768.log

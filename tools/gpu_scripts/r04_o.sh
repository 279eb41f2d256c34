cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 32768; do
rm -rf $R/gpurun_out/prof_multi_$e
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_multi_$e -- python3 $R/tools/multi_sweep.py $e > $R/gpurun_out/prof_multi_$e.log 2>&1; echo multi_$e=$?
done
cd $R
grep "env_step_multi_kernel" gpurun_out/prof_multi_32768/*/*kernel_stats.csv | cut -c1-200
find gpurun_out/prof_multi_* -name "*.csv" -size +1M -delete
timeout -k 10 300 python -m pytest tests/test_env_gpu.py -x -q -m gpu -k "multi" 2>&1 | tail -3

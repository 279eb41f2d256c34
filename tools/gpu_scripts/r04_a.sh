# round 4, first GPU call: the new env launch-family tests, the bench line with both beyond-cache points, shape sweep under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python -m pytest tests/test_env_gpu.py -x -q -m gpu > gpurun_out/r04_a_envtests.log 2>&1; echo envtests=$?; tail -3 gpurun_out/r04_a_envtests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_a_bench.json 2> gpurun_out/r04_a_bench.err; echo bench=$?
cd /tmp
rm -rf $R/gpurun_out/r04_sweep
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_sweep -- python3 $R/tools/shape_sweep.py > $R/gpurun_out/r04_sweep.log 2> $R/gpurun_out/r04_sweep.err; echo sweep=$?
cd $R
python3 tools/shape_sweep.py --summarize gpurun_out/r04_sweep gpurun_out/r04_sweep.log > gpurun_out/r04_sweep.md 2>> gpurun_out/r04_sweep.err
find gpurun_out/r04_sweep -name "*.csv" -size +1M -delete
cat gpurun_out/r04_sweep.md

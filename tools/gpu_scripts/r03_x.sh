cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/r03_sweep
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_sweep -- python3 $R/tools/shape_sweep.py > $R/gpurun_out/r03_sweep.log 2> $R/gpurun_out/r03_sweep.err; echo sweep=$?
cd $R
python3 tools/shape_sweep.py --summarize gpurun_out/r03_sweep gpurun_out/r03_sweep.log > gpurun_out/r03_sweep.md 2>> gpurun_out/r03_sweep.err
find gpurun_out/r03_sweep -name "*.csv" -size +1M -delete
cat gpurun_out/r03_sweep.md

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
mkdir -p $O; rm -rf $O/sweep
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/sweep -- python3 $R/tools/shape_sweep.py > $O/sweep.log 2> $O/sweep.err; echo sweep=$?
cd $R
python3 tools/shape_sweep.py --summarize $O/sweep $O/sweep.log > $O/shape_sweep.md 2>> $O/sweep.err
find $O/sweep -name "*.csv" -size +1M -delete
cat $O/shape_sweep.md
timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3

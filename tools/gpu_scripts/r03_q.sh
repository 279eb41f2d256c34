cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ab in 0 4 8 32 0; do
rm -rf $R/gpurun_out/r03_q
MAPF_STEP_ABLATE=$ab timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_q -- python3 $R/tools/shape_sweep.py 4096,32,40 4096,16,40 16384,32,40 > $R/gpurun_out/r03_q.log 2> $R/gpurun_out/r03_q.err
(cd $R && echo "ablate=$ab" && python3 tools/shape_sweep.py --summarize gpurun_out/r03_q gpurun_out/r03_q.log 2>/dev/null | cut -d'|' -f2-4,7-8,11 | tail -3; find gpurun_out/r03_q -name "*.csv" -size +1M -delete)
done

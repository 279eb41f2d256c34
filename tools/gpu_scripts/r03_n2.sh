cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ab in 0 8; do
rm -rf $R/gpurun_out/r03_ab$ab
MAPF_STEP_ABLATE=$ab timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_ab$ab -- python3 $R/tools/shape_sweep.py 4096,32,40 8192,32,40 16384,32,40 32768,32,40 65536,20,6 262144,10,1 > $R/gpurun_out/r03_ab$ab.log 2> $R/gpurun_out/r03_ab$ab.err
(cd $R && echo "MAPF_STEP_ABLATE=$ab" && python3 tools/shape_sweep.py --summarize gpurun_out/r03_ab$ab gpurun_out/r03_ab$ab.log 2>/dev/null | cut -d'|' -f2-5,7-8,11 | tail -6; find gpurun_out/r03_ab$ab -name "*.csv" -size +1M -delete)
done

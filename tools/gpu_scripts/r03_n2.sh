cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nt in 128 256; do
rm -rf $R/gpurun_out/r03_nt$nt
MAPF_STEP_THREADS=$nt timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_nt$nt -- python3 $R/tools/shape_sweep.py 6144,32,40 8192,32,40 12288,32,40 16384,32,40 32768,32,40 > $R/gpurun_out/r03_nt$nt.log 2> $R/gpurun_out/r03_nt$nt.err
(cd $R && echo "MAPF_STEP_THREADS=$nt" && python3 tools/shape_sweep.py --summarize gpurun_out/r03_nt$nt gpurun_out/r03_nt$nt.log 2>/dev/null | cut -d'|' -f2-5,7-8,11 | tail -5; find gpurun_out/r03_nt$nt -name "*.csv" -size +1M -delete)
done

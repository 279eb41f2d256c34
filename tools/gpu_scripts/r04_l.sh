cd $GRAFT_REPO_ROOT
MAPF_UPDATE_GRAPH=0 MODE=update timeout -k 10 200 python -X faulthandler tools/micro/graph_gemm_probe.py > gpurun_out/r04_l_probe.log 2>&1; rc=$?; echo probe=$rc; grep -v "Extension" gpurun_out/r04_l_probe.log | tail -6
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_l_probe.log; then exit 1; fi

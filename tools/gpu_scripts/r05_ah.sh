cd $GRAFT_REPO_ROOT
python tools/micro/recur_bwd_trace.py run 32 192 16 2>&1 | grep -v amdgpu > gpurun_out/r05_recur_bwd_trace_32.txt; head -45 gpurun_out/r05_recur_bwd_trace_32.txt

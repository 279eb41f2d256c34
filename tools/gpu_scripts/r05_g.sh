cd $GRAFT_REPO_ROOT
O=gpurun_out
python tools/micro/recur_trace.py run 40 192 > $O/r05_recur_trace_40_192.txt 2>&1; echo rc=$?
python tools/micro/recur_trace.py run 6 192 > $O/r05_recur_trace_6_192.txt 2>&1; echo rc=$?
python tools/micro/recur_trace.py run 6 4096 > $O/r05_recur_trace_6_4096.txt 2>&1; echo rc=$?
cat $O/r05_recur_trace_6_192.txt

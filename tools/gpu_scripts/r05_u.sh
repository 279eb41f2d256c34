cd $GRAFT_REPO_ROOT
O=gpurun_out
t0=$(date +%s)
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $O/r05_u_gputests.log 2>&1; rc=$?; echo gputests=$rc $(( $(date +%s) - t0 ))s
tail -8 $O/r05_u_gputests.log

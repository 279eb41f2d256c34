cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python tools/micro/recur_multi.py run 1x4096x40 18x192x40 18x192x6 > $O/r05_recur_w16.txt 2>&1; echo "multi rc=$?"
grep -v "max |dh|" $O/r05_recur_w16.txt

cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python tools/micro/recur_multi.py run 1x4096x40 18x192x40 18x192x6 18x192x24 > $O/r05_recur_stagger.txt 2>&1; echo "multi rc=$?"
cat $O/r05_recur_stagger.txt

cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  WATCHDOG=30 timeout -k 5 60 python tools/hang_repro.py > gpurun_out/hm.log 2>&1; rc=$?
  echo "try $i rc=$rc $(grep 'updates in' gpurun_out/hm.log | head -1)"
done

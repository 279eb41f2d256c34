cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python tools/micro/recur_multi.py run 1x4096x40 18x192x40 18x192x6 1x4096x6 1x1400x16 1x5000x40 > $O/r05_recur_multi_2.txt 2>&1; echo "multi rc=$?"
cat $O/r05_recur_multi_2.txt
MAPF_RECUR_PERSIST=0 timeout -k 10 300 python tools/micro/recur_multi.py run 1x4096x40 1x4096x6 > $O/r05_recur_multi_2_nopersist.txt 2>&1; echo "multi rc=$?"
cat $O/r05_recur_multi_2_nopersist.txt
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_big_goldens_gpu.py tests/test_actor_gpu.py -x -q > $O/r05_h_tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/r05_h_tests.log

cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python tools/micro/recur_multi.py run > $O/r05_recur_multi_1.txt 2>&1; echo "multi rc=$?"
cat $O/r05_recur_multi_1.txt
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_big_goldens_gpu.py -x -q > $O/r05_e_tests.log 2>&1; echo "tests rc=$?"
tail -8 $O/r05_e_tests.log

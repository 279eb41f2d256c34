cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python tools/micro/tall_gemm_bench.py > $O/r05_tall_gemm_bench.txt 2>&1; echo "bench rc=$?"
cat $O/r05_tall_gemm_bench.txt
timeout -k 10 1100 python -m pytest tests/test_update_gpu.py tests/test_learner_gpu.py tests/test_big_goldens_gpu.py tests/test_gemm_gpu.py -x -q > $O/r05_k_tests.log 2>&1; echo "tests rc=$?"
tail -15 $O/r05_k_tests.log

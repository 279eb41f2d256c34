cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python tools/micro/tall_gemm_bench.py > $O/r05_tall_gemm_bench_2.txt 2>&1; echo "bench rc=$?"
cat $O/r05_tall_gemm_bench_2.txt
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q > $O/r05_l_tests.log 2>&1; echo "tests rc=$?"
tail -8 $O/r05_l_tests.log

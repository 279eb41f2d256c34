cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q > $O/r05_j_tests.log 2>&1; echo "tests rc=$?"
tail -25 $O/r05_j_tests.log

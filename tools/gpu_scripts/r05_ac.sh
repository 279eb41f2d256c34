cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py tests/test_update_gpu.py tests/test_big_goldens_gpu.py tests/test_learner_gpu.py -x -q > gpurun_out/r05_ac_tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r05_ac_tests.log

cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_reset_gpu.py -x -q > gpurun_out/r05_af_tests.log 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r05_af_tests.log

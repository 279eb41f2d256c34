cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r03_w_tests.log 2>&1; echo tests=$?; tail -15 gpurun_out/r03_w_tests.log

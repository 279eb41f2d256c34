cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_learner_gpu.py tests/test_entrypoints_gpu.py tests/test_curriculum_gpu.py -x -q > gpurun_out/r05_ab_tests.log 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r05_ab_tests.log

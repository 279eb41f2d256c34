cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_curriculum_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r05_aa_tests.log 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r05_aa_tests.log

cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_big_goldens_gpu.py tests/test_env_gpu.py tests/test_entrypoints_gpu.py -q -m gpu > gpurun_out/r02_c_tests.log 2>&1; echo tests=$?
tail -40 gpurun_out/r02_c_tests.log
for g in 1 2 4 8; do echo "group cap $g"; MAPF_STEP_GROUP=$g timeout -k 10 200 python tools/shape_sweep.py 8192,20,6 16384,20,6 32768,20,6 16384,10,1 65536,10,1 4096,16,8 32768,16,8 4096,24,12 32768,24,12 2>&1 | grep SHAPE; done > gpurun_out/r02_c_sweep.log 2>&1
cat gpurun_out/r02_c_sweep.log

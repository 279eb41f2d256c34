cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/enc_grad_check.py b40 > gpurun_out/r02_b_encgrad.log 2>&1; echo encgrad=$?
tail -45 gpurun_out/r02_b_encgrad.log
timeout -k 10 900 python -m pytest tests/test_big_goldens_gpu.py tests/test_model_gpu.py tests/test_reset_gpu.py tests/test_entrypoints_gpu.py tests/test_curriculum_gpu.py tests/test_env_gpu.py -q -m gpu > gpurun_out/r02_b_tests.log 2>&1; echo tests=$?
tail -40 gpurun_out/r02_b_tests.log
timeout -k 10 300 python tools/shape_sweep.py 8192,20,6 65536,20,6 16384,10,1 262144,10,1 65536,15,3 4096,40,16 16384,40,16 4096,32,40 > gpurun_out/r02_b_sweep.log 2>&1; echo sweep=$?
cat gpurun_out/r02_b_sweep.log
for g in 1 2 4; do MAPF_STEP_GROUP=$g timeout -k 10 200 python tools/shape_sweep.py 65536,20,6 262144,10,1 16384,40,16 > gpurun_out/r02_b_sweep_g$g.log 2>&1; echo "group cap $g"; cat gpurun_out/r02_b_sweep_g$g.log; done

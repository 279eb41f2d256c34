cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu > gpurun_out/r03_v_enc_tests.log 2>&1; echo enc_tests=$?; tail -15 gpurun_out/r03_v_enc_tests.log
timeout -k 10 300 python tools/enc_grad_check.py b40 2>&1 | grep -v amdgpu > gpurun_out/r03_v_enc_grad_check.txt; echo check=$?
cat gpurun_out/r03_v_enc_grad_check.txt
timeout -k 10 300 python tools/update_times.py 2>&1 | grep -v amdgpu > gpurun_out/r03_v_update_times.txt; echo ut=$?; tail -30 gpurun_out/r03_v_update_times.txt

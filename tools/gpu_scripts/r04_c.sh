# round 4: graph-replayed update -- parity tests, then host / wall time per update at 6 agents
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_learner_gpu.py tests/test_update_gpu.py -x -q -m gpu > gpurun_out/r04_c_tests.log 2>&1; rc=$?; echo tests=$rc; tail -30 gpurun_out/r04_c_tests.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_c_tests.log; then exit 1; fi
timeout -k 10 300 python tools/update_times.py 6 20 2048 > gpurun_out/r04_c_update_times_6.log 2>&1; rc=$?; echo ut=$rc; tail -12 gpurun_out/r04_c_update_times_6.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_c_update_times_6.log; then exit 1; fi

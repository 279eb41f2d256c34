cd $GRAFT_REPO_ROOT
MODE=update timeout -k 10 200 python -X faulthandler tools/micro/graph_gemm_probe.py > gpurun_out/r04_l_probe.log 2>&1; rc=$?; echo probe=$rc; grep -v "Extension" gpurun_out/r04_l_probe.log | tail -3
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_l_probe.log; then exit 1; fi
timeout -k 10 600 python -m pytest tests/test_curriculum_gpu.py tests/test_update_gpu.py tests/test_learner_gpu.py tests/test_encoder_gpu.py -x -q -m gpu > gpurun_out/r04_l_tests.log 2>&1; rc=$?; echo tests=$rc; tail -6 gpurun_out/r04_l_tests.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_l_tests.log; then exit 1; fi
rm -rf models
timeout -k 10 400 python train.py --envs 512 --minutes 5 > gpurun_out/r04_train_curriculum_5min_graph.log 2> gpurun_out/r04_train_5min_graph.err; rc=$?; echo train_graph=$rc
grep "update speed\|number of updates\|buffer update" gpurun_out/r04_train_curriculum_5min_graph.log | tail -6
if [ $rc -ne 0 ]; then tail -5 gpurun_out/r04_train_5min_graph.err; exit 1; fi
rm -rf models
MAPF_UPDATE_GRAPH=0 timeout -k 10 400 python train.py --envs 512 --minutes 5 > gpurun_out/r04_train_curriculum_5min_nograph.log 2> gpurun_out/r04_train_5min_nograph.err; echo train_nograph=$?
grep "update speed\|number of updates\|buffer update" gpurun_out/r04_train_curriculum_5min_nograph.log | tail -6

# round 2, call A: new parity tests (big goldens, wide recurrence forward, generator statistics, entry points) + a bench line
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_big_goldens_gpu.py tests/test_model_gpu.py tests/test_reset_gpu.py tests/test_entrypoints_gpu.py tests/test_curriculum_gpu.py -x -q -m gpu -s > gpurun_out/r02_a_tests.log 2>&1; echo tests=$?
tail -30 gpurun_out/r02_a_tests.log
timeout -k 10 400 python bench.py > gpurun_out/r02_a_bench.json 2> gpurun_out/r02_a_bench.err; echo bench=$?
tail -c 3000 gpurun_out/r02_a_bench.json; tail -5 gpurun_out/r02_a_bench.err

cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r02_d_tests.log 2>&1; echo tests=$?
tail -40 gpurun_out/r02_d_tests.log
timeout -k 10 400 python bench.py > gpurun_out/r02_d_bench.json 2> gpurun_out/r02_d_bench.err; echo bench=$?
tail -c 2500 gpurun_out/r02_d_bench.json; tail -3 gpurun_out/r02_d_bench.err

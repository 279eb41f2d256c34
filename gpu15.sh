timeout 900 python bench.py --cpu-seconds 6 > gpurun_out/bench3.json 2> gpurun_out/bench3.err; echo rc=$?; cat gpurun_out/bench3.json; tail -3 gpurun_out/bench3.err

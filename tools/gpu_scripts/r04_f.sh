# round 4: multi-handle env launches (parity), env suite, bench sanity (the body refactor must not cost the main kernel), graph update timing
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_env_gpu.py -x -q -m gpu > gpurun_out/r04_f_envtests.log 2>&1; rc=$?; echo envtests=$rc; tail -15 gpurun_out/r04_f_envtests.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_f_envtests.log; then exit 1; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-dqn > gpurun_out/r04_f_bench.json 2> gpurun_out/r04_f_bench.err; echo bench=$?
python -c "
import json;r=json.loads(open('gpurun_out/r04_f_bench.json').read().strip().splitlines()[-1]);ro=r['roofline'];print(r['value'],ro['frac'],ro['kernel_avg_us'],ro.get('frac_out_of_cache'),ro.get('frac_hbm_proper'),ro.get('frac_hbm_proper_2x'))"
timeout -k 10 600 python -m pytest tests/test_learner_gpu.py -x -q -m gpu > gpurun_out/r04_f_tests.log 2>&1; rc=$?; echo tests=$rc; tail -5 gpurun_out/r04_f_tests.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_f_tests.log; then exit 1; fi
ITERS=300 timeout -k 10 300 python tools/update_times.py 6 20 2048 > gpurun_out/r04_f_update_times_6.log 2>&1; rc=$?; echo ut=$rc; tail -5 gpurun_out/r04_f_update_times_6.log

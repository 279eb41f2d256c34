cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-dqn > gpurun_out/r04_h_bench.json 2> gpurun_out/r04_h_bench.err; echo bench=$?
python -c "
import json;r=json.loads(open('gpurun_out/r04_h_bench.json').read().strip().splitlines()[-1]);ro=r['roofline'];print(r['value'],ro['frac'],ro['kernel_avg_us'],ro.get('frac_out_of_cache'),ro.get('frac_hbm_proper'),ro.get('frac_hbm_proper_2x'))"
timeout -k 10 700 python -m pytest tests/test_curriculum_gpu.py tests/test_env_gpu.py -x -q -m gpu > gpurun_out/r04_h_tests.log 2>&1; rc=$?; echo tests=$rc; tail -5 gpurun_out/r04_h_tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-dqn --no-out-of-cache 2>/dev/null | python -c "
import json,sys;r=json.loads(sys.stdin.read().strip().splitlines()[-1]);ro=r['roofline'];print('second run', r['value'],ro['frac'],ro['kernel_avg_us'])"

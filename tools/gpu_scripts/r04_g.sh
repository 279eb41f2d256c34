# round 4: merged-level actor iteration (+ graph): parity tests, iteration timing, bench re-check
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_curriculum_gpu.py tests/test_env_gpu.py tests/test_actor_gpu.py -x -q -m gpu > gpurun_out/r04_g_tests.log 2>&1; rc=$?; echo tests=$rc; tail -25 gpurun_out/r04_g_tests.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_g_tests.log; then exit 1; fi
timeout -k 10 300 python tools/curriculum_iter.py 512 200 > gpurun_out/r04_g_curriculum_iter.log 2>&1; rc=$?; echo iter=$rc; tail -6 gpurun_out/r04_g_curriculum_iter.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_g_curriculum_iter.log; then exit 1; fi
timeout -k 10 300 python tools/curriculum_iter.py 1024 200 > gpurun_out/r04_g_curriculum_iter_1024.log 2>&1; rc=$?; echo iter=$rc; tail -6 gpurun_out/r04_g_curriculum_iter_1024.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-dqn > gpurun_out/r04_g_bench.json 2> gpurun_out/r04_g_bench.err; echo bench=$?
python -c "
import json;r=json.loads(open('gpurun_out/r04_g_bench.json').read().strip().splitlines()[-1]);ro=r['roofline'];print(r['value'],ro['frac'],ro['kernel_avg_us'],ro.get('frac_out_of_cache'),ro.get('frac_hbm_proper'),ro.get('frac_hbm_proper_2x'))"

timeout 900 python -m pytest tests/test_model_gpu.py tests/test_learner_gpu.py tests/test_actor_gpu.py -m gpu -x -q 2>&1 | tail -3 | grep -v amdgpu.ids | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/bench4.json 2> gpurun_out/bench4.err; python -c "
import json; d=json.load(open('gpurun_out/bench4.json')); print(d['value'], d['roofline']['kernel_avg_us'], d['extra'])"

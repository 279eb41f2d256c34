timeout 600 python -m pytest tests/test_learner_gpu.py -m gpu -x -q 2>&1 | tail -5 | grep -v amdgpu.ids | cut -c1-300
timeout 900 python tools/dqn_bench.py 2>&1 | grep -v amdgpu.ids | tail -20

timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | grep -v amdgpu.ids | cut -c1-300
TE=1024 timeout 900 python tools/dqn_bench.py 2>&1 | grep -v amdgpu.ids | tail -12

timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | grep -v amdgpu.ids | cut -c1-300

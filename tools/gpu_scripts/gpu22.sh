mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | grep -v amdgpu.ids | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench6.json 2> gpurun_out/bench6.err; echo bench=$?; cut -c1-2600 gpurun_out/bench6.json

TE=1024 TCAP=512 TB=192 timeout 1400 python tools/nan_debug.py 2>&1 | grep -v amdgpu.ids | grep -E "update|dt|weights|tree"

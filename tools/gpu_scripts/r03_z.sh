cd $GRAFT_REPO_ROOT
PROFILE_HOST=1 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep -v amdgpu > gpurun_out/r03_z_update6.txt; echo u6=$?
cat gpurun_out/r03_z_update6.txt | cut -c1-180

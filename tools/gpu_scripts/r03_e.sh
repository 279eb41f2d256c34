cd $GRAFT_REPO_ROOT
PROFILE_HOST=1 timeout -k 10 300 python tools/actor_times.py 2>&1 | grep -v amdgpu > gpurun_out/r03_e_actor_host.txt; echo rc=$?
grep -n "reuse=" gpurun_out/r03_e_actor_host.txt; sed -n '/function calls/,$p' gpurun_out/r03_e_actor_host.txt | cut -c1-170 | head -60

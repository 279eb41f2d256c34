cd $GRAFT_REPO_ROOT
PROFILE_HOST=1 timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep -v amdgpu | sed -n '/function calls/,$p' | cut -c1-150 | head -70

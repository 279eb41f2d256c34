cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/micro/enc_in_update.py 2>&1 | grep -v amdgpu > gpurun_out/r05_enc_in_update.txt; cat gpurun_out/r05_enc_in_update.txt

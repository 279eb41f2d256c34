cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/micro/enc_small_m.py > gpurun_out/r05_enc_small_m.txt 2>&1; cat gpurun_out/r05_enc_small_m.txt | grep -v amdgpu

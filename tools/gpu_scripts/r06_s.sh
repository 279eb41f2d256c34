#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O
timeout -k 10 300 python tools/micro/enc_in_update.py 2>&1 | grep -v amdgpu | tee $O/enc_in_update.txt

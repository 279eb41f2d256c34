#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r; mkdir -p $O
for rows in 14494 4096 122880; do ROWS=$rows MODES=0,1000,1 TURNS=3 timeout -k 10 200 python tools/micro/enc_ablate.py 2>&1 | grep rows= | tee -a $O/enc_saves_priced.txt; done
for m in base free base free; do
MODE=$m timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 200 2>&1 | grep "MODE=" | tee -a $O/train_loop_free.txt
done

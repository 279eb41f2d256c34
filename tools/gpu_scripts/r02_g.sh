cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/micro/tall_tn.py > gpurun_out/r02_g_tall_tn.log 2>&1; echo talltn=$?
cat gpurun_out/r02_g_tall_tn.log

cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/enc_grad_precision.py b40 2>&1 | grep -v amdgpu > gpurun_out/r03_enc_grad_precision.txt; echo prec=$?
cat gpurun_out/r03_enc_grad_precision.txt | head -12
for nt in 64 128; do
  echo "MAPF_STEP_THREADS=$nt"; MAPF_STEP_THREADS=$nt timeout -k 10 120 python tools/shape_sweep.py 4096,40,16 8192,20,6 4096,32,16 2>&1 | grep -v amdgpu
done
for g in 1 2 4; do
  echo "MAPF_STEP_GROUP=$g"; MAPF_STEP_GROUP=$g timeout -k 10 120 python tools/shape_sweep.py 8192,20,6 2>&1 | grep -v amdgpu
done
TE=4096 TL=40 TN=16 timeout -k 10 60 python tools/stamps.py 2>&1 | grep -v amdgpu
TE=8192 TL=20 TN=6 timeout -k 10 60 python tools/stamps.py 2>&1 | grep -v amdgpu

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
for cfg in "20 1" "21 1" "20 0" "16 1"; do
set -- $cfg
MAPF_WGRAD_PARTS=$1 MAPF_WGRAD_MERGED=$2 timeout -k 10 200 python tools/micro/graph_eager_drift.py 14 2>&1 | grep -v amdgpu.ids | tee -a $O/drift.txt
done
python - <<'PY'
import torch
a=torch.load('/tmp/drift_eager_20_1.pt'); b=torch.load('/tmp/drift_eager_21_1.pt'); c=torch.load('/tmp/drift_eager_20_0.pt')
d=lambda x,y: sum(float((p-q).pow(2).sum()) for p,q in zip(x,y))
print('eager parts20 vs eager parts21', d(a,b), ' eager merged vs eager per-layer', d(a,c))
PY

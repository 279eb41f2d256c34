#!/bin/bash
# round 6: bench line with the tape policy in the reference-shape legs; 6-rank shared-GPU rehearsals (the box's process guard allows 6
# processes on the card, so 8 ranks cannot be rehearsed on it) of bench.py and train.py
cd $GRAFT_REPO_ROOT
O=gpurun_out
t0=$(date +%s)
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r06b_bench.json 2> $O/r06b_bench.err; echo bench=$? $(( $(date +%s) - t0 ))s
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06b_bench.json').read().strip().splitlines()[-1])
for k,v in d.items():
    if (isinstance(v,(int,float)) or v is None or k in ('dqn_error','learner_path','learner_ref_shape_path')) and ('ref_shape' in k or 'curriculum' in k or 'train_loop' in k or 'error' in k): print(k, v)
PY
tail -3 $O/r06b_bench.err
t0=$(date +%s)
MAPF_BENCH_SHARE_GPU=1 timeout -k 10 600 python bench.py --gpus 6 --dist-backend gloo --envs 512 --steps 20 --warmup 5 > $O/r06b_bench_6rank.json 2> $O/r06b_bench_6rank.err; echo bench6=$? $(( $(date +%s) - t0 ))s
python - <<'PY'
import json
ls=[l for l in open('gpurun_out/r06b_bench_6rank.json').read().strip().splitlines() if l.startswith('{')]
print(len(ls), 'lines')
d=json.loads(ls[-1])
for k,v in d.items():
    if isinstance(v,(int,float)) or v is None or k in ('dqn_error','learner_path','learner_ref_shape_path'): print(k, v)
print(json.dumps(d.get('multi_rank'))[:1500])
PY
grep -v amdgpu.ids $O/r06b_bench_6rank.err | tail -8
t0=$(date +%s)
MAPF_TRAIN_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 6 --master-addr 127.0.0.1 --master-port 29611 train.py --agents 40 --map 32 --envs 256 --minutes 1 --interval 15 --learning-starts 20000 --dist-backend gloo > $O/r06b_train_fixed_6rank.log 2>&1; echo train_fixed6=$? $(( $(date +%s) - t0 ))s
grep -v amdgpu.ids $O/r06b_train_fixed_6rank.log | tail -12
t0=$(date +%s)
MAPF_TRAIN_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 6 --master-addr 127.0.0.1 --master-port 29612 train.py --envs 128 --minutes 1 --interval 15 --learning-starts 20000 --dist-backend gloo > $O/r06b_train_curriculum_6rank.log 2>&1; echo train_cur6=$? $(( $(date +%s) - t0 ))s
grep -v amdgpu.ids $O/r06b_train_curriculum_6rank.log | tail -14

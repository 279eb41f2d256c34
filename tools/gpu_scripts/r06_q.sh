#!/bin/bash
# round 6: probes for the config-2 learner (saved activations priced, per-window tile counts), configs[0] on the GPU, the driver's
# launcher (torch.distributed.run) with 2 ranks incl. an injected fault, a 3-minute train.py at config 2's shape, curriculum seed 2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; mkdir -p $O
for rows in 14494 122880; do ROWS=$rows MODES=0,1,1000 timeout -k 10 120 python tools/micro/enc_ablate.py 2>&1 | grep rows= | tee -a $O/enc_saves_priced.txt; done
timeout -k 10 200 python tools/micro/recur_mixed_tiles.py 2>&1 | grep -v amdgpu | tee $O/recur_mixed_tiles.txt
timeout -k 10 300 python tools/config0_times.py 2>&1 | grep -v amdgpu | tee $O/config0_gpu.txt
t0=$(date +%s)
MAPF_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 2 --dist-backend gloo --envs 512 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_torchrun_2rank.json 2> $O/bench_torchrun_2rank.err; echo torchrun_bench=$? $(( $(date +%s) - t0 ))s; grep -c '^{' $O/bench_torchrun_2rank.json
t0=$(date +%s)
MAPF_BENCH_FAULT=1:learner MAPF_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29622 bench.py --gpus 2 --dist-backend gloo --envs 512 --steps 20 --warmup 5 --no-cpu-baseline --no-out-of-cache > $O/bench_torchrun_fault.json 2> $O/bench_torchrun_fault.err; echo torchrun_fault=$? $(( $(date +%s) - t0 ))s; grep -c '^{' $O/bench_torchrun_fault.json; grep "injected\|leaving the job" $O/bench_torchrun_fault.err | head -3
rm -rf models
timeout -k 10 260 python train.py --agents 40 --map 32 --envs 4096 --minutes 3 --interval 20 > $O/train_c2_3min.log 2> $O/train_c2_3min.err; echo train_c2=$?
grep "update speed\|buffer update speed" $O/train_c2_3min.log | tail -8
rm -rf models
t0=$(date +%s)
timeout -k 10 520 python train.py --envs 512 --minutes 8 --interval 20 --seed 2 > $O/train_to_stop_seed2.log 2> $O/train_seed2.err; echo train_seed2=$? $(( $(date +%s) - t0 ))s
grep "number of updates\|update speed" $O/train_to_stop_seed2.log | tail -2
CK=models/$(ls -t models | head -1)
timeout -k 10 200 python tools/eval_checkpoint.py $CK > $O/eval_seed2.txt 2>> $O/train_seed2.err; cat $O/eval_seed2.txt
